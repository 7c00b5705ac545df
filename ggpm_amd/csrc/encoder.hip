// Whole-encoder drivers: HierMPNEncoder.forward / its backward (reference ggpm/encoder.py:96-157) as ONE C call each.
//
// The per-op entry points of this library are what the Python host composes op by op (functional.py); at MI355X
// speeds that composition costs more host time than the GPU needs for the step (~150 ctypes calls, ~140
// allocations, 21 autograd nodes per step).  These two drivers issue exactly the same kernels in the same order
// from C++, carve every intermediate out of two caller-provided arenas (`saved`: forward -> backward, `work`:
// backward temporaries), accumulate gradient contributions in place (GEMM / segmented-sum `accumulate` flags instead
// of separate add kernels), build the transposed CSRs beside the forward on the second stream and run all weight
// gradient contractions there, event-ordered.  GRU message function, dropout 0 (the host keeps the op-by-op path
// for everything else).
#include "common.h"
#include <condition_variable>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

hipEvent_t ggpm_wgrad_event(int i);          // small pool of re-recordable events (mpn_gru.hip)

namespace {

struct Arena {
    char* base;
    size_t off;
    bool overflow;
    size_t cap;
    template <typename T>
    T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        if (base && off > cap) overflow = true;
        return p;
    }
};

struct Csr {
    int32_t *rowptr, *col;
    int rows, ncols, cap;
    int32_t *rowptrT, *colT, *cursor;
};

struct LevelSaved {
    float *X, *Hs, *Cs, *Qs, *St, *wpack, *nei;
};

struct Saved {          // layout of the `saved` arena (pointers are re-derived by replaying the same takes)
    Csr gpred, gagr, tpred, tagr, tcgr, tsrc, root, motif, attach;
    int32_t *src, *attr0, *motif_id, *attach_id, *iota;
    float *hnode_a, *hmess_a;
    LevelSaved lv[3];          // 0 tree, 1 inter, 2 graph
    float *finput_i, *pooled, *hnode_i, *hmess_i, *finput_t, *hnode_t, *hmess_t, *f, *n;
};

struct Dims {
    int H, Hp, He, Hep, depthT, depthG, atom, n_motif, n_attach;
    int N1g, E1g, Kga, Kgb, N1t, E1t, Kta, Ktb, Ktc, B;
    int ld_n, ld_m, ld_t, Ig, It;
    int lstm, nX, lcount;          // cell type, hoisted-input slots per level (3 / 4), parameter slots per level (9 / 10)
    int tree_chain;                // longest dependency chain of the tree messages (0: unknown)
    float dropout;                 // training-mode drop probability (0: none)
    unsigned int seed_lo, seed_hi;
    int gate_dtype;                // 0 fp32, 1 bf16 operands for the gate products of the depth loops, 2 fp32 on fp32 MFMA only
};

// the level calls of this thread take the gate-product dtype from a thread-local (tile_mma.h) for the driver's duration
struct GateDtypeScope {
    int prev;
    explicit GateDtypeScope(int dtype) : prev(ggpm_gate_dtype()) { ggpm_set_gate_dtype(dtype); }
    ~GateDtypeScope() { ggpm_set_gate_dtype(prev); }      // (nests: the side-stream bodies may run on the caller's thread)
};

// dropout sites (include/ggpm_hip.h)
enum { DS_EI = 0, DS_EC, DS_WO_ATOM, DS_WI, DS_WO_INTER, DS_WC, DS_WO_TREE };
inline int drop(const Dims& d, float* x, int rows, int cols, int ld, int site, ggpm_stream_t s) {
    if (d.dropout <= 0.f) return GGPM_OK;
    return ggpm_dropout(x, rows, cols, ld, d.dropout, d.seed_lo, d.seed_hi, site, s);
}

Dims make_dims(const ggpm_enc_dims* d) {
    Dims x;
    x.H = d->H; x.Hp = ggpm_padded_hidden(d->H); x.He = d->He; x.Hep = ggpm_padded_hidden(d->He);
    x.depthT = d->depthT; x.depthG = d->depthG; x.atom = d->atom_size; x.n_motif = d->n_motif; x.n_attach = d->n_attach;
    x.N1g = d->N1g; x.E1g = d->E1g; x.Kga = d->Kg_a; x.Kgb = d->Kg_b;
    x.N1t = d->N1t; x.E1t = d->E1t; x.Kta = d->Kt_a; x.Ktb = d->Kt_b; x.Ktc = d->Kt_c; x.B = d->B;
    x.ld_n = ggpm_round_up(x.atom, 4);
    x.Ig = x.atom + 4 + 20; x.ld_m = ggpm_round_up(x.Ig, 4);
    x.It = x.H + 20; x.ld_t = ggpm_round_up(x.It, 4);
    x.lstm = d->rnn_type == 1; x.nX = x.lstm ? 4 : 3; x.lcount = x.lstm ? 10 : 9;
    x.tree_chain = d->tree_chain;
    x.dropout = d->dropout; x.seed_lo = d->seed_lo; x.seed_hi = d->seed_hi;
    x.gate_dtype = (d->gate_dtype >= 1 && d->gate_dtype <= 3) ? d->gate_dtype : 0;
    return x;
}

void take_csr(Arena& A, Csr& c, int rows, int ncols, int cap) {
    c.rows = rows; c.ncols = ncols; c.cap = cap < 1 ? 1 : cap;
    c.rowptr = A.take<int32_t>(rows + 1);
    c.col = A.take<int32_t>(c.cap);
    c.rowptrT = A.take<int32_t>(ncols + 1);
    c.colT = A.take<int32_t>(c.cap);
    c.cursor = A.take<int32_t>(ncols);
}

void take_level(Arena& A, LevelSaved& L, int E1, int N1, int Hp, int H, int depth, bool lstm) {
    const size_t slot = (size_t)E1 * Hp;
    L.X = A.take<float>((lstm ? 4 : 3) * slot);
    L.Hs = A.take<float>((size_t)(depth + 1) * slot);
    L.Cs = lstm ? A.take<float>((size_t)(depth + 1) * slot) : nullptr;
    L.Qs = A.take<float>((size_t)depth * slot);
    L.St = A.take<float>((size_t)5 * depth * slot);
    L.wpack = A.take<float>(lstm ? ggpm_lstm_pack_floats(H) : ggpm_gru_pack_floats(H));
    L.nei = A.take<float>((size_t)N1 * Hp);
}

void layout_saved(Arena& A, const Dims& d, Saved& s) {
    take_csr(A, s.gpred, d.E1g, d.E1g, d.E1g * d.Kgb);
    take_csr(A, s.gagr, d.N1g, d.E1g, d.N1g * d.Kga);
    take_csr(A, s.tpred, d.E1t, d.E1t, d.E1t * d.Ktb);
    take_csr(A, s.tagr, d.N1t, d.E1t, d.N1t * d.Kta);
    take_csr(A, s.tcgr, d.N1t, d.N1g, d.N1t * d.Ktc);
    s.src = A.take<int32_t>(d.E1t); s.attr0 = A.take<int32_t>(d.E1t);
    s.motif_id = A.take<int32_t>(d.N1t); s.attach_id = A.take<int32_t>(d.N1t);
    const int nio = (d.E1t > d.N1t ? d.E1t : d.N1t) + 1;
    s.iota = A.take<int32_t>(nio > d.B + 1 ? nio : d.B + 1);
    // index "CSRs" (one entry per row): rowptr = iota, col = the index array; only their transposes are stored
    auto take_index = [&](Csr& c, int rows, int ncols) {
        c.rows = rows; c.ncols = ncols; c.cap = rows;
        c.rowptr = s.iota; c.col = nullptr;
        c.rowptrT = A.take<int32_t>(ncols + 1); c.colT = A.take<int32_t>(rows); c.cursor = A.take<int32_t>(ncols);
    };
    take_index(s.tsrc, d.E1t, d.N1t);
    take_index(s.root, d.B, d.N1t);
    take_index(s.motif, d.N1t, d.n_motif);
    take_index(s.attach, d.N1t, d.n_attach);
    s.hnode_a = A.take<float>((size_t)d.N1g * d.ld_n);
    s.hmess_a = A.take<float>((size_t)d.E1g * d.ld_m);
    take_level(A, s.lv[0], d.E1t, d.N1t, d.Hp, d.H, d.depthT, d.lstm);
    take_level(A, s.lv[1], d.E1t, d.N1t, d.Hp, d.H, d.depthT, d.lstm);
    take_level(A, s.lv[2], d.E1g, d.N1g, d.Hp, d.H, d.depthG, d.lstm);
    s.finput_i = A.take<float>((size_t)d.N1t * d.Hep); s.pooled = A.take<float>((size_t)d.N1t * d.Hp);
    s.hnode_i = A.take<float>((size_t)d.N1t * d.Hp); s.hmess_i = A.take<float>((size_t)d.E1t * d.ld_t);
    s.finput_t = A.take<float>((size_t)d.N1t * d.Hep);
    s.hnode_t = A.take<float>((size_t)d.N1t * d.Hp); s.hmess_t = A.take<float>((size_t)d.E1t * d.ld_t);
    s.f = A.take<float>((size_t)d.B * d.Hp); s.n = A.take<float>((size_t)d.B * d.Hp);
}

__global__ void iota_k(int32_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// parameter slots
enum { P_EC = 0, P_EI, P_WC, P_BC, P_WI, P_BI, P_WROOT, P_BROOT, P_LEVEL0 };
enum { L_WO = 0, L_BO, L_WZ, L_BZ, L_WR, L_UR, L_BU, L_WH, L_BH, L_COUNT };               // GRU level
enum { Q_WO = 0, Q_BO, Q_WI, Q_BI, Q_WOG, Q_BOG, Q_WU, Q_BU, Q_WF, Q_BF, Q_COUNT };      // LSTM level
inline int lp(int level, int which) { return P_LEVEL0 + level * L_COUNT + which; }
inline int lq(int level, int which) { return P_LEVEL0 + level * Q_COUNT + which; }
inline int lwo(bool lstm, int level) { return lstm ? lq(level, Q_WO) : lp(level, L_WO); }
inline int lbo(bool lstm, int level) { return lstm ? lq(level, Q_BO) : lp(level, L_BO); }

// GGPM_SPLIT_TAIL=0: the input-half gradients of the last level behind its tall contractions on the second stream again
inline bool split_tail_enabled() { static const bool v = !(ggpm_dev_env("GGPM_SPLIT_TAIL") && atoi(ggpm_dev_env("GGPM_SPLIT_TAIL")) == 0); return v; }
// ---- launch worker of the second stream -----------------------------------------------------------------------------------
// A step issues ~285 launches, ~90 of them on the second stream (index transposes beside the forward, every weight-gradient
// contraction of the backward).  They are independent of what the calling thread issues next, so a worker thread issues
// them: the caller records the ordering event on its own stream, queues a closure and goes on with the latency-critical
// depth loops; the worker waits on the event (on the second stream) and launches.  The drivers drain the queue before
// they return (and before anything waits on the second stream), so no closure outlives the buffers it names.
// GGPM_SIDE_WORKER=0: the calling thread issues everything itself, in the same order.
class SideWorker {
public:
    static SideWorker* get() {
        static const bool on = !(ggpm_dev_env("GGPM_SIDE_WORKER") && atoi(ggpm_dev_env("GGPM_SIDE_WORKER")) == 0);
        if (!on) return nullptr;
        static SideWorker w;
        return &w;
    }
    void post(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(std::move(job));
        }
        cv_.notify_one();
    }
    // blocks until everything posted so far has been issued; -> the first error of those jobs
    int drain() {
        std::unique_lock<std::mutex> lk(mu_);
        idle_.wait(lk, [&] { return q_.empty() && !busy_; });
        const int e = err_;
        err_ = GGPM_OK;
        return e;
    }
    void use_device(int dev) {
        std::lock_guard<std::mutex> lk(mu_);
        want_dev_ = dev;
    }

private:
    SideWorker() : th_([this] { run(); }) {}
    ~SideWorker() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_one();
        if (th_.joinable()) th_.join();
    }
    void run() {
        int dev = -1;
        for (;;) {
            std::function<int()> job;
            int want;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;          // stop requested and nothing left
                job = std::move(q_.front());
                q_.pop_front();
                busy_ = true;
                want = want_dev_;
            }
            if (want != dev && want >= 0) { (void)hipSetDevice(want); dev = want; }
            int rc = job();
            if (!rc && hipGetLastError() != hipSuccess) rc = GGPM_ERR_LAUNCH;       // (the error state is per thread)
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (rc && !err_) err_ = rc;
                busy_ = false;
                if (q_.empty()) idle_.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    std::deque<std::function<int()>> q_;
    bool busy_ = false, stop_ = false;
    int err_ = GGPM_OK, want_dev_ = -1;
    std::thread th_;          // last member: starts when everything above is constructed
};

// every return path of a driver leaves the worker idle (the closures name caller buffers)
struct DrainGuard {
    SideWorker* wk;
    ~DrainGuard() { if (wk) (void)wk->drain(); }
};

#define CK(expr)                    \
    do {                            \
        const int rc__ = (expr);    \
        if (rc__) return rc__;      \
    } while (0)

// y[:, :N] = act(sum_i x_i W[:, off_i : off_i + K_i]^T + b), pad columns zero (functional._Linear.forward)
int linear2(int M, int N, const float* x1, int ld1, int K1, const float* x2, int ld2, int K2, const float* W,
            const float* b, int act, int zero_row0, float* y, int ldy, ggpm_stream_t s) {
    const int ldw = K1 + K2;
    const float* A[2] = {x1, x2};
    const float* B[2] = {W, W + K1};
    const int lda[2] = {ld1, ld2}, ldb[2] = {ldw, ldw}, K[2] = {K1, K2};
    return ggpm_gemm_ksegments(1, M, N, 2, A, lda, B, ldb, K, y, ldy, ldy, b, 0, act, zero_row0, s);
}

// Copies slot `src` of up to 8 slotted arrays to their slots lo..hi (a level that reached its fixed point early).
struct ReplicateArgs {
    float* base[8];
    int src[8], lo[8], hi[8];
    size_t slot4;      // float4 per slot
};
__global__ void __launch_bounds__(256) replicate_slots_k(ReplicateArgs r) {
    const int a = blockIdx.y;
    if (r.hi[a] < r.lo[a]) return;
    float4* base = reinterpret_cast<float4*>(r.base[a]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < r.slot4; i += (size_t)gridDim.x * 256) {
        const float4 v = base[(size_t)r.src[a] * r.slot4 + i];
        for (int k = r.lo[a]; k <= r.hi[a]; ++k) base[(size_t)k * r.slot4 + i] = v;
    }
}

// steps a tree-side level really has to run: chain + 1 (the step after the last change also fixes the stash)
// (a level whose depth-loop arrays are kept in bf16 -- gate mode 1 on >= 6144 messages, tile_mma.h -- runs all its steps: the
// replication below copies fp32 slots)
inline bool bf16_stored(const Dims& d, int E1) { return d.gate_dtype == 1 && ggpm_bf16_storage_applies(E1, d.H); }
inline int run_steps(const Dims& d, int level, int depth, int E1) {
    static const bool off = ggpm_dev_env("GGPM_TREE_FIXED_POINT") && atoi(ggpm_dev_env("GGPM_TREE_FIXED_POINT")) == 0;
    if (off || level == 2 || bf16_stored(d, E1) || d.tree_chain <= 0 || d.tree_chain + 1 >= depth) return depth;
    return d.tree_chain + 1;
}

// first step the backward of a tree-side level has to run: lo = max(1, D - chain + 1) (common.h: below it d(h^t) is
// exactly zero); 1 = all steps
inline int backward_lo(const Dims& d, int level, int depth, int E1) {
    static const bool off = ggpm_dev_env("GGPM_TREE_FIXED_POINT") && atoi(ggpm_dev_env("GGPM_TREE_FIXED_POINT")) == 0;
    if (off || level == 2 || bf16_stored(d, E1) || d.tree_chain <= 0) return 1;
    const int lo = depth - d.tree_chain + 1;
    return lo < 1 ? 1 : lo;
}

int replicate_tail(const Dims& d, int E1, int depth, int run, int level, const LevelSaved& L, ggpm_stream_t s) {
    if (run >= depth) return GGPM_OK;
    const size_t slot = (size_t)E1 * d.Hp, ds = (size_t)depth * slot;
    const int blo = backward_lo(d, level, depth, E1);     // the backward only reads state slots >= blo, stash slots >= blo - 1
    ReplicateArgs r = {};
    int n = 0;
    auto add = [&](float* base, int src, int lo, int hi) {
        r.base[n] = base; r.src[n] = src; r.lo[n] = lo > src ? lo : src + 1; r.hi[n] = hi; ++n;
    };
    add(L.Hs, run, blo, depth);                           // h^run == h^t for every later t
    if (L.Cs) add(L.Cs, run, blo, depth);
    add(L.Qs, run, blo, depth - 1);
    for (int k = 0; k < 5; ++k) add(L.St + k * ds, run - 1, blo - 1, depth - 1);     // stash slot of step t is t - 1
    r.slot4 = slot / 4;
    dim3 grid((unsigned)ggpm_ceil_div((int)r.slot4, 256 * 4), n);
    replicate_slots_k<<<grid, 256, 0, (hipStream_t)s>>>(r);
    return GGPM_OK;
}

int level_forward(const Dims& d, int E1, int N1, int I, int depth, const float* x, int ldx, float* const* P, int level,
                  const Csr& pred, const Csr& agr, LevelSaved& L, ggpm_stream_t s) {
    const int H = d.H, Hp = d.Hp;
    const size_t slot = (size_t)E1 * Hp;
    const int run = run_steps(d, level, depth, E1);
    struct Tag { Tag(int level) { ggpm_timing_tag(3 - level); } ~Tag() { ggpm_timing_tag(0); } } tag(level);
    if (d.lstm) {
        const float* W[4] = {P[lq(level, Q_WI)], P[lq(level, Q_WOG)], P[lq(level, Q_WU)], P[lq(level, Q_WF)]};
        const float* b[4] = {P[lq(level, Q_BI)], P[lq(level, Q_BOG)], P[lq(level, Q_BU)], P[lq(level, Q_BF)]};
        GgpmGemmProblem gp[4];
        for (int k = 0; k < 4; ++k) gp[k] = {x, ldx, W[k], I + H, L.X + k * slot, Hp, Hp, b[k], 0, GGPM_ACT_NONE, 0};
        CK(ggpm_gemm_grouped(0, 1, E1, H, I, 4, gp, s));      // the four input projections in one launch
        const size_t dsl = (size_t)depth * slot;
        ggpm_forward_run_depth(run);
        CK(ggpm_lstm_forward(E1, H, depth, L.X, L.X + slot, L.X + 2 * slot, L.X + 3 * slot, W[0] + I, I + H, W[1] + I, I + H,
                             W[2] + I, I + H, W[3] + I, I + H, pred.rowptr, pred.col, L.Hs, L.Cs, L.Qs, L.St, L.St + dsl,
                             L.St + 2 * dsl, L.St + 3 * dsl, L.St + 4 * dsl, L.wpack, 1, s));
        CK(replicate_tail(d, E1, depth, run, level, L, s));
        CK(ggpm_segment_sum(L.Hs + (size_t)depth * slot, Hp, agr.rowptr, agr.col, N1, H, L.nei, Hp, 0, Hp, s));
        return GGPM_OK;
    }
    const float *Wz = P[lp(level, L_WZ)], *Wr = P[lp(level, L_WR)], *Wh = P[lp(level, L_WH)];
    const GgpmGemmProblem gp[3] = {{x, ldx, Wz, I + H, L.X, Hp, Hp, P[lp(level, L_BZ)], 0, GGPM_ACT_NONE, 0},
                                   {x, ldx, Wr, I, L.X + slot, Hp, Hp, nullptr, 0, GGPM_ACT_NONE, 0},
                                   {x, ldx, Wh, I + H, L.X + 2 * slot, Hp, Hp, P[lp(level, L_BH)], 0, GGPM_ACT_NONE, 0}};
    CK(ggpm_gemm_grouped(0, 1, E1, H, I, 3, gp, s));          // the three input projections in one launch
    const size_t ds = (size_t)depth * slot;
    ggpm_forward_run_depth(run);
    CK(ggpm_gru_forward(E1, H, depth, L.X, L.X + slot, L.X + 2 * slot, Wz + I, I + H, P[lp(level, L_UR)], H,
                            P[lp(level, L_BU)], Wh + I, I + H, pred.rowptr, pred.col,
                            L.Hs, L.Qs, L.St, L.St + ds, L.St + 2 * ds, L.St + 3 * ds, L.St + 4 * ds, L.wpack, 1, s));
    CK(replicate_tail(d, E1, depth, run, level, L, s));
    CK(ggpm_segment_sum(L.Hs + (size_t)depth * slot, Hp, agr.rowptr, agr.col, N1, H, L.nei, Hp, 0, Hp, s));
    return GGPM_OK;
}

int transpose(const Csr& c, const int32_t* col, ggpm_stream_t s) {
    return ggpm_csr_transpose(c.rowptr, col ? col : c.col, c.rows, c.ncols, c.rowptrT, c.colT, c.cursor, s);
}

}  // namespace

extern "C" size_t ggpm_encoder_saved_bytes(const ggpm_enc_dims* dims) {
    if (!dims) return 0;
    const Dims d = make_dims(dims);
    Arena A = {nullptr, 0, false, 0};
    Saved s;
    layout_saved(A, d, s);
    return A.off;
}

extern "C" int ggpm_encoder_forward(const ggpm_enc_dims* dims, float* const* params, const int64_t* tfnode,
                                    const int64_t* tfmess, const int64_t* tagraph, const int64_t* tbgraph,
                                    const int64_t* tcgraph, const int64_t* gfnode, const int64_t* gfmess,
                                    const int64_t* gagraph, const int64_t* gbgraph, const int32_t* roots, void* saved,
                                    size_t saved_bytes, float* hroot, float* hnode, float* hinter, float* hatom,
                                    ggpm_stream_t stream, ggpm_stream_t side_stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!dims || !params || !tfnode || !tfmess || !tagraph || !tbgraph || !tcgraph || !gfnode || !gfmess || !gagraph ||
        !gbgraph || !roots || !saved || !hroot || !hnode || !hinter || !hatom)
        return GGPM_ERR_ARG;
    const Dims d = make_dims(dims);
    GateDtypeScope gate_scope(d.gate_dtype);
    Arena A = {reinterpret_cast<char*>(saved), 0, false, saved_bytes};
    Saved S;
    layout_saved(A, d, S);
    if (A.overflow) return GGPM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* const* P = params;
    const int H = d.H, Hp = d.Hp, He = d.He;

    // ---- graph layout.  Only the atom level's two CSRs sit in front of the first depth loop; the tree-side layout (three
    // CSRs, four index columns, iota) is built on the second stream beside the atom level and joined before the
    // attachment level, and the transposes (read by the backward only) follow there.
    ggpm_stream_t ts = side_stream ? side_stream : stream;
    SideWorker* wk = side_stream ? SideWorker::get() : nullptr;
    DrainGuard guard = {wk};
    hipEvent_t ev_in = nullptr;
    if (side_stream) {
        ev_in = ggpm_wgrad_event(60);
        if (!ev_in) return GGPM_ERR_LAUNCH;
        (void)hipEventRecord(ev_in, s);              // the caller's index tensors are ready from here on
    }
    CK(ggpm_padded_to_csr(gbgraph, d.E1g, d.Kgb, S.gpred.rowptr, S.gpred.col, stream));
    CK(ggpm_padded_to_csr(gagraph, d.N1g, d.Kga, S.gagr.rowptr, S.gagr.col, stream));
    hipEvent_t ev_atom = nullptr, ev_tree = nullptr;
    if (side_stream) {
        ev_atom = ggpm_wgrad_event(54);
        ev_tree = ggpm_wgrad_event(53);
        if (!ev_atom || !ev_tree) return GGPM_ERR_LAUNCH;
        (void)hipEventRecord(ev_atom, s);
    }
    // everything the second stream does in the forward, as one closure (issued by the launch worker when there is one)
    auto tree_layout = [=]() -> int {
        if (side_stream) (void)hipStreamWaitEvent((hipStream_t)side_stream, ev_in, 0);
        CK(ggpm_padded_to_csr(tbgraph, d.E1t, d.Ktb, S.tpred.rowptr, S.tpred.col, ts));
        CK(ggpm_padded_to_csr(tagraph, d.N1t, d.Kta, S.tagr.rowptr, S.tagr.col, ts));
        CK(ggpm_padded_to_csr(tcgraph, d.N1t, d.Ktc, S.tcgr.rowptr, S.tcgr.col, ts));
        CK(ggpm_extract_column(tfmess, d.E1t, 4, 0, S.src, ts));
        CK(ggpm_extract_column(tfmess, d.E1t, 4, 2, S.attr0, ts));
        CK(ggpm_extract_column(tfnode, d.N1t, 2, 0, S.motif_id, ts));
        CK(ggpm_extract_column(tfnode, d.N1t, 2, 1, S.attach_id, ts));
        {
            const int nio = (d.E1t > d.N1t ? d.E1t : d.N1t) + 1;
            const int n = nio > d.B + 1 ? nio : d.B + 1;
            iota_k<<<ggpm_ceil_div(n, 256), 256, 0, (hipStream_t)ts>>>(S.iota, n);
        }
        // inputs of the two tree-side levels that do not depend on the level below: embedding rows, one-hot bond positions
        CK(ggpm_gather_rows(P[P_EI], He, S.attach_id, d.N1t, He, S.finput_i, d.Hep, 0, d.Hep, ts));
        CK(drop(d, S.finput_i, d.N1t, He, d.Hep, DS_EI, ts));
        CK(ggpm_gather_rows(P[P_EC], He, S.motif_id, d.N1t, He, S.finput_t, d.Hep, 0, d.Hep, ts));
        CK(drop(d, S.finput_t, d.N1t, He, d.Hep, DS_EC, ts));
        CK(ggpm_onehot(S.attr0, d.E1t, 20, S.hmess_i, d.ld_t, H, d.ld_t, ts));
        CK(ggpm_onehot(S.attr0, d.E1t, 20, S.hmess_t, d.ld_t, H, d.ld_t, ts));
        if (side_stream) {
            (void)hipEventRecord(ev_tree, (hipStream_t)side_stream);
            (void)hipStreamWaitEvent((hipStream_t)side_stream, ev_atom, 0);     // the atom CSRs, for their transposes
        }
        CK(transpose(S.gpred, nullptr, ts));
        CK(transpose(S.gagr, nullptr, ts));
        CK(transpose(S.tpred, nullptr, ts));
        CK(transpose(S.tagr, nullptr, ts));
        CK(transpose(S.tcgr, nullptr, ts));
        CK(transpose(S.tsrc, S.src, ts));
        CK(transpose(S.root, roots, ts));
        CK(transpose(S.motif, S.motif_id, ts));
        CK(transpose(S.attach, S.attach_id, ts));
        return GGPM_OK;
    };
    if (wk) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        wk->use_device(dev);
        wk->post(tree_layout);
    } else {
        CK(tree_layout());
    }

    // ---- atom level (embed_graph, graph_encoder)
    CK(ggpm_embed_graph(gfnode, d.N1g, gfmess, d.E1g, d.atom, 4, 20, S.hnode_a, d.ld_n, S.hmess_a, d.ld_m, stream));
    CK(level_forward(d, d.E1g, d.N1g, d.Ig, d.depthG, S.hmess_a, d.ld_m, P, 2, S.gpred, S.gagr, S.lv[2], stream));
    CK(linear2(d.N1g, H, S.hnode_a, d.ld_n, d.atom, S.lv[2].nei, Hp, H, P[lwo(d.lstm, 2)], P[lbo(d.lstm, 2)], GGPM_ACT_RELU, 1,
               hatom, Hp, stream));
    CK(drop(d, hatom, d.N1g, H, Hp, DS_WO_ATOM, stream));

    // ---- attachment level (embed_inter, inter_encoder)
    if (wk) CK(wk->drain());                                       // (ev_tree is recorded once the closure has run)
    if (side_stream) (void)hipStreamWaitEvent(s, ev_tree, 0);      // tree-side layout built beside the atom level
    CK(ggpm_segment_sum(hatom, Hp, S.tcgr.rowptr, S.tcgr.col, d.N1t, H, S.pooled, Hp, 0, Hp, stream));
    CK(linear2(d.N1t, H, S.finput_i, d.Hep, He, S.pooled, Hp, H, P[P_WI], P[P_BI], GGPM_ACT_RELU, 0, S.hnode_i, Hp,
               stream));
    CK(drop(d, S.hnode_i, d.N1t, H, Hp, DS_WI, stream));
    CK(ggpm_gather_rows(S.hnode_i, Hp, S.src, d.E1t, H, S.hmess_i, d.ld_t, 0, 0, stream));
    CK(level_forward(d, d.E1t, d.N1t, d.It, d.depthT, S.hmess_i, d.ld_t, P, 1, S.tpred, S.tagr, S.lv[1], stream));
    CK(linear2(d.N1t, H, S.hnode_i, Hp, H, S.lv[1].nei, Hp, H, P[lwo(d.lstm, 1)], P[lbo(d.lstm, 1)], GGPM_ACT_RELU, 1, hinter, Hp,
               stream));
    CK(drop(d, hinter, d.N1t, H, Hp, DS_WO_INTER, stream));

    // ---- motif level (embed_tree, tree_encoder)
    CK(linear2(d.N1t, H, S.finput_t, d.Hep, He, hinter, Hp, H, P[P_WC], P[P_BC], GGPM_ACT_RELU, 0, S.hnode_t, Hp, stream));
    CK(drop(d, S.hnode_t, d.N1t, H, Hp, DS_WC, stream));
    CK(ggpm_gather_rows(S.hnode_t, Hp, S.src, d.E1t, H, S.hmess_t, d.ld_t, 0, 0, stream));
    CK(level_forward(d, d.E1t, d.N1t, d.It, d.depthT, S.hmess_t, d.ld_t, P, 0, S.tpred, S.tagr, S.lv[0], stream));
    CK(linear2(d.N1t, H, S.hnode_t, Hp, H, S.lv[0].nei, Hp, H, P[lwo(d.lstm, 0)], P[lbo(d.lstm, 0)], GGPM_ACT_RELU, 1, hnode, Hp,
               stream));
    CK(drop(d, hnode, d.N1t, H, Hp, DS_WO_TREE, stream));

    // ---- root readout (embed_root)
    CK(ggpm_gather_rows(S.hnode_t, Hp, roots, d.B, H, S.f, Hp, 0, Hp, stream));
    CK(ggpm_gather_rows(S.lv[0].nei, Hp, roots, d.B, H, S.n, Hp, 0, Hp, stream));
    CK(linear2(d.B, H, S.f, Hp, H, S.n, Hp, H, P[P_WROOT], P[P_BROOT], GGPM_ACT_TANH, 0, hroot, Hp, stream));
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

namespace {

struct BwdWork {
    float *dpre_root, *df, *dn, *d_hnode_t, *d_nei_t, *dpre, *d_h, *dX, *dx_mess, *d_finput, *d_hinter, *d_hnode_i,
        *d_nei_i, *d_pooled, *d_hatom, *d_nei_g, *level_work, *skws, *csws;
    size_t level_work_bytes, skws_bytes;
    // per level: split-K slabs and column-sum scratch of its input-half weight gradients (level_backward's x_part runs on
    // either stream, beside the other levels' parts: nothing shared)
    float *xws[3], *xcs[3];
    size_t xws_bytes;
};

void layout_work(Arena& A, const Dims& d, BwdWork& w) {
    const int Hp = d.Hp;
    const int Nmax = d.N1g > d.N1t ? d.N1g : d.N1t, Emax = d.E1g > d.E1t ? d.E1g : d.E1t;
    w.dpre_root = A.take<float>((size_t)d.B * Hp); w.df = A.take<float>((size_t)d.B * Hp);
    w.dn = A.take<float>((size_t)d.B * Hp);
    w.d_hnode_t = A.take<float>((size_t)d.N1t * Hp); w.d_nei_t = A.take<float>((size_t)d.N1t * Hp);
    // dpre of the five linear layers: the weight-gradient stream still reads one while the main stream writes the
    // next, so each gets its own buffer
    w.dpre = A.take<float>((size_t)5 * Nmax * Hp);
    w.d_h = A.take<float>((size_t)Emax * Hp);
    w.dX = A.take<float>((size_t)3 * 4 * Emax * Hp);          // per level (read by the second stream afterwards)
    w.dx_mess = A.take<float>((size_t)d.E1t * d.ld_t);
    w.d_finput = A.take<float>((size_t)2 * d.N1t * d.Hep);
    w.d_hinter = A.take<float>((size_t)d.N1t * Hp);
    w.d_hnode_i = A.take<float>((size_t)d.N1t * Hp); w.d_nei_i = A.take<float>((size_t)d.N1t * Hp);
    w.d_pooled = A.take<float>((size_t)d.N1t * Hp);
    w.d_hatom = A.take<float>((size_t)d.N1g * Hp); w.d_nei_g = A.take<float>((size_t)d.N1g * Hp);
    size_t lw = d.lstm ? ggpm_lstm_backward_workspace_bytes(d.E1g, d.H, d.depthG)
                       : ggpm_gru_backward_workspace_bytes(d.E1g, d.H, d.depthG);
    const size_t lwt = d.lstm ? ggpm_lstm_backward_workspace_bytes(d.E1t, d.H, d.depthT)
                              : ggpm_gru_backward_workspace_bytes(d.E1t, d.H, d.depthT);
    w.level_work_bytes = lw > lwt ? lw : lwt;
    w.level_work = A.take<float>(3 * (w.level_work_bytes / 4 + 64));      // one per level (stashes read by stream 2)
    size_t sk = 0;
    const int Imax = d.It > d.Ig ? d.It : d.Ig;
    const size_t cand[] = {ggpm_gemm_workspace_bytes(d.H, Imax, Emax), ggpm_gemm_workspace_bytes(d.H, d.Hp, Nmax),
                           ggpm_gemm_workspace_bytes(d.H, d.Hp, d.B)};
    for (size_t c : cand) sk = c > sk ? c : sk;
    w.skws_bytes = sk + 1024;
    w.skws = A.take<float>(w.skws_bytes / 4);
    w.csws = A.take<float>((size_t)256 * (Hp > d.ld_t ? Hp : d.ld_t));
    const int gates = d.lstm ? 4 : 3;
    const size_t xa = ggpm_gemm_grouped_splitk_workspace_bytes(d.H, d.It, d.E1t, gates);
    const size_t xb = ggpm_gemm_grouped_splitk_workspace_bytes(d.H, d.Ig, d.E1g, gates);
    w.xws_bytes = (xa > xb ? xa : xb) + 256;
    for (int l = 0; l < 3; ++l) {
        w.xws[l] = A.take<float>(w.xws_bytes / 4);
        w.xcs[l] = A.take<float>((size_t)256 * Hp);
    }
}

struct Streams {
    ggpm_stream_t main, side;
    int n_ev;
    SideWorker* wk;          // issues the second stream's launches when set (see SideWorker)
    // order the second stream behind everything issued on the main stream so far
    int side_after_main() {
        if (!side) return GGPM_OK;
        hipEvent_t ev = ggpm_wgrad_event(n_ev++ & 31);
        if (!ev) return GGPM_ERR_LAUNCH;
        (void)hipEventRecord(ev, (hipStream_t)main);
        ggpm_stream_t sd = side;
        return on_side([sd, ev]() -> int {
            (void)hipStreamWaitEvent((hipStream_t)sd, ev, 0);
            return GGPM_OK;
        });
    }
    // run `f` (launches on w()) in issue order with the other second-stream work
    template <class F>
    int on_side(F f) {
        if (!wk) return f();
        wk->post(std::function<int()>(std::move(f)));
        return GGPM_OK;
    }
    int drain() { return wk ? wk->drain() : GGPM_OK; }
    ggpm_stream_t w() const { return side ? side : main; }      // where weight gradients go
};

// d(x1), d(x2) of y = act(x1 W1^T + x2 W2^T + b) from dpre: two products with the same left operand, one launch when
// the halves have the same width
int linear2_dx(int M, int H, int K1, const float* dpre, int ldp, const float* W, float* d1, int ld1, int acc1, float* d2,
               int ld2, int acc2, ggpm_stream_t s) {
    const int ldw = K1 + H;
    if (K1 == H) {
        const GgpmGemmProblem gp[2] = {{dpre, ldp, W, ldw, d1, ld1, ld1, nullptr, acc1, GGPM_ACT_NONE, 0},
                                       {dpre, ldp, W + K1, ldw, d2, ld2, ld2, nullptr, acc2, GGPM_ACT_NONE, 0}};
        return ggpm_gemm_grouped(0, 0, M, H, H, 2, gp, s);
    }
    CK(ggpm_gemm(0, 0, M, K1, H, dpre, ldp, W, ldw, d1, ld1, ld1, nullptr, acc1, GGPM_ACT_NONE, 0, nullptr, 0, s));
    CK(ggpm_gemm(0, 0, M, H, H, dpre, ldp, W + K1, ldw, d2, ld2, ld2, nullptr, acc2, GGPM_ACT_NONE, 0, nullptr, 0, s));
    return GGPM_OK;
}

// weight / bias gradients of y = act(x1 W1^T + x2 W2^T + b): dW[:, :K1] = dpre^T x1, dW[:, K1:] = dpre^T x2, db = colsum
int linear2_wgrad(int M, int N, const float* dpre, int ldp, const float* x1, int ld1, int K1, const float* x2, int ld2,
                  int K2, float* dW, float* db, BwdWork& w, Streams& st) {
    CK(st.side_after_main());
    const int ldw = K1 + K2;
    const ggpm_stream_t sw = st.w();
    float* const skws = w.skws;
    float* const csws = w.csws;
    const size_t skb = w.skws_bytes;
    return st.on_side([=]() -> int {
        CK(ggpm_gemm(1, 0, N, K1, M, dpre, ldp, x1, ld1, dW, ldw, K1, nullptr, 0, GGPM_ACT_NONE, 0, skws, skb, sw));
        CK(ggpm_gemm(1, 0, N, K2, M, dpre, ldp, x2, ld2, dW + K1, ldw, K2, nullptr, 0, GGPM_ACT_NONE, 0, skws, skb, sw));
        if (db) CK(ggpm_colsum(dpre, ldp, M, N, db, csws, sw));
        return GGPM_OK;
    });
}

// backward of one level given dHD = d(h_D); dx (the gradient of the level's message inputs) is optional
int level_backward(const Dims& d, int E1, int I, int depth, const float* x, int ldx, float* const* P, float* const* G,
                   int level, const Csr& pred, const LevelSaved& L, const float* dHD, float* dX, float* level_work,
                   float* dx, int lddx, BwdWork& w, Streams& st) {
    const int H = d.H, Hp = d.Hp;
    const size_t slot = (size_t)E1 * Hp, ds = (size_t)depth * slot;
    const int blo = backward_lo(d, level, depth, E1);
    const int gate_dtype = d.gate_dtype;
    struct Tag { Tag(int level) { ggpm_timing_tag(3 - level); } ~Tag() { ggpm_timing_tag(0); } } tag(level);
    // no input gradient wanted (the atom level: one-hot inputs): the summed gate-input gradients are not needed on this
    // stream at all -- the depth launches skip their read-modify-write and the second stream sums the stashed gate
    // gradients before it contracts them (GGPM_SKIP_XSUM=0: per-depth accumulation everywhere)
    static const bool xsum_env = !(ggpm_dev_env("GGPM_SKIP_XSUM") && atoi(ggpm_dev_env("GGPM_SKIP_XSUM")) == 0);
    const bool skip_xsum = xsum_env && dx == nullptr && depth > 1;
    if (d.lstm) {
        const float* W[4] = {P[lq(level, Q_WI)], P[lq(level, Q_WOG)], P[lq(level, Q_WU)], P[lq(level, Q_WF)]};
        float* dW[4] = {G[lq(level, Q_WI)], G[lq(level, Q_WOG)], G[lq(level, Q_WU)], G[lq(level, Q_WF)]};
        float* db[4] = {G[lq(level, Q_BI)], G[lq(level, Q_BOG)], G[lq(level, Q_BU)], G[lq(level, Q_BF)]};
        ggpm_backward_lo_depth(blo);
        if (skip_xsum) ggpm_backward_skip_x_sums(1);
        CK(ggpm_lstm_backward(E1, H, depth, L.X + 3 * slot, W[0] + I, I + H, W[1] + I, I + H, W[2] + I, I + H, W[3] + I, I + H,
                              pred.rowptr, pred.col, pred.rowptrT, pred.colT, L.Hs, L.Cs, L.Qs, L.St, L.St + ds,
                              L.St + 2 * ds, L.St + 3 * ds, L.St + 4 * ds, dHD, dX, dX + slot, dX + 2 * slot, dX + 3 * slot,
                              dW[0] + I, I + H, dW[1] + I, I + H, dW[2] + I, I + H, dW[3] + I, I + H, level_work,
                              w.level_work_bytes, 0, st.main));
        if (dx) {       // dx = sum over the gates of dX_k W_k[:, :I]: one launch over four K segments
            const float* A[4] = {dX, dX + slot, dX + 2 * slot, dX + 3 * slot};
            const int lda[4] = {Hp, Hp, Hp, Hp}, ldb[4] = {I + H, I + H, I + H, I + H}, K[4] = {H, H, H, H};
            CK(ggpm_gemm_ksegments(0, E1, I, 4, A, lda, W, ldb, K, dx, lddx, lddx, nullptr, 0, GGPM_ACT_NONE, 0, st.main));
        }
        CK(st.side_after_main());
        const ggpm_stream_t sw = st.w();
        const BwdWork wc = w;
        float* const dW0 = dW[0]; float* const dW1 = dW[1]; float* const dW2 = dW[2]; float* const dW3 = dW[3];
        float* const db0 = db[0]; float* const db1 = db[1]; float* const db2 = db[2]; float* const db3 = db[3];
        const float* Hs = L.Hs; const float* St = L.St;
        // input halves + bias sums from the summed gate-input gradients
        auto x_part = [=](ggpm_stream_t sx) -> int {
            float* const dWk[4] = {dW0, dW1, dW2, dW3};
            float* const dbk[4] = {db0, db1, db2, db3};
            if (skip_xsum) {
                float *DI = nullptr, *DO = nullptr, *DU = nullptr;
                CK(ggpm_lstm_backward_stashes(level_work, E1, H, depth, &DI, &DO, &DU));
                const int lo_ = blo < 1 ? 1 : blo;      // backward steps depth .. lo ran: stash slots lo-1 .. depth-1
                float* const src[3] = {DI, DO, DU};
                const bool b16 = bf16_stored(d, E1);    // (then lo_ == 1 and the stashes are bf16 in the first half of their buffers)
                for (int k = 0; k < 3; ++k)
                    CK(ggpm_sum_slots_any(src[k] + (b16 ? 0 : (size_t)(lo_ - 1) * slot), depth - lo_ + 1, slot, dX + (size_t)k * slot,
                                          b16, sx));
            }
            // (the light column sums first, the products last: see the GRU branch)
            for (int k = 0; k < 4; ++k) CK(ggpm_colsum(dX + k * slot, Hp, E1, H, dbk[k], wc.xcs[level], sx));
            if (ggpm_gemm_prefers_grouped(H, I, E1, 4)) {      // the four in one launch
                GgpmGemmProblem gp[4];
                for (int k = 0; k < 4; ++k) gp[k] = {dX + k * slot, Hp, x, ldx, dWk[k], I + H, I, nullptr, 0, GGPM_ACT_NONE, 0};
                CK(ggpm_gemm_grouped_splitk(1, 0, H, I, E1, 4, gp, wc.xws[level], wc.xws_bytes, sx));
            } else {
                for (int k = 0; k < 4; ++k)
                    CK(ggpm_gemm(1, 0, H, I, E1, dX + k * slot, Hp, x, ldx, dWk[k], I + H, I, nullptr, 0, GGPM_ACT_NONE, 0,
                                 wc.skws, wc.skws_bytes, sx));
            }
            return GGPM_OK;
        };
        // a level with nothing behind it on this stream (the atom level: no input gradient): its input halves run HERE,
        // beside the tall contractions on the second stream, instead of behind them
        const bool x_on_main = skip_xsum && st.side != nullptr && split_tail_enabled();
        const int rc_side = st.on_side([=]() -> int {
            float* const dWk[4] = {dW0, dW1, dW2, dW3};
            if (!x_on_main) CK(x_part(sw));
            ggpm_wgrad_lo_depth(blo);
            {       // (thread-local like the hint above: this body may run on the side worker's thread)
                GateDtypeScope tall_dtype(gate_dtype);
                CK(ggpm_lstm_weight_grads(E1, H, depth, Hs, St, level_work, wc.level_work_bytes, dWk[0] + I, I + H, dWk[1] + I,
                                          I + H, dWk[2] + I, I + H, dWk[3] + I, I + H, sw));
            }
            return GGPM_OK;
        });
        if (rc_side) return rc_side;
        return x_on_main ? x_part(st.main) : GGPM_OK;
    }
    const float *Wz = P[lp(level, L_WZ)], *Wr = P[lp(level, L_WR)], *Wh = P[lp(level, L_WH)];
    float *dWz = G[lp(level, L_WZ)], *dWr = G[lp(level, L_WR)], *dWh = G[lp(level, L_WH)], *dUr = G[lp(level, L_UR)];
    ggpm_backward_lo_depth(blo);
    if (skip_xsum) ggpm_backward_skip_x_sums(1);
    CK(ggpm_gru_backward(E1, H, depth, L.X + slot, Wz + I, I + H, P[lp(level, L_UR)], H, Wh + I, I + H,
                             pred.rowptr, pred.col, pred.rowptrT, pred.colT, L.Hs,
                             L.Qs, L.St, L.St + ds, L.St + 2 * ds, L.St + 3 * ds, L.St + 4 * ds, dHD, dX, dX + slot,
                             dX + 2 * slot, dWz + I, I + H, dUr, H, G[lp(level, L_BU)], dWh + I, I + H, level_work,
                             w.level_work_bytes, 0, st.main));
    if (dx) {       // needed upstream right away: main stream
        // dx = dX_z W_z[:, :I] + dX_r W_r + dX_h W_h[:, :I]: one launch over three K segments
        const float* A[3] = {dX, dX + slot, dX + 2 * slot};
        const float* B[3] = {Wz, Wr, Wh};
        const int lda[3] = {Hp, Hp, Hp}, ldb[3] = {I + H, I, I + H}, K[3] = {H, H, H};
        CK(ggpm_gemm_ksegments(0, E1, I, 3, A, lda, B, ldb, K, dx, lddx, lddx, nullptr, 0, GGPM_ACT_NONE, 0, st.main));
    }
    CK(st.side_after_main());
    const ggpm_stream_t sw = st.w();
    const BwdWork wc = w;
    const float* Hs = L.Hs; const float* St = L.St;
    float* const dbu = G[lp(level, L_BU)]; float* const dbz = G[lp(level, L_BZ)]; float* const dbh = G[lp(level, L_BH)];
    // a level with nothing behind it on this stream (the atom level: no input gradient): its input halves run HERE, beside
    // the tall contractions on the second stream, instead of behind them
    const bool x_on_main = skip_xsum && st.side != nullptr && split_tail_enabled();
    const bool bu_here = x_on_main;
    auto x_part = [=](ggpm_stream_t sx) -> int {       // input halves + bias sums from the summed gate-input gradients
        if (skip_xsum) {
            float *DMP = nullptr, *DZP = nullptr;
            CK(ggpm_gru_backward_stashes(level_work, E1, H, depth, &DMP, &DZP));
            const int lo_ = blo < 1 ? 1 : blo;          // backward steps depth .. lo ran: stash slots lo-1 .. depth-1
            const bool b16 = bf16_stored(d, E1);        // (then lo_ == 1 and the stashes are bf16 in the first half of their buffers)
            CK(ggpm_sum_slots_any(DZP + (b16 ? 0 : (size_t)(lo_ - 1) * slot), depth - lo_ + 1, slot, dX, b16, sx));
            CK(ggpm_sum_slots_any(DMP + (b16 ? 0 : (size_t)(lo_ - 1) * slot), depth - lo_ + 1, slot, dX + 2 * slot, b16, sx));
        }
        // the light column sums first, the products last: beside the tall contractions of the second stream (one 129 KB
        // workgroup on every CU) a 600-workgroup product takes 145 us instead of 20, and whatever follows it on this stream
        // ends the step -- so what follows it is nothing
        CK(ggpm_colsum(dX, Hp, E1, H, dbz, wc.xcs[level], sx));
        CK(ggpm_colsum(dX + 2 * slot, Hp, E1, H, dbh, wc.xcs[level], sx));
        // db_u here too when this part has the main stream to itself: the second stream then ends with the tall contraction
        if (bu_here) CK(ggpm_gru_bias_u_grad(E1, H, depth, blo, level_work, dbu, wc.xcs[level], sx));
        if (ggpm_gemm_prefers_grouped(H, I, E1, 3)) {      // the three in one launch
            const GgpmGemmProblem gp[3] = {{dX, Hp, x, ldx, dWz, I + H, I, nullptr, 0, GGPM_ACT_NONE, 0},
                                           {dX + slot, Hp, x, ldx, dWr, I, I, nullptr, 0, GGPM_ACT_NONE, 0},
                                           {dX + 2 * slot, Hp, x, ldx, dWh, I + H, I, nullptr, 0, GGPM_ACT_NONE, 0}};
            CK(ggpm_gemm_grouped_splitk(1, 0, H, I, E1, 3, gp, wc.xws[level], wc.xws_bytes, sx));
        } else {
            CK(ggpm_gemm(1, 0, H, I, E1, dX, Hp, x, ldx, dWz, I + H, I, nullptr, 0, GGPM_ACT_NONE, 0, wc.skws, wc.skws_bytes, sx));
            CK(ggpm_gemm(1, 0, H, I, E1, dX + slot, Hp, x, ldx, dWr, I, I, nullptr, 0, GGPM_ACT_NONE, 0, wc.skws, wc.skws_bytes,
                         sx));
            CK(ggpm_gemm(1, 0, H, I, E1, dX + 2 * slot, Hp, x, ldx, dWh, I + H, I, nullptr, 0, GGPM_ACT_NONE, 0, wc.skws,
                         wc.skws_bytes, sx));
        }
        return GGPM_OK;
    };
    const int rc_side = st.on_side([=]() -> int {
        if (!x_on_main) CK(x_part(sw));
        ggpm_wgrad_lo_depth(blo);
        ggpm_wgrad_skip_bias_u(bu_here ? 1 : 0);
        GateDtypeScope tall_dtype(gate_dtype);      // (thread-local: this body may run on the side worker's thread)
        CK(ggpm_gru_weight_grads(E1, H, depth, Hs, St, St + ds, level_work, wc.level_work_bytes, dWz + I, I + H, dUr, H,
                                 dbu, dWh + I, I + H, sw));
        return GGPM_OK;
    });
    if (rc_side) return rc_side;
    return x_on_main ? x_part(st.main) : GGPM_OK;
}

}  // namespace

extern "C" size_t ggpm_encoder_work_bytes(const ggpm_enc_dims* dims) {
    if (!dims) return 0;
    const Dims d = make_dims(dims);
    Arena A = {nullptr, 0, false, 0};
    BwdWork w;
    layout_work(A, d, w);
    return A.off;
}

// grads: one buffer per parameter slot (same shapes as the parameters, every element written).  d_* may be null
// (no gradient arrives for that output).  On return the main stream is ordered behind the second stream.
// phase 0: the whole backward.  phase 1: everything except the atom level (whose parameters come last in the slot
// order), phase 2: the atom level + the final join -- the caller may start reducing the gradients of slots
// [0, first graph_encoder slot) across ranks on the second stream between the two calls.
extern "C" int ggpm_encoder_backward(const ggpm_enc_dims* dims, float* const* params, float* const* grads,
                                     const int32_t* roots, void* saved, size_t saved_bytes, const float* hroot,
                                     const float* hnode, const float* hinter, const float* hatom, const float* d_hroot,
                                     const float* d_hnode, const float* d_hinter, const float* d_hatom, void* work,
                                     size_t work_bytes, int phase, ggpm_stream_t stream, ggpm_stream_t side_stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!dims || !params || !grads || !roots || !saved || !work || !hroot || !hnode || !hinter || !hatom || phase < 0 ||
        phase > 2)
        return GGPM_ERR_ARG;
    const Dims d = make_dims(dims);
    GateDtypeScope gate_scope(d.gate_dtype);
    Arena A = {reinterpret_cast<char*>(saved), 0, false, saved_bytes};
    Saved S;
    layout_saved(A, d, S);
    Arena B = {reinterpret_cast<char*>(work), 0, false, work_bytes};
    BwdWork w;
    layout_work(B, d, w);
    if (A.overflow || B.overflow) return GGPM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* const* P = params;
    float* const* G = grads;
    const int H = d.H, Hp = d.Hp, He = d.He;
    Streams st = {stream, side_stream, 0, side_stream ? SideWorker::get() : nullptr};
    DrainGuard guard = {st.wk};
    if (st.wk) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        st.wk->use_device(dev);
    }
    const int Nmax = d.N1g > d.N1t ? d.N1g : d.N1t, Emax = d.E1g > d.E1t ? d.E1g : d.E1t;
    float* dpre[5];
    for (int i = 0; i < 5; ++i) dpre[i] = w.dpre + (size_t)i * Nmax * Hp;
    float* dXl[3];
    for (int i = 0; i < 3; ++i) dXl[i] = w.dX + (size_t)i * 4 * Emax * Hp;
    float* lwork[3];
    for (int i = 0; i < 3; ++i) lwork[i] = w.level_work + (size_t)i * (w.level_work_bytes / 4 + 64);
    const size_t nt = (size_t)d.N1t * Hp * sizeof(float), ng = (size_t)d.N1g * Hp * sizeof(float);

    if (side_stream && phase != 2) {      // the transposed CSRs were built on the second stream during the forward
        hipEvent_t ev = ggpm_wgrad_event(61);
        if (!ev) return GGPM_ERR_LAUNCH;
        (void)hipEventRecord(ev, (hipStream_t)side_stream);
        (void)hipStreamWaitEvent(s, ev, 0);
    }
    if (phase != 2) {
    // ---- root readout
    if (d_hroot) {
        CK(ggpm_act_backward(d_hroot, hroot, d.B, H, Hp, GGPM_ACT_TANH, 0, w.dpre_root, stream));
        CK(linear2_dx(d.B, H, H, w.dpre_root, Hp, P[P_WROOT], w.df, Hp, 0, w.dn, Hp, 0, stream));
        CK(ggpm_segment_sum(w.df, Hp, S.root.rowptrT, S.root.colT, d.N1t, H, w.d_hnode_t, Hp, 0, Hp, stream));
        CK(ggpm_segment_sum(w.dn, Hp, S.root.rowptrT, S.root.colT, d.N1t, H, w.d_nei_t, Hp, 0, Hp, stream));
        CK(linear2_wgrad(d.B, H, w.dpre_root, Hp, S.f, Hp, H, S.n, Hp, H, G[P_WROOT], G[P_BROOT], w, st));
    } else {
        (void)hipMemsetAsync(w.d_hnode_t, 0, nt, s);
        (void)hipMemsetAsync(w.d_nei_t, 0, nt, s);
        float* const gw = G[P_WROOT]; float* const gb = G[P_BROOT];
        const ggpm_stream_t sw = st.w();
        CK(st.on_side([=]() -> int {
            (void)hipMemsetAsync(gw, 0, (size_t)H * 2 * H * sizeof(float), (hipStream_t)sw);
            (void)hipMemsetAsync(gb, 0, (size_t)H * sizeof(float), (hipStream_t)sw);
            return GGPM_OK;
        }));
    }

    // ---- motif level: W_o, message function, W_c / E_c
    if (d_hnode) {
        CK(ggpm_act_backward(d_hnode, hnode, d.N1t, H, Hp, GGPM_ACT_RELU, 1, dpre[0], stream));
        CK(drop(d, dpre[0], d.N1t, H, Hp, DS_WO_TREE, stream));     // d(dropout) commutes with the ReLU mask
        CK(linear2_dx(d.N1t, H, H, dpre[0], Hp, P[lwo(d.lstm, 0)], w.d_hnode_t, Hp, 1, w.d_nei_t, Hp, 1, stream));
        CK(linear2_wgrad(d.N1t, H, dpre[0], Hp, S.hnode_t, Hp, H, S.lv[0].nei, Hp, H, G[lwo(d.lstm, 0)], G[lbo(d.lstm, 0)], w, st));
    } else {
        float* const gw = G[lwo(d.lstm, 0)]; float* const gb = G[lbo(d.lstm, 0)];
        const ggpm_stream_t sw = st.w();
        CK(st.on_side([=]() -> int {
            (void)hipMemsetAsync(gw, 0, (size_t)H * 2 * H * sizeof(float), (hipStream_t)sw);
            (void)hipMemsetAsync(gb, 0, (size_t)H * sizeof(float), (hipStream_t)sw);
            return GGPM_OK;
        }));
    }
    CK(ggpm_segment_sum(w.d_nei_t, Hp, S.tagr.rowptrT, S.tagr.colT, d.E1t, H, w.d_h, Hp, 0, Hp, stream));
    CK(level_backward(d, d.E1t, d.It, d.depthT, S.hmess_t, d.ld_t, P, G, 0, S.tpred, S.lv[0], w.d_h, dXl[0], lwork[0],
                      w.dx_mess, d.ld_t, w, st));
    CK(ggpm_segment_sum(w.dx_mess, d.ld_t, S.tsrc.rowptrT, S.tsrc.colT, d.N1t, H, w.d_hnode_t, Hp, 1, 0, stream));
    CK(ggpm_act_backward(w.d_hnode_t, S.hnode_t, d.N1t, H, Hp, GGPM_ACT_RELU, 0, dpre[1], stream));
    CK(drop(d, dpre[1], d.N1t, H, Hp, DS_WC, stream));
    if (d_hinter) (void)hipMemcpyAsync(w.d_hinter, d_hinter, nt, hipMemcpyDeviceToDevice, s);
    CK(linear2_dx(d.N1t, H, He, dpre[1], Hp, P[P_WC], w.d_finput, d.Hep, 0, w.d_hinter, Hp, d_hinter ? 1 : 0, stream));
    CK(drop(d, w.d_finput, d.N1t, He, d.Hep, DS_EC, stream));
    CK(linear2_wgrad(d.N1t, H, dpre[1], Hp, S.finput_t, d.Hep, He, hinter, Hp, H, G[P_WC], G[P_BC], w, st));
    {
        const float* src = w.d_finput; const int32_t* rp = S.motif.rowptrT; const int32_t* cl = S.motif.colT;
        float* const g = G[P_EC]; const ggpm_stream_t sw = st.w(); const int Hep = d.Hep, rows = d.n_motif;
        CK(st.on_side([=]() -> int { return ggpm_segment_sum(src, Hep, rp, cl, rows, He, g, He, 0, He, sw); }));
    }

    // ---- attachment level: W_o, message function, W_i / E_i, pooling over atoms
    CK(ggpm_act_backward(w.d_hinter, hinter, d.N1t, H, Hp, GGPM_ACT_RELU, 1, dpre[2], stream));
    CK(drop(d, dpre[2], d.N1t, H, Hp, DS_WO_INTER, stream));
    CK(linear2_dx(d.N1t, H, H, dpre[2], Hp, P[lwo(d.lstm, 1)], w.d_hnode_i, Hp, 0, w.d_nei_i, Hp, 0, stream));
    CK(linear2_wgrad(d.N1t, H, dpre[2], Hp, S.hnode_i, Hp, H, S.lv[1].nei, Hp, H, G[lwo(d.lstm, 1)], G[lbo(d.lstm, 1)], w, st));
    CK(ggpm_segment_sum(w.d_nei_i, Hp, S.tagr.rowptrT, S.tagr.colT, d.E1t, H, w.d_h, Hp, 0, Hp, stream));
    CK(level_backward(d, d.E1t, d.It, d.depthT, S.hmess_i, d.ld_t, P, G, 1, S.tpred, S.lv[1], w.d_h, dXl[1], lwork[1],
                      w.dx_mess, d.ld_t, w, st));
    CK(ggpm_segment_sum(w.dx_mess, d.ld_t, S.tsrc.rowptrT, S.tsrc.colT, d.N1t, H, w.d_hnode_i, Hp, 1, 0, stream));
    CK(ggpm_act_backward(w.d_hnode_i, S.hnode_i, d.N1t, H, Hp, GGPM_ACT_RELU, 0, dpre[3], stream));
    CK(drop(d, dpre[3], d.N1t, H, Hp, DS_WI, stream));
    float* d_finput_i = w.d_finput + (size_t)d.N1t * d.Hep;
    CK(linear2_dx(d.N1t, H, He, dpre[3], Hp, P[P_WI], d_finput_i, d.Hep, 0, w.d_pooled, Hp, 0, stream));
    CK(drop(d, d_finput_i, d.N1t, He, d.Hep, DS_EI, stream));
    CK(linear2_wgrad(d.N1t, H, dpre[3], Hp, S.finput_i, d.Hep, He, S.pooled, Hp, H, G[P_WI], G[P_BI], w, st));
    {
        const int32_t* rp = S.attach.rowptrT; const int32_t* cl = S.attach.colT;
        float* const g = G[P_EI]; const ggpm_stream_t sw = st.w(); const int Hep = d.Hep, rows = d.n_attach;
        CK(st.on_side([=]() -> int { return ggpm_segment_sum(d_finput_i, Hep, rp, cl, rows, He, g, He, 0, He, sw); }));
    }
    if (d_hatom) (void)hipMemcpyAsync(w.d_hatom, d_hatom, ng, hipMemcpyDeviceToDevice, s);
    CK(ggpm_segment_sum(w.d_pooled, Hp, S.tcgr.rowptrT, S.tcgr.colT, d.N1g, H, w.d_hatom, Hp, d_hatom ? 1 : 0,
                        d_hatom ? 0 : Hp, stream));

    }   // phase != 2
    if (phase == 1) {      // everything but the atom level's parameter gradients is issued (second stream: in flight)
        CK(st.drain());
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }

    // ---- atom level: W_o, message function (its inputs are constants)
    CK(ggpm_act_backward(w.d_hatom, hatom, d.N1g, H, Hp, GGPM_ACT_RELU, 1, dpre[4], stream));
    CK(drop(d, dpre[4], d.N1g, H, Hp, DS_WO_ATOM, stream));
    CK(ggpm_gemm(0, 0, d.N1g, H, H, dpre[4], Hp, P[lwo(d.lstm, 2)] + d.atom, d.atom + H, w.d_nei_g, Hp, Hp, nullptr, 0,
                 GGPM_ACT_NONE, 0, nullptr, 0, stream));
    CK(linear2_wgrad(d.N1g, H, dpre[4], Hp, S.hnode_a, d.ld_n, d.atom, S.lv[2].nei, Hp, H, G[lwo(d.lstm, 2)], G[lbo(d.lstm, 2)], w,
                     st));
    CK(ggpm_segment_sum(w.d_nei_g, Hp, S.gagr.rowptrT, S.gagr.colT, d.E1g, H, w.d_h, Hp, 0, Hp, stream));
    CK(level_backward(d, d.E1g, d.Ig, d.depthG, S.hmess_a, d.ld_m, P, G, 2, S.gpred, S.lv[2], w.d_h, dXl[2], lwork[2],
                      nullptr, 0, w, st));

    CK(st.drain());
    if (side_stream) {      // every gradient buffer is complete once the main stream has passed this point
        hipEvent_t ev = ggpm_wgrad_event(62);
        if (!ev) return GGPM_ERR_LAUNCH;
        (void)hipEventRecord(ev, (hipStream_t)side_stream);
        (void)hipStreamWaitEvent(s, ev, 0);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
