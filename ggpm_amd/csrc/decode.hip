// Step loop of the teacher-forced decoder's atom level as ONE C call per direction.
//
// Reference: HierMPNDecoder.forward calls IncHierMPNEncoder.forward once per decode step (ggpm/decoder.py:201-222); its
// atom-level block (ggpm/encoder.py:235-239 -> IncMPNEncoder.forward, :165-179) recomputes the bond messages of the
// step's motif with `diterG` iterations of sparse_forward (ggpm/rnn.py:52-59, 110-121).  ggpm_amd/atom_decode.py runs
// every step on its compact row set (the step's bonds + the frozen older bonds they read) out of stacked state / stash
// buffers; what is left inside the loop is `gather the frozen rows -> sparse_forward` and, backwards,
// `sparse_backward -> scatter-add to the rows' producers`.  These two drivers issue exactly those calls, in the same
// order, from C++: ~25 launches per step and direction that Python + ctypes otherwise issue one by one (the full VAE step
// is as long on the host as on the GPU).
#include "common.h"
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// GGPM_DECODE_FOLD=0 (dev A/B, tests): the start-state gather and the incoming-gradient scatter of every decode step as
// launches of their own instead of inside the step's first / last launch
inline bool decode_fold() { static const bool v = [] { const char* e = ggpm_dev_env("GGPM_DECODE_FOLD"); return !e || atoi(e) != 0; }(); return v; }

struct Offs {
    size_t f0, f1, r0, q0;      // F-id range of the step, first row of its [depth][n] and [depth + 1][n] blocks
};

inline Offs offs(const ggpm_decode_steps* d, int t) {
    return {(size_t)d->foff[t], (size_t)d->foff[t + 1], (size_t)d->roff[t], (size_t)d->qoff[t]};
}

inline bool steps_ok(const ggpm_decode_steps* d) {
    return d && d->T > 0 && d->H > 0 && d->depth > 0 && d->n && d->foff && d->roff && d->qoff && d->srcH && d->srcF &&
           d->frozen && d->pred_rowptr && d->pred_col && d->succ_rowptr && d->succ_col;
}

}  // namespace

extern "C" int ggpm_decode_steps_forward(const ggpm_decode_steps* d, const float* const* W, const int* ldw, const float* bu,
                                         const float* X_all, float* Hs_all, float* Cs_all, float* Qs_all, float* St_all,
                                         size_t st_stride, float* wpack, float* tmp, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!steps_ok(d) || !W || !ldw || !X_all || !Hs_all || !Qs_all || !St_all || !wpack || !tmp || (d->lstm && !Cs_all) ||
        (!d->lstm && !bu))
        return GGPM_ERR_ARG;
    const int H = d->H, Hp = ggpm_padded_hidden(H), G = d->lstm ? 4 : 3;
    int nmax = 0;
    for (int t = 0; t < d->T; ++t) nmax = d->n[t] > nmax ? d->n[t] : nmax;
    (void)tmp;
    const bool fold = decode_fold();
    for (int t = 0; t < d->T; ++t) {
        const int n = d->n[t];
        const Offs o = offs(d, t);
        const float* x = X_all + (size_t)G * o.f0 * Hp;
        const size_t xs = (size_t)n * Hp;
        float* hs = Hs_all + o.q0 * Hp;
        float* qs = Qs_all + o.r0 * Hp;
        float* st[5];
        for (int k = 0; k < 5; ++k) st[k] = St_all + (size_t)k * st_stride + o.r0 * Hp;
        // the step's start state goes straight into slot 0 of its block: frozen rows from the blocks of the steps that
        // produced them, zero for the rows the step recomputes (srcH = -1) -- exactly the masked state sparse_forward would
        // build.  The step's first launch (q^0 = U_r h^0 / qf^0 = Wf_h h^0) fetches the rows through srcH and writes slot 0
        // on the way (ggpm_forward_gather_state); GGPM_DECODE_FOLD=0: a gather launch of its own, as before.
        int rc = GGPM_OK;
        if (fold) {
            ggpm_forward_gather_state(Hs_all, d->lstm ? Cs_all : nullptr, d->srcH[t]);
        } else {
            rc = ggpm_gather_rows(Hs_all, Hp, d->srcH[t], n, Hp, hs, Hp, 0, 0, stream);
            if (rc) return rc;
        }
        if (t > 0) ggpm_weights_packed(1);      // same weights, same `wpack`: packed by the first step
        if (d->lstm) {
            float* cs = Cs_all + o.q0 * Hp;
            if (!fold) {
                rc = ggpm_gather_rows(Cs_all, Hp, d->srcH[t], n, Hp, cs, Hp, 0, 0, stream);
                if (rc) return rc;
            }
            rc = ggpm_lstm_sparse_forward(n, H, d->depth, hs, cs, d->frozen[t], x, x + xs, x + 2 * xs, x + 3 * xs, W[0],
                                          ldw[0], W[1], ldw[1], W[2], ldw[2], W[3], ldw[3], d->pred_rowptr[t], d->pred_col[t],
                                          hs, cs, qs, st[0], st[1], st[2], st[3], st[4], wpack, 1, stream);
        } else {
            rc = ggpm_gru_sparse_forward(n, H, d->depth, hs, d->frozen[t], x, x + xs, x + 2 * xs, W[0], ldw[0], W[1], ldw[1],
                                         bu, W[2], ldw[2], d->pred_rowptr[t], d->pred_col[t], hs, qs, st[0], st[1], st[2],
                                         st[3], st[4], wpack, 1, stream);
        }
        if (rc) return rc;
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_decode_steps_backward(const ggpm_decode_steps* d, const float* const* W, const int* ldw,
                                          const float* X_all, const float* Hs_all, const float* Cs_all, const float* Qs_all,
                                          const float* St_all, size_t st_stride, float* dF, float* dCF, float* dX_all,
                                          float* DG_all, size_t dg_stride, float* DQ_all, float* const* dW_unused,
                                          float* work, size_t work_bytes, float* tmp, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!steps_ok(d) || !W || !ldw || !X_all || !Hs_all || !Qs_all || !St_all || !dF || !dX_all || !DG_all || !DQ_all ||
        !dW_unused || !work || !tmp || (d->lstm && (!Cs_all || !dCF)))
        return GGPM_ERR_ARG;
    const int H = d->H, Hp = ggpm_padded_hidden(H), G = d->lstm ? 4 : 3;
    int nmax = 0;
    for (int t = 0; t < d->T; ++t) nmax = d->n[t] > nmax ? d->n[t] : nmax;
    float* dhin = tmp;
    float* dcin = tmp + (size_t)nmax * Hp;
    const bool fold = decode_fold();
    for (int t = d->T - 1; t >= 0; --t) {
        const int n = d->n[t];
        const Offs o = offs(d, t);
        const size_t xs = (size_t)n * Hp;
        const float* x = X_all + (size_t)G * o.f0 * Hp;
        float* dx = dX_all + (size_t)G * o.f0 * Hp;
        const float* hs = Hs_all + o.q0 * Hp;
        const float* qs = Qs_all + o.r0 * Hp;
        const float* st[5];
        for (int k = 0; k < 5; ++k) st[k] = St_all + (size_t)k * st_stride + o.r0 * Hp;
        float* dg[3];
        for (int k = 0; k < 3; ++k) dg[k] = DG_all + (size_t)k * dg_stride + o.r0 * Hp;
        float* dhd = dF + o.f0 * Hp;
        if (t < d->T - 1) ggpm_weights_packed(1);      // same weights, same `work`: the transposes were packed by the first call
        int rc;
        // the frozen rows' gradient goes to the step that produced their state (rows recomputed here: none): added there by
        // the step's last launch (ggpm_backward_scatter_state), or by scatter launches of their own
        if (fold) ggpm_backward_scatter_state(dF, d->lstm ? dCF : nullptr, d->srcF[t]);
        if (d->lstm) {
            ggpm_backward_defer_stash(dg[0], dg[1], dg[2], DQ_all + o.q0 * Hp);
            rc = ggpm_lstm_sparse_backward(n, H, d->depth, d->frozen[t], x + 3 * xs, W[0], ldw[0], W[1], ldw[1], W[2], ldw[2],
                                           W[3], ldw[3], d->pred_rowptr[t], d->pred_col[t], d->succ_rowptr[t],
                                           d->succ_col[t], hs, Cs_all + o.q0 * Hp, qs, st[0], st[1], st[2], st[3], st[4], dhd,
                                           dCF + o.f0 * Hp, dhin, dcin, dx, dx + xs, dx + 2 * xs, dx + 3 * xs, dW_unused[0],
                                           H, dW_unused[1], H, dW_unused[2], H, dW_unused[3], H, work, work_bytes, stream);
            if (rc) return rc;
            if (!fold) rc = ggpm_scatter_rows(dcin, Hp, d->srcF[t], n, Hp, dCF, Hp, 1, stream);
        } else {
            ggpm_backward_defer_stash(dg[0], dg[1], DQ_all + o.q0 * Hp, nullptr);
            rc = ggpm_gru_sparse_backward(n, H, d->depth, d->frozen[t], x + xs, W[0], ldw[0], W[1], ldw[1], W[2], ldw[2],
                                          d->pred_rowptr[t], d->pred_col[t], d->succ_rowptr[t], d->succ_col[t], hs, qs, st[0],
                                          st[1], st[2], st[3], st[4], dhd, dhin, dx, dx + xs, dx + 2 * xs, dW_unused[0], H,
                                          dW_unused[1], H, dW_unused[3], dW_unused[2], H, work, work_bytes, stream);
        }
        if (rc) return rc;
        if (!fold) rc = ggpm_scatter_rows(dhin, Hp, d->srcF[t], n, Hp, dF, Hp, 1, stream);
        if (rc) return rc;
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}


// ---- issuing the step loops from a worker thread ------------------------------------------------------------------------
// A step loop is ~280 launches = 1.5-1.8 ms of host time inside ONE call.  While the calling thread sits in it, nothing else
// of the training step can be issued: in the backward pass the autograd engine reaches the atom level's node and the
// encoder's node at about the same time, and whichever it takes second starts that much later on the GPU although the two
// chains do not depend on each other ("the two chains have to be ISSUED side by side").  The _async
// forms hand the loop to a thread of the library and return at once; ggpm_decode_join waits until that thread has issued
// everything (it does not wait for the GPU).  The caller keeps every buffer and the descriptor alive until then and
// enqueues what follows the loop on the same stream only after the join.  One worker, jobs in order.
namespace {

class DecodeWorker {
public:
    static DecodeWorker& get() { static DecodeWorker w; return w; }
    void post(int dev, std::function<int()> job) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back({dev, std::move(job)}); }
        cv_.notify_one();
    }
    int join() {
        std::unique_lock<std::mutex> lk(mu_);
        idle_.wait(lk, [&] { return q_.empty() && !busy_; });
        const int e = err_;
        err_ = GGPM_OK;
        return e;
    }

private:
    DecodeWorker() : th_([this] { run(); }) {}
    ~DecodeWorker() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_one();
        if (th_.joinable()) th_.join();
    }
    void run() {
        int dev = -1;
        for (;;) {
            std::pair<int, std::function<int()>> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
                busy_ = true;
            }
            if (job.first != dev && job.first >= 0) { (void)hipSetDevice(job.first); dev = job.first; }
            int rc = job.second();
            if (!rc && hipGetLastError() != hipSuccess) rc = GGPM_ERR_LAUNCH;
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
    std::deque<std::pair<int, std::function<int()>>> q_;
    bool busy_ = false, stop_ = false;
    int err_ = GGPM_OK;
    std::thread th_;
};

inline int current_device() { int d = -1; (void)hipGetDevice(&d); return d; }

}  // namespace

extern "C" int ggpm_decode_steps_forward_async(const ggpm_decode_steps* d, const float* const* W, const int* ldw, const float* bu,
                                               const float* X_all, float* Hs_all, float* Cs_all, float* Qs_all, float* St_all,
                                               size_t st_stride, float* wpack, float* tmp, ggpm_stream_t stream) {
    if (!d || !W || !ldw) return GGPM_ERR_ARG;
    const float* Wc[4] = {W[0], W[1], W[2], W[3]};
    const int lc[4] = {ldw[0], ldw[1], ldw[2], ldw[3]};
    DecodeWorker::get().post(current_device(), [=]() -> int {
        return ggpm_decode_steps_forward(d, Wc, lc, bu, X_all, Hs_all, Cs_all, Qs_all, St_all, st_stride, wpack, tmp, stream);
    });
    return GGPM_OK;
}

extern "C" int ggpm_decode_steps_backward_async(const ggpm_decode_steps* d, const float* const* W, const int* ldw,
                                                const float* X_all, const float* Hs_all, const float* Cs_all,
                                                const float* Qs_all, const float* St_all, size_t st_stride, float* dF,
                                                float* dCF, float* dX_all, float* DG_all, size_t dg_stride, float* DQ_all,
                                                float* const* dW_unused, float* work, size_t work_bytes, float* tmp,
                                                ggpm_stream_t stream) {
    if (!d || !W || !ldw || !dW_unused) return GGPM_ERR_ARG;
    const float* Wc[4] = {W[0], W[1], W[2], W[3]};
    const int lc[4] = {ldw[0], ldw[1], ldw[2], ldw[3]};
    float* dWc[4] = {dW_unused[0], dW_unused[1], dW_unused[2], dW_unused[3]};
    DecodeWorker::get().post(current_device(), [=]() -> int {
        return ggpm_decode_steps_backward(d, Wc, lc, X_all, Hs_all, Cs_all, Qs_all, St_all, st_stride, dF, dCF, dX_all, DG_all,
                                          dg_stride, DQ_all, dWc, work, work_bytes, tmp, stream);
    });
    return GGPM_OK;
}

extern "C" int ggpm_decode_join(void) { return DecodeWorker::get().join(); }
