// Shared device helpers for the gfx950 kernels (wave64, MFMA f32, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "../../include/ggpm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GGPM_WAVE 64

#define GGPM_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return GGPM_ERR_LAUNCH; \
    } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process (PyTorch
// leaves benign errors behind, e.g. from capability probes); clear it on entry so that the check after our
// launches reports OUR launches only.
#define GGPM_CLEAR_STALE_ERROR() (void)hipGetLastError()

// Tuning / ablation switches (tile splits, kernel selections, phases switched off for timing, ...) are read from the
// environment only in the DEV variant of the library -- `python -m ggpm_amd.build --variant dev -DGGPM_DEV_SWITCHES`, selected
// at run time with GGPM_LIB_PATH.  The product build compiles every one of them to its default: it has no environment
// switches of its own.  tools/README.md lists the names.
#ifdef GGPM_DEV_SWITCHES
inline const char* ggpm_dev_env(const char* name) { return getenv(name); }
#else
inline const char* ggpm_dev_env(const char*) { return nullptr; }
#endif

// Raise a kernel's dynamic-LDS limit to at least `bytes`: once per (kernel instantiation, device, size) instead of once per
// launch (the attribute call costs host microseconds; a level issues ~40 launches per direction).  Launches come from several
// host threads at once (autograd's, the library's decode and side workers): the limit is only ever RAISED, under a lock, and
// the size on record is stored after the call succeeded -- a thread can therefore never see a size on record that is larger
// than the attribute in force.
// The record is keyed by the kernel's ADDRESS and the device: every instantiation of a kernel template shares one function-
// pointer type (gru_fwd_a<true,2,1> and gru_fwd_a<true,0,2> are both void(*)(GruFwdArgs)), so a record per TYPE would let
// one instantiation's size stand for another's.  A small open-addressed table (the library has ~150 kernel instantiations);
// a full table only costs the attribute call again.
inline void ggpm_set_lds_addr(const void* kernel, size_t bytes) {
    constexpr int MAXDEV = 16, SLOTS = 1024;
    struct Rec { std::atomic<const void*> key; std::atomic<size_t> have[MAXDEV]; };
    static Rec table[SLOTS];                      // zero-initialised
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) dev = 0;
    size_t h = (reinterpret_cast<uintptr_t>(kernel) >> 4) * 0x9E3779B97F4A7C15ull >> 54;      // 10 bits
    Rec* rec = nullptr;
    for (int probe = 0; probe < SLOTS; ++probe, h = (h + 1) % SLOTS) {
        const void* k = table[h].key.load(std::memory_order_acquire);
        if (k == kernel) { rec = &table[h]; break; }
        if (k == nullptr) {
            const void* expect = nullptr;
            if (table[h].key.compare_exchange_strong(expect, kernel, std::memory_order_acq_rel) || expect == kernel) { rec = &table[h]; break; }
        }
    }
    if (rec && bytes <= rec->have[dev].load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lock(mu);
    if (rec && bytes <= rec->have[dev].load(std::memory_order_relaxed)) return;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) {
        if (rec) rec->have[dev].store(bytes, std::memory_order_release);
    } else {
        (void)hipGetLastError();                   // (the launch that follows reports the failure)
    }
}
template <typename K>
inline void ggpm_set_lds(K kernel, size_t bytes) { ggpm_set_lds_addr(reinterpret_cast<const void*>(kernel), bytes); }

static inline int ggpm_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int ggpm_round_up(int a, int b) { return ggpm_ceil_div(a, b) * b; }

__device__ __forceinline__ float ggpm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float4 ggpm_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void ggpm_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// base (uniform, in scalar registers) + a 32-bit BYTE offset per lane: the backend then addresses with the scalar base and ONE
// offset register for every array that shares the offset, instead of a 64-bit address pair per array
__device__ __forceinline__ float4 ggpm_ld4o(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void ggpm_st4o(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

__device__ __forceinline__ float4 operator+(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 operator*(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 ggpm_sigmoid4(float4 a) {
    return make_float4(ggpm_sigmoid(a.x), ggpm_sigmoid(a.y), ggpm_sigmoid(a.z), ggpm_sigmoid(a.w));
}
__device__ __forceinline__ float4 ggpm_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// The gather phases of the depth kernels evaluate one sigmoid per gathered neighbour element (rnn.py:31-32, 90) and are VALU
// bound with libm expf + IEEE division (1.0 / 2.3 us per atom-level launch, round 4).  Two forms on the hardware units:
// FAST (rounds 1-4; ablation builds only now): v_exp_f32(-x * log2e), v_rcp_f32.  The rounding of the product -x * log2e alone is worth
// |x| * log2e * 2^-24 * ln2 relative error of the exponential (1.4 ulp at |x| = 5, 3 ulp at |x| = 10), then v_exp_f32 and
// v_rcp_f32 add ~1 ulp each: measured mean 1.30 ulp, 25 % of the arguments above 1 ulp, 29.5 ulp at x = -29.8
// (tools/probe/sigmoid_ulp.hip), and a sum-aggregating GRU over 30 depths amplifies it (twice the distance
// to fp64 on the deciding configs[4] tensors).
__device__ __forceinline__ float ggpm_fsigmoid_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }
// ACCURATE (shipped, every gate mode): measured mean 0.42 ulp, 6.9 % above 1 ulp, worst 3.4 ulp -- at |x| ~ 16.7, where 1 + e itself
// rounds -- against 0.40 / 5.3 % / 2.5 for libm expf + IEEE division, in 11 instructions instead of ~35 (and FEWER issue
// slots than the fast form, whose __expf carries denormal-range handling: 2.57 against 3.82 ms for 2^33 evaluations):
//   t = -x * log2e as a two-term product t_hi + t_lo (log2e = L_hi + L_lo, the rounding of the leading product recovered by
//   fma); exp2(t_hi) on v_exp_f32 -- the unit's own range reduction of the ROUNDED argument is exact --, times
//   2^t_lo = 1 + t_lo ln2 (|t_lo| < 2^-17: first order is exact to 2^-35); then 1 / (1 + e) as v_rcp_f32 + one Newton step.
//   |x| is clamped to 88 first so that e stays finite (inf * 0 in the correction would be NaN); sigmoid(+-88) is 1 / 6e-39.
__device__ __forceinline__ float ggpm_fsigmoid_acc(float x) {
    const float L_hi = 1.44269502162933349609375f, L_lo = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    const float nx = -__builtin_amdgcn_fmed3f(x, -88.f, 88.f);
    const float t = nx * L_hi;
    float lo = __builtin_fmaf(nx, L_hi, -t);
    lo = __builtin_fmaf(nx, L_lo, lo);
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, lo * LN2, e);
    const float d = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
// Round 5: the accurate form in EVERY gate mode -- under bf16 gate products, too, it is the faster one (configs[4] bf16 leg, same
// box, alternating: 19.98 -> 19.47 ms per step, gru_fwd_a 129.8 -> 124.4 us, gru_bwd_a 115.8 -> 108.7 us: those gathers are VALU
// bound) and the bf16 oracle tests see fewer flipped roundings with it.  The template argument stays for the ablation builds.
#ifdef GGPM_ABL_EXACT_GATHER_SIGMOID      // ablation build (python -m ggpm_amd.build --variant exactsig -DGGPM_ABL_EXACT_GATHER_SIGMOID):
template <bool BF16> __device__ __forceinline__ float ggpm_fsigmoid(float x) { return ggpm_sigmoid(x); }      // libm expf + IEEE division in the gathers too
#elif defined(GGPM_ABL_FAST_GATHER_SIGMOID)      // ablation build: the round-4 form in every gate mode (timing A/B)
template <bool BF16> __device__ __forceinline__ float ggpm_fsigmoid(float x) { return ggpm_fsigmoid_fast(x); }
#elif defined(GGPM_ABL_FAST_SIGMOID_BF16)        // ablation build: the round-4 form under bf16 gate products only
template <bool BF16> __device__ __forceinline__ float ggpm_fsigmoid(float x) { return BF16 ? ggpm_fsigmoid_fast(x) : ggpm_fsigmoid_acc(x); }
#else
template <bool BF16> __device__ __forceinline__ float ggpm_fsigmoid(float x) { return ggpm_fsigmoid_acc(x); }
#endif
template <bool BF16> __device__ __forceinline__ float4 ggpm_fsigmoid4(float4 a) {
    return make_float4(ggpm_fsigmoid<BF16>(a.x), ggpm_fsigmoid<BF16>(a.y), ggpm_fsigmoid<BF16>(a.z), ggpm_fsigmoid<BF16>(a.w));
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global stores / loads
// (__syncthreads() also drains vmcnt, which puts every store's acknowledgement on the critical path).  A wave
// that published data by LDS-DMA waits for vmcnt itself before calling this.
__device__ __forceinline__ void ggpm_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// true when `count` products of this shape are better issued as ONE unsplit grouped launch (small output, K short enough
// for the in-workgroup K split) than as separate split-K launches
bool ggpm_gemm_prefers_grouped(int M, int N, int K, int count);
// `count` (<= 4) tall contractions C_i = A_i^T B_i of one output shape (K_i may differ) in one launch + one reduce;
// falls back to sequential ggpm_gemm calls when a member does not qualify for the tall kernel
// bf16 == 1: operands rounded to bf16 (RNE) on their way into LDS, fp32 accumulate (gemm_tn_tall_bf16); bf16 == 2: the
// operands ARE bf16 in memory (lda / ldb in elements; the group must qualify for the tall kernel: GGPM_ERR_UNSUPPORTED if not)
int ggpm_gemm_tall_grouped(int M, int N, int count, const ggpm_gemm_problem* p, const int* K, float* ws, size_t ws_bytes,
                           ggpm_stream_t stream, int bf16 = 0);
#define GGPM_GEMM_MAX_GROUP 4          // members of ggpm_gemm_grouped / segments of ggpm_gemm_ksegments
typedef ggpm_gemm_problem GgpmGemmProblem;

// Message passing on a tree reaches a fixed point after as many steps as the longest dependency chain: the next dense
// level forward of this thread (GRU or LSTM) issues only `run_depth` of its `depth` steps (0 / >= depth: all of them);
// the caller then replicates the last computed slot of every stash array (encoder.hip).  Consumed by one call.
void ggpm_forward_run_depth(int run_depth);
int ggpm_take_run_depth();
// The same acyclic structure makes the Jacobian of a tree level's recurrence nilpotent: with a longest dependency chain
// of C messages, d(h^{D-k}) is exactly zero for k >= C, so the backward of such a level only has to run its steps
// t = D .. lo with lo = max(1, D - C + 1); every skipped launch would compute exact zeros and every skipped stash slot
// would add exact zeros to the weight-gradient contractions.  The next dense level backward / weight-gradient call of
// this thread stops at step `lo` (<= 1: all steps).  Consumed by one call each.
void ggpm_backward_lo_depth(int lo);
int ggpm_take_backward_lo();
void ggpm_wgrad_lo_depth(int lo);
int ggpm_take_wgrad_lo();
// The next ggpm_gru_weight_grads of this thread leaves db_u alone: the caller forms it with ggpm_gru_bias_u_grad on another
// stream (the encoder driver's atom level: the column sum over all dq stash rows, 52 us, runs on the main stream beside the
// tall contractions instead of behind them).  `lo` as ggpm_wgrad_lo_depth; `csws`: 256 * Hp floats.
void ggpm_wgrad_skip_bias_u(int yes);
int ggpm_gru_bias_u_grad(int E1, int H, int depth, int lo, float* work, float* dbu, float* csws, ggpm_stream_t stream);
// ggpm_backward_defer_stash (include/ggpm_hip.h): caller-owned gate-gradient stashes for the next sparse backward of this
// thread.  -> true (and the pointers) once.
bool ggpm_take_defer_stash(float* (&out)[4]);
// The next SPARSE backward of this thread (GRU or LSTM) leaves the hidden-half weight gradients to a later call of
// ggpm_gru_sparse_weight_grads / ggpm_lstm_sparse_weight_grads with the same arguments -- the same launches, on whatever stream
// that call names (tree_level.hip: beside the rest of the level's backward instead of in front of it).  Consumed by one call.
void ggpm_sparse_backward_skip_wgrads(int yes);
bool ggpm_take_sparse_skip_wgrads();
int ggpm_gru_sparse_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs, float* work,
                                 size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                 int ld_dwh, ggpm_stream_t stream);
int ggpm_lstm_sparse_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, float* work, size_t work_bytes,
                                  float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h, int ld_dwu, float* dWf_h,
                                  int ld_dwf, ggpm_stream_t stream);
hipEvent_t ggpm_wgrad_event(int i);          // small pool of re-recordable events, per thread (mpn_gru.hip)
// ggpm_backward_skip_x_sums (include/ggpm_hip.h): consumed by the next dense level backward of this thread.
bool ggpm_take_skip_x_sums();
// ggpm_level_prefer_narrow (include/ggpm_hip.h): state of this thread's switch (mpn_gru.hip).
bool ggpm_prefer_narrow();
// ggpm_weights_packed (include/ggpm_hip.h): the next level / sparse call of this thread finds its packed weights in place.
bool ggpm_take_weights_packed();
// ggpm_forward_gather_state / ggpm_backward_scatter_state (include/ggpm_hip.h): consumed by the next sparse forward /
// backward of this thread.  -> true (and the pointers) once.
bool ggpm_take_gather_state(const float** src_h, const float** src_c, const int32_t** idx);
bool ggpm_take_scatter_state(float** dst_h, float** dst_c, const int32_t** idx);

// Gate-product dtype of the level calls issued by this thread: 0 fp32 (default), 1 bf16 operands.  Set by the encoder
// drivers from ggpm_enc_dims.gate_dtype for the duration of their call (mpn_gru.hip).
void ggpm_set_gate_dtype(int dtype);
int ggpm_gate_dtype();

// Levels whose depth-loop arrays are kept in bf16 under gate mode 1 (tile_mma.h: "bf16 STORAGE"): dense training levels large
// enough that every weight-gradient contraction over their stashes runs on the bf16 tall kernel, which then reads the
// stashes as they are.  (GRU levels; exported as ggpm_level_bf16_storage for the tests' choice of oracle mode.)
bool ggpm_bf16_storage_applies(int E1, int H);       // (gemm.hip: one stash slot alone must qualify for the bf16 tall kernel)
// out[i] = sum_t src[t][i] over `slots` slots of `slot_floats` elements, src fp32 or bf16 (first half of its buffer); fixed order
int ggpm_sum_slots_any(const float* src, int slots, size_t slot_floats, float* out, bool src_bf16, ggpm_stream_t stream);
// column sums of a [rows, ld] fp32 or bf16 matrix (losses.hip / gemm.hip: ggpm_colsum is the fp32 entry point)
int ggpm_colsum_any(const float* A, int lda, int rows, int cols, float* out, float* scratch, bool a_bf16, ggpm_stream_t stream);

// Optional per-launch timing (bench.py roofline): implemented in capi.hip.
void ggpm_timing_tag(int tag);       // 1 atom, 2 attachment, 3 motif level, 0 untagged; collected as which + 8 * tag
void ggpm_timing_begin(int which, hipStream_t s, double flops);
void ggpm_timing_end(int which, hipStream_t s);
