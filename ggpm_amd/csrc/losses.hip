// Decoder score-head losses -- reference ggpm/decoder.py:66-69, 136-164, 262-283.
//
//   softmax cross entropy, reduction = sum, with the reference's additive vocabulary mask fused in:
//       icls_scores = iclsNN(cls_vecs) + vocab.get_mask(cls_labs)     (decoder.py:143-157, vocab.py:34-41,56-58)
//       loss = sum_m ( logsumexp_n z[m, n] - z[m, label[m]] ),   z[m, :] = logits[m, :] + mask[mask_row[m], :]
//     One wave per prediction row walks the class dimension three times out of L2 (max, sum of exponentials, gradient);
//     the per-row losses are reduced in a fixed order, so the loss is bitwise reproducible.  The backward gradient
//     dz = softmax(z) - onehot(label) is written by the same launch (the loss is always differentiated in training).
//   BCE with logits, reduction = sum (topology head): loss = sum max(x,0) - x y + log(1 + exp(-|x|)), dx = sigmoid(x) - y.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256) softmax_ce_k(const float* __restrict__ logits, int ld, int M, int N,
                                                    const float* __restrict__ mask, int ld_mask,
                                                    const int32_t* __restrict__ mask_row,
                                                    const int32_t* __restrict__ label, float* __restrict__ row_loss,
                                                    float* __restrict__ dlogits, int ld_d, int32_t* __restrict__ argmax) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* z = logits + (size_t)m * ld;
    const float* mk = (mask && mask_row) ? mask + (size_t)mask_row[m] * ld_mask : nullptr;
    const int lab = label[m];
    float mx = -3.0e38f;
    int amax = 0;
    for (int n = lane; n < N; n += 64) {
        const float v = z[n] + (mk ? mk[n] : 0.f);
        if (v > mx) { mx = v; amax = n; }
    }
    const float wmx = wave_max(mx);
    if (argmax) {        // first index attaining the maximum (torch.max semantics)
        int cand = (mx == wmx) ? amax : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
        if (lane == 0) argmax[m] = cand;
    }
    float se = 0.f;
    for (int n = lane; n < N; n += 64) se += expf(z[n] + (mk ? mk[n] : 0.f) - wmx);
    se = wave_sum(se);
    const float lse = wmx + logf(se);
    if (lane == 0) row_loss[m] = lse - (z[lab] + (mk ? mk[lab] : 0.f));
    if (dlogits) {
        float* d = dlogits + (size_t)m * ld_d;
        const float inv = 1.f / se;
        for (int n = lane; n < N; n += 64)
            d[n] = expf(z[n] + (mk ? mk[n] : 0.f) - wmx) * inv - (n == lab ? 1.f : 0.f);
    }
}

__global__ void bce_logits_k(const float* __restrict__ x, const float* __restrict__ y, int M,
                             float* __restrict__ row_loss, float* __restrict__ dx) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float v = x[m], t = y[m];
    row_loss[m] = fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
    if (dx) dx[m] = ggpm_sigmoid(v) - t;
}

// out[0] = sum_m v[m] in a fixed order (single workgroup, pairwise tree over fixed slots)
__global__ void __launch_bounds__(256) sum_rows_k(const float* __restrict__ v, int M, float* __restrict__ out) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) acc += v[m];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

__global__ void scale_rows_k(float* __restrict__ d, int ld, int M, int N, const float* __restrict__ scale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n < N) d[(size_t)m * ld + n] *= scale[0];
}

}  // namespace

// loss[0] = sum of the row losses; dlogits (nullable, [M][ld_d]) = d loss / d logits for an upstream gradient of 1;
// argmax (nullable, [M]) = predicted class per row (for get_accuracy, ggpm/nnutils.py:84-87); work: M floats.
extern "C" int ggpm_softmax_ce(const float* logits, int ld, int M, int N, const float* mask, int ld_mask,
                               const int32_t* mask_row, const int32_t* label, float* loss, float* dlogits, int ld_d,
                               int32_t* argmax, float* work, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!logits || !label || !loss || !work || M <= 0 || N <= 0 || ld < N || (dlogits && ld_d < N) ||
        ((mask == nullptr) != (mask_row == nullptr)))
        return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    softmax_ce_k<<<ggpm_ceil_div(M, 4), 256, 0, s>>>(logits, ld, M, N, mask, ld_mask, mask_row, label, work, dlogits, ld_d,
                                                    argmax);
    sum_rows_k<<<1, 256, 0, s>>>(work, M, loss);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_bce_logits(const float* x, const float* y, int M, float* loss, float* dx, float* work,
                               ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!x || !y || !loss || !work || M <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bce_logits_k<<<ggpm_ceil_div(M, 256), 256, 0, s>>>(x, y, M, work, dx);
    sum_rows_k<<<1, 256, 0, s>>>(work, M, loss);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// d[m, 0:N] *= scale[0]   (chain rule for an upstream gradient that is a device scalar)
extern "C" int ggpm_scale_rows(float* d, int ld, int M, int N, const float* scale, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!d || !scale || M <= 0 || N <= 0 || ld < N) return GGPM_ERR_ARG;
    dim3 grid(ggpm_ceil_div(N, 256), M);
    scale_rows_k<<<grid, 256, 0, (hipStream_t)stream>>>(d, ld, M, N, scale);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The four accuracies of a teacher-forced decoder pass in ONE launch (reference ggpm/decoder.py:262-283, get_accuracy /
// get_accuracy_bin / get_accuracy_sym of ggpm/nnutils.py:84-97): block 0 = motif class, 1 = attachment class (arg-max
// of the loss kernel against the label), 2 = topology (score >= 0 against the label), 3 = attachment (the first
// candidate holds the row maximum).  Counts are integers: the result does not depend on the summation order.
namespace {
struct AccArgs {
    const int32_t *cls_pred, *icls_pred;
    const void *cls_lab, *icls_lab, *topo_lab;
    const float *topo, *assm;
    int n_cls, n_topo, ld_topo, P, C, ld_assm, lab64;
    float* out;
};
__device__ __forceinline__ long long acc_label(const void* p, int i, int lab64) {
    return lab64 ? reinterpret_cast<const long long*>(p)[i] : (long long)reinterpret_cast<const int32_t*>(p)[i];
}
__global__ void __launch_bounds__(256) head_accuracies_k(AccArgs a) {
    __shared__ int red[256];
    const int which = blockIdx.x;
    const int n = which < 2 ? a.n_cls : which == 2 ? a.n_topo : a.P;
    int hit = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (which == 0) hit += (long long)a.cls_pred[i] == acc_label(a.cls_lab, i, a.lab64);
        else if (which == 1) hit += (long long)a.icls_pred[i] == acc_label(a.icls_lab, i, a.lab64);
        else if (which == 2) hit += (long long)(a.topo[(size_t)i * a.ld_topo] >= 0.f) == acc_label(a.topo_lab, i, a.lab64);
        else {
            const float* row = a.assm + (size_t)i * a.ld_assm;
            float m = row[0];
            for (int c = 1; c < a.C; ++c) m = fmaxf(m, row[c]);
            hit += row[0] == m;
        }
    }
    red[threadIdx.x] = hit;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[which] = (which == 3 && n <= 0) ? 1.f : (float)red[0] / (float)n;
}
}  // namespace

extern "C" int ggpm_head_accuracies(const int32_t* cls_pred, const void* cls_lab, const int32_t* icls_pred, const void* icls_lab,
                                    int n_cls, const float* topo, int ld_topo, const void* topo_lab, int n_topo,
                                    const float* assm, int ld_assm, int P, int C, int labels_int64, float* out4,
                                    ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!out4 || n_cls < 0 || n_topo < 0 || P < 0 || (n_cls > 0 && (!cls_pred || !cls_lab || !icls_pred || !icls_lab)) ||
        (n_topo > 0 && (!topo || !topo_lab || ld_topo < 1)) || (P > 0 && (!assm || C < 1 || ld_assm < C)))
        return GGPM_ERR_ARG;
    AccArgs a = {cls_pred, icls_pred, cls_lab, icls_lab, topo_lab, topo, assm, n_cls, n_topo, ld_topo, P, C, ld_assm,
                 labels_int64 ? 1 : 0, out4};
    head_accuracies_k<<<4, 256, 0, (hipStream_t)stream>>>(a);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// KL head of HierPropertyVAE.rsample (reference ggpm/property_vae.py:26-33), the elementwise part after the two
// [B,H]x[H,L] products:  lv = -|pv| ; kl = -0.5 * sum(1 + lv - mean^2 - exp(lv)) / B ; z = mean + exp(lv/2) * eps.
// One workgroup (B*L is ~1e3); the sum runs over fixed slots -> bitwise reproducible.
namespace {
__global__ void __launch_bounds__(256) rsample_fwd_k(const float* __restrict__ mean, const float* __restrict__ pv,
                                                     const float* __restrict__ eps, int n, int B, float* __restrict__ z,
                                                     float* __restrict__ kl) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float m = mean[i], lv = -fabsf(pv[i]);
        acc += 1.f + lv - m * m - expf(lv);
        z[i] = eps ? m + expf(0.5f * lv) * eps[i] : m;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) kl[0] = -0.5f * part[0] / (float)B;
}

__global__ void rsample_bwd_k(const float* __restrict__ mean, const float* __restrict__ pv, const float* __restrict__ eps,
                              const float* __restrict__ dz, const float* __restrict__ dkl, int n, int B,
                              float* __restrict__ dmean, float* __restrict__ dpv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = mean[i], p = pv[i], lv = -fabsf(p);
    const float gk = dkl ? dkl[0] : 0.f, gz = dz ? dz[i] : 0.f;
    dmean[i] = gz + gk * m / (float)B;
    float dlv = gk * (-0.5f / (float)B) * (1.f - expf(lv));
    if (eps) dlv += gz * eps[i] * 0.5f * expf(0.5f * lv);
    dpv[i] = (p > 0.f ? -1.f : (p < 0.f ? 1.f : 0.f)) * dlv;      // d(-|p|)/dp = -sign(p)  (0 at p = 0, as torch.abs)
}
}  // namespace

extern "C" int ggpm_rsample_forward(const float* mean, const float* pv, const float* eps, int B, int L, float* z,
                                    float* kl, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!mean || !pv || !z || !kl || B <= 0 || L <= 0) return GGPM_ERR_ARG;
    rsample_fwd_k<<<1, 256, 0, (hipStream_t)stream>>>(mean, pv, eps, B * L, B, z, kl);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_rsample_backward(const float* mean, const float* pv, const float* eps, const float* dz,
                                     const float* dkl, int B, int L, float* dmean, float* dpv, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!mean || !pv || !dmean || !dpv || B <= 0 || L <= 0) return GGPM_ERR_ARG;
    rsample_bwd_k<<<ggpm_ceil_div(B * L, 256), 256, 0, (hipStream_t)stream>>>(mean, pv, eps, dz, dkl, B * L, B, dmean, dpv);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
