// Row gathers and segmented sums over CSR lists (HBM / L2 bound; fp32 rows of H floats).
// One wave owns one destination row and walks its (short) list; lanes stride the feature dimension
// with 16-byte accesses when the strides allow it, so every load/store is a coalesced row segment.
#include "common.h"
#include <cmath>

namespace {

template <bool VEC>
__global__ void __launch_bounds__(256) segment_sum_k(const float* __restrict__ src, int ld_src,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, int rows, int width,
                                                     float* __restrict__ out, int ld_out, int accumulate, int zero_to) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lo = rowptr[r], hi = rowptr[r + 1];
    float* dst = out + (size_t)r * ld_out;
    for (int c = width + lane; c < zero_to; c += 64) dst[c] = 0.f;      // pad columns of the row
    if (VEC) {
        for (int c = lane * 4; c < width; c += 256) {
            float4 acc = accumulate ? ggpm_ld4(dst + c) : ggpm_zero4();
            for (int j = lo; j < hi; ++j) acc = acc + ggpm_ld4(src + (size_t)col[j] * ld_src + c);
            ggpm_st4(dst + c, acc);
        }
    } else {
        for (int c = lane; c < width; c += 64) {
            float acc = accumulate ? dst[c] : 0.f;
            for (int j = lo; j < hi; ++j) acc += src[(size_t)col[j] * ld_src + c];
            dst[c] = acc;
        }
    }
}

__global__ void __launch_bounds__(256) gather_rows_k(const float* __restrict__ table, int ld_table,
                                                     const int32_t* __restrict__ idx, int rows, int width,
                                                     float* __restrict__ out, int ld_out, int col_off, int zero_to) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int id = idx[r];
    float* dst = out + (size_t)r * ld_out + col_off;
    for (int c = width + lane; c < zero_to - col_off; c += 64) dst[c] = 0.f;
    if (id < 0) {
        for (int c = lane; c < width; c += 64) dst[c] = 0.f;
    } else {
        const float* s = table + (size_t)id * ld_table;
        for (int c = lane; c < width; c += 64) dst[c] = s[c];
    }
}

// dst[idx[r]] (+)= src[r]: the write side of gather_rows_k for UNIQUE indices (no atomics), one wave per row.
__global__ void __launch_bounds__(256) scatter_rows_k(const float* __restrict__ src, int ld_src,
                                                      const int32_t* __restrict__ idx, int rows, int width,
                                                      float* __restrict__ dst, int ld_dst, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int id = idx[r];
    if (id < 0) return;
    const float* s = src + (size_t)r * ld_src;
    float* d = dst + (size_t)id * ld_dst;
    if (accumulate) for (int c = lane; c < width; c += 64) d[c] += s[c];
    else for (int c = lane; c < width; c += 64) d[c] = s[c];
}

__global__ void onehot_k(const int32_t* __restrict__ idx, int rows, int classes, float* __restrict__ out,
                         int ld_out, int col_off, int zero_to) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (col_off + c >= (zero_to > col_off + classes ? zero_to : col_off + classes)) return;
    out[(size_t)r * ld_out + col_off + c] = (c < classes && idx[r] == c) ? 1.f : 0.f;
}

__global__ void embed_graph_k(const int64_t* __restrict__ fnode, int N1, const int64_t* __restrict__ fmess,
                              int E1, int atom_size, int bond_types, int max_pos, float* __restrict__ hnode,
                              int ld_n, float* __restrict__ hmess, int ld_m) {
    const int r = blockIdx.x;
    const int c = threadIdx.x;
    if (r < N1) {
        if (c < ld_n) hnode[(size_t)r * ld_n + c] = (c < atom_size && fnode[r] == c) ? 1.f : 0.f;
    } else {
        const int e = r - N1;
        if (e < E1 && c < ld_m) {
            const int64_t* f = fmess + (size_t)e * 4;
            const int64_t a = fnode[f[0]];
            float v = 0.f;
            if (c < atom_size) v = (a == c) ? 1.f : 0.f;
            else if (c < atom_size + bond_types) v = (f[2] == c - atom_size) ? 1.f : 0.f;
            else if (c < atom_size + bond_types + max_pos) v = (f[3] == c - atom_size - bond_types) ? 1.f : 0.f;
            hmess[(size_t)e * ld_m + c] = v;
        }
    }
}

}  // namespace

extern "C" int ggpm_segment_sum(const float* src, int ld_src, const int32_t* rowptr, const int32_t* col,
                                int rows, int width, float* out, int ld_out, int accumulate, int zero_to,
                                ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!src || !rowptr || !col || !out || rows <= 0 || width <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (width % 4 == 0) && (ld_src % 4 == 0) && (ld_out % 4 == 0) &&
                     (((uintptr_t)src & 15) == 0) && (((uintptr_t)out & 15) == 0);
    const int grid = ggpm_ceil_div(rows, 4);
    if (vec) segment_sum_k<true><<<grid, 256, 0, s>>>(src, ld_src, rowptr, col, rows, width, out, ld_out, accumulate,
                                                           zero_to);
    else segment_sum_k<false><<<grid, 256, 0, s>>>(src, ld_src, rowptr, col, rows, width, out, ld_out, accumulate, zero_to);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_gather_rows(const float* table, int ld_table, const int32_t* idx, int rows, int width,
                                float* out, int ld_out, int col_off, int zero_to, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!table || !idx || !out || rows <= 0 || width <= 0) return GGPM_ERR_ARG;
    gather_rows_k<<<ggpm_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(table, ld_table, idx, rows, width, out,
                                                                          ld_out, col_off, zero_to);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_scatter_rows(const float* src, int ld_src, const int32_t* idx, int rows, int width, float* dst,
                                 int ld_dst, int accumulate, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!src || !idx || !dst || rows <= 0 || width <= 0) return GGPM_ERR_ARG;
    scatter_rows_k<<<ggpm_ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(src, ld_src, idx, rows, width, dst, ld_dst,
                                                                           accumulate);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_onehot(const int32_t* idx, int rows, int classes, float* out, int ld_out, int col_off, int zero_to,
                           ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!idx || !out || rows <= 0 || classes <= 0) return GGPM_ERR_ARG;
    const int span = (zero_to > col_off + classes ? zero_to - col_off : classes);
    dim3 grid(ggpm_ceil_div(span, 64), rows);
    onehot_k<<<grid, 64, 0, (hipStream_t)stream>>>(idx, rows, classes, out, ld_out, col_off, zero_to);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_embed_graph(const int64_t* fnode, int N1, const int64_t* fmess, int E1, int atom_size,
                                int bond_types, int max_pos, float* hnode, int ld_n, float* hmess, int ld_m,
                                ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!fnode || !fmess || !hnode || !hmess || N1 <= 0 || E1 <= 0) return GGPM_ERR_ARG;
    if (ld_n > 256 || ld_m > 256 || ld_n < atom_size || ld_m < atom_size + bond_types + max_pos)
        return GGPM_ERR_UNSUPPORTED;
    embed_graph_k<<<N1 + E1, 256, 0, (hipStream_t)stream>>>(fnode, N1, fmess, E1, atom_size, bond_types, max_pos,
                                                           hnode, ld_n, hmess, ld_m);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}


// ---------------------------------------------------------------- dropout (counter-based, stateless)
namespace {
__device__ __forceinline__ unsigned int fmix32(unsigned int h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__global__ void dropout_k(float* __restrict__ x, int rows, int cols, int ld, unsigned int thresh, float scale,
                          unsigned int seed_lo, unsigned int seed_hi, unsigned int site) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    const unsigned int idx = (unsigned int)r * (unsigned int)cols + (unsigned int)c;
    unsigned int h = fmix32(idx * 0x9E3779B1u + seed_lo);
    h = fmix32(h ^ (seed_hi + site * 0x7F4A7C15u));
    float* p = x + (size_t)r * ld + c;
    *p = ((h >> 8) >= thresh) ? *p * scale : 0.f;
}
}  // namespace

namespace {
// Adam (torch.optim.Adam's arithmetic, vae_train.py:60) over one flat fp32 buffer: 16 B per lane, grid-stride.
__global__ void __launch_bounds__(256) adam_flat_k(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n, float b1,
                                                   float b2, float eps, float wd, float step_size, float inv_bc2_sqrt) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = ggpm_ld4(p + 4 * i), gg = ggpm_ld4(g + 4 * i), mm = ggpm_ld4(m + 4 * i), vv = ggpm_ld4(v + 4 * i);
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = wd != 0.f ? G[k] + wd * P[k] : G[k];
            M[k] = M[k] + (1.f - b1) * (gr - M[k]);
            V[k] = b2 * V[k] + (1.f - b2) * gr * gr;
            P[k] -= step_size * M[k] / (sqrtf(V[k]) * inv_bc2_sqrt + eps);
        }
        ggpm_st4(p + 4 * i, pp); ggpm_st4(m + 4 * i, mm); ggpm_st4(v + 4 * i, vv);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {          // tail of up to three elements
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gr = wd != 0.f ? g[i] + wd * p[i] : g[i];
        const float mk = m[i] + (1.f - b1) * (gr - m[i]);
        const float vk = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mk; v[i] = vk;
        p[i] -= step_size * mk / (sqrtf(vk) * inv_bc2_sqrt + eps);
    }
}
}  // namespace

extern "C" int ggpm_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!p || !g || !m || !v || n == 0 || step < 1) return GGPM_ERR_ARG;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return GGPM_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    size_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    adam_flat_k<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, beta1, beta2, eps, weight_decay, step_size,
                                                                 inv_bc2_sqrt);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_dropout(float* x, int rows, int cols, int ld, float p, unsigned int seed_lo, unsigned int seed_hi,
                            int site, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!x || rows <= 0 || cols <= 0 || ld < cols || p < 0.f || p >= 1.f) return GGPM_ERR_ARG;
    if (p == 0.f) return GGPM_OK;
    const unsigned int thresh = (unsigned int)((double)p * 16777216.0);
    dim3 grid(ggpm_ceil_div(cols, 256), rows);
    dropout_k<<<grid, 256, 0, (hipStream_t)stream>>>(x, rows, cols, ld, thresh, 1.0f / (1.0f - p), seed_lo, seed_hi,
                                                      (unsigned int)site);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
