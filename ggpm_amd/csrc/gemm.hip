// Generic fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32), used for everything that is
// NOT inside the per-depth message loop: hoisted input projections, readouts W_o/W_i/W_c/W_root, and
// all weight/input gradients (tall split-K contractions over depth*E rows).
//
//   C[m,n] = act( sum_k A'(m,k) B'(k,n) + bias[n] + (accumulate ? C[m,n] : 0) )
//
// 64x64x16 workgroup tile, 4 waves (2x2), each wave one 32x32 accumulator (16 AGPR/VGPRs).
// Operands are staged k-major in LDS so that the MFMA A/B fragments (lane l: row l&31, k = l>>5)
// are conflict-free ds_read_b32.  MFMA f32 is an exact k-ordered fmaf chain (guide section 3), so the
// result does not depend on how tiles are scheduled; split-K partial slabs are reduced in a fixed order.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDT = 68;   // LDT: padded LDS row (floats), 16B multiple

struct GemmArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc; int n_pad;
    const float* bias;
    int accumulate, act, zero_row0;
    int vecA, vecB;          // 16-byte vector loads legal for A / B
    int k_chunk;             // K range per blockIdx.z (multiple of BK); == K when not split
    float* ws;               // split-K slabs [gridDim.z][M][N] or nullptr
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case GGPM_ACT_RELU: return v > 0.f ? v : 0.f;
        case GGPM_ACT_TANH: return tanhf(v);
        case GGPM_ACT_SIGMOID: return ggpm_sigmoid(v);
        default: return v;
    }
}

// One BK x 64 operand tile = 512 float4 = 2 per thread.  CONTIG_K: global element (row r, k) at
// P[r*ld + k] (k contiguous) else at P[k*ld + r] (row contiguous).  fetch: global -> registers (issued
// one k-step ahead so the loads fly under the MFMAs); stash: registers -> LDS (k-major).
template <bool CONTIG_K>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ P, int ld, int r0, int R, int k0, int kend,
                                           bool vec, float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + 256 * i;
        float4 x = ggpm_zero4();
        if (CONTIG_K) {
            const int gr = r0 + (f >> 3), gk = k0 + (f & 7) * 4;
            if (gr < R && gk < kend) {
                const float* src = P + (size_t)gr * ld + gk;
                if (vec && gk + 3 < kend) x = ggpm_ld4(src);
                else {
                    x.x = src[0];
                    if (gk + 1 < kend) x.y = src[1];
                    if (gk + 2 < kend) x.z = src[2];
                    if (gk + 3 < kend) x.w = src[3];
                }
            }
        } else {
            const int gk = k0 + (f >> 4), gr = r0 + (f & 15) * 4;
            if (gk < kend && gr < R) {
                const float* src = P + (size_t)gk * ld + gr;
                if (vec && gr + 3 < R) x = ggpm_ld4(src);
                else {
                    x.x = src[0];
                    if (gr + 1 < R) x.y = src[1];
                    if (gr + 2 < R) x.z = src[2];
                    if (gr + 3 < R) x.w = src[3];
                }
            }
        }
        v[i] = x;
    }
}

template <bool CONTIG_K>
__device__ __forceinline__ void stash_tile(const float4 (&v)[2], float (*T)[LDT]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + 256 * i;
        if (CONTIG_K) {
            const int r = f >> 3, kq = (f & 7) * 4;
            T[kq + 0][r] = v[i].x; T[kq + 1][r] = v[i].y; T[kq + 2][r] = v[i].z; T[kq + 3][r] = v[i].w;
        } else {
            const int k = f >> 4, rq = (f & 15) * 4;
            *reinterpret_cast<float4*>(&T[k][rq]) = v[i];
        }
    }
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (n0 < g.N) {
        float4 ra[2], rb[2];
        fetch_tile<!TA>(g.A, g.lda, m0, g.M, kbeg, kend, g.vecA != 0, ra);
        fetch_tile<TB>(g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB != 0, rb);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            stash_tile<!TA>(ra, As);
            stash_tile<TB>(rb, Bs);
            __syncthreads();
            if (k0 + BK < kend) {       // next k-step's operands fly under this step's MFMAs
                fetch_tile<!TA>(g.A, g.lda, m0, g.M, k0 + BK, kend, g.vecA != 0, ra);
                fetch_tile<TB>(g.B, g.ldb, n0, g.N, k0 + BK, kend, g.vecB != 0, rb);
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
                float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }

    const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= g.M) continue;
        if (g.ws) {
            if (n < g.N) g.ws[((size_t)blockIdx.z * g.M + m) * g.N + n] = acc[r];
            continue;
        }
        if (n < g.N) {
            float v = acc[r];
            if (g.bias) v += g.bias[n];
            float* dst = g.C + (size_t)m * g.ldc + n;
            if (g.accumulate) v += *dst;
            v = apply_act(v, g.act);
            if (g.zero_row0 && m == 0) v = 0.f;
            *dst = v;
        } else if (n < g.n_pad) {
            g.C[(size_t)m * g.ldc + n] = 0.f;
        }
    }
}

__global__ void splitk_reduce(GemmArgs g, int splits) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= g.n_pad) return;
    float* dst = g.C + (size_t)m * g.ldc + n;
    if (n >= g.N) { *dst = 0.f; return; }
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += g.ws[((size_t)z * g.M + m) * g.N + n];   // fixed order
    if (g.bias) v += g.bias[n];
    if (g.accumulate) v += *dst;
    v = apply_act(v, g.act);
    if (g.zero_row0 && m == 0) v = 0.f;
    *dst = v;
}

// choose the split: enough workgroups to cover the chip a few times over, chunks of >= 256 k.
inline int choose_splits(int M, int N, int K) {
    const int tiles = ggpm_ceil_div(M, BM) * ggpm_ceil_div(N, BN);
    if (K < 2048 || tiles >= 512) return 1;
    int want = ggpm_ceil_div(1024, tiles);
    int maxs = K / 256;
    int s = want < maxs ? want : maxs;
    return s < 1 ? 1 : s;
}

constexpr int CS_ROWS = 256;   // max row chunks of the column-sum first stage (workspace = 256*N floats)

// stage 1: block = 64 columns x 4 row lanes; grid (ceil(N/64), CS_ROWS); chunk c sums rows [c*per, (c+1)*per)
__global__ void __launch_bounds__(256) colsum_stage1(const float* __restrict__ A, int lda, int M, int N,
                                                     float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int chunk = blockIdx.y;
    const int per = (M + gridDim.y - 1) / gridDim.y;
    const int lo = chunk * per, hi = min(M, lo + per);
    float v = 0.f;
    if (n < N)
        for (int m = lo + rl; m < hi; m += 4) v += A[(size_t)m * lda + n];
    red[rl][threadIdx.x & 63] = v;
    __syncthreads();
    if (rl == 0 && n < N) ws[(size_t)chunk * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// stage 2: block = 64 columns x 4 chunk lanes; fixed order -> deterministic
__global__ void __launch_bounds__(256) colsum_stage2(const float* __restrict__ ws, int N, int chunks,
                                                     float* __restrict__ out) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float v = 0.f;
    if (n < N)
        for (int c = rl; c < chunks; c += 4) v += ws[(size_t)c * N + n];
    red[rl][threadIdx.x & 63] = v;
    __syncthreads();
    if (rl == 0 && n < N) out[n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void act_backward_k(const float* __restrict__ dy, const float* __restrict__ y, int rows, int cols,
                               int ld, int act, int zero_row0, float* __restrict__ dpre) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    const size_t i = (size_t)r * ld + c;
    float g = dy[i], o = y[i];
    switch (act) {
        case GGPM_ACT_RELU: g = o > 0.f ? g : 0.f; break;
        case GGPM_ACT_TANH: g = g * (1.f - o * o); break;
        case GGPM_ACT_SIGMOID: g = g * o * (1.f - o); break;
        default: break;
    }
    if (zero_row0 && r == 0) g = 0.f;
    dpre[i] = g;
}

}  // namespace

extern "C" size_t ggpm_gemm_workspace_bytes(int M, int N, int K) {
    int s = choose_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int ggpm_gemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda,
                         const float* B, int ldb, float* C, int ldc, int n_pad, const float* bias,
                         int accumulate, int act, int zero_row0, float* splitk_ws, size_t splitk_ws_bytes,
                         ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || n_pad < N || n_pad > ldc) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.n_pad = n_pad; g.bias = bias; g.accumulate = accumulate; g.act = act; g.zero_row0 = zero_row0;
    g.vecA = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
    g.vecB = ((ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);
    int splits = 1;
    if (splitk_ws) {
        splits = choose_splits(M, N, K);
        if (splits > 1 && (size_t)splits * M * N * sizeof(float) > splitk_ws_bytes) splits = 1;
    }
    if (splits > 1) {
        g.k_chunk = ggpm_round_up(ggpm_ceil_div(K, splits), BK);
        splits = ggpm_ceil_div(K, g.k_chunk);
    }
    if (splits <= 1) { splits = 1; g.k_chunk = ggpm_round_up(K, BK); g.ws = nullptr; } else { g.ws = splitk_ws; }
    dim3 grid(ggpm_ceil_div(splits > 1 ? N : n_pad, BN), ggpm_ceil_div(M, BM), splits);
    if (!trans_a && !trans_b) gemm_kernel<false, false><<<grid, 256, 0, s>>>(g);
    else if (!trans_a && trans_b) gemm_kernel<false, true><<<grid, 256, 0, s>>>(g);
    else if (trans_a && !trans_b) gemm_kernel<true, false><<<grid, 256, 0, s>>>(g);
    else gemm_kernel<true, true><<<grid, 256, 0, s>>>(g);
    if (splits > 1) {
        dim3 rg(ggpm_ceil_div(n_pad, 256), M);
        splitk_reduce<<<rg, 256, 0, s>>>(g, splits);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_colsum(const float* A, int lda, int M, int N, float* out, float* ws, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!A || !out || !ws || M <= 0 || N <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int chunks = ggpm_ceil_div(M, 128);          // >= 128 rows per chunk, at most CS_ROWS chunks
    if (chunks > CS_ROWS) chunks = CS_ROWS;
    dim3 g1(ggpm_ceil_div(N, 64), chunks);
    colsum_stage1<<<g1, 256, 0, s>>>(A, lda, M, N, ws);
    colsum_stage2<<<ggpm_ceil_div(N, 64), 256, 0, s>>>(ws, N, chunks, out);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_act_backward(const float* dy, const float* y, int rows, int cols, int ld, int act,
                                 int zero_row0, float* dpre, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!dy || !y || !dpre || rows <= 0 || cols <= 0) return GGPM_ERR_ARG;
    dim3 grid(ggpm_ceil_div(cols, 256), rows);
    act_backward_k<<<grid, 256, 0, (hipStream_t)stream>>>(dy, y, rows, cols, ld, act, zero_row0, dpre);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
