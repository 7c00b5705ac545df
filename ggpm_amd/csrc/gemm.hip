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
//
// Tall weight-gradient contractions (C = A^T B with K = depth*E in the tens of thousands and a 300 x 300 output) take
// gemm_tn_tall instead: 160 x 160 output tile per workgroup on v_mfma_f32_16x16x4_f32 (25 independent accumulator
// blocks per wave, 10 LDS reads per 25 MFMAs), operands double-buffered in LDS with ONE LDS-only barrier per k-step
// and the next step's global loads issued a full step ahead.  The 64 x 64 kernel moved 31 B of operands per KFLOP
// through L2 and spent as long on that as on the MFMAs (253 + 270 us of a 500 us launch, not overlapped).
#include <stdio.h>
#include <stdlib.h>
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
    unsigned long long* dbg; // dev (GGPM_GEMM_DEBUG): loop stamps of workgroup 0
    // K segments (gemm_kernel_v2 only): C = sum_s A_s B_s with operand bases / leading dimensions / lengths per
    // segment, e.g. the three gate slabs of dX against the three gate weights.  nseg == 0: the single (A, B, K) above.
    int nseg;
    const float* segA[GGPM_GEMM_MAX_GROUP];
    const float* segB[GGPM_GEMM_MAX_GROUP];
    int seg_lda[GGPM_GEMM_MAX_GROUP], seg_ldb[GGPM_GEMM_MAX_GROUP], segK[GGPM_GEMM_MAX_GROUP];
};

struct GemmGroupArgs {       // independent problems of one shape in one launch (blockIdx.z picks the problem)
    GemmArgs p[GGPM_GEMM_MAX_GROUP];
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

// ---------------------------------------------------------------------------------------------------------
// The same 64 x 64 tile with the pipeline of gemm_tn_tall (used whenever both operands allow 16-byte loads):
// 64-wide k-steps, LDS double-buffered with ONE LDS-only barrier per step, two register sets so that every global
// load has two full steps to land, buffer loads whose descriptors make row / column / K tails read zeros (no
// branches), and the staging instructions dealt out by hand between pairs of MFMAs.  The old loop exposed one global
// round trip per 32-wide k-step, which is what bounded the short-K projections (5-30 steps per workgroup).
constexpr int BK2 = 64;

template <bool CONTIG_K>
struct OperandStage {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff[4], vstep;
    int loff[4];       // LDS float offset of the thread's 4 float4 inside one operand buffer
    int kq[4];         // CONTIG_K: first k of the float4 inside the step (for the K-tail mask)

    // P: operand base, R rows/columns of the output dimension starting at r0, k range [kbeg, kend)
    __device__ __forceinline__ void init(const float* P, int ld, int r0, int R, int kbeg, int kend, int total_rows) {
        // The descriptor ends EXACTLY behind the last element this operand owns (element (r, k) sits at P[r*ld + k] if
        // CONTIG_K, at P[k*ld + r] otherwise): P may point into the middle of a wider matrix whose allocation ends
        // less than a row behind that element, and everything past the end must read as zero, not as memory.
        const size_t last = CONTIG_K ? (size_t)(total_rows - 1) * ld + kend : (size_t)(max(kend, 1) - 1) * ld + R;
        const size_t bytes = kend > 0 ? last * 4 : 0;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (unsigned)bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = threadIdx.x + 256 * i;           // 1024 float4 per operand per step
            if (CONTIG_K) {
                const int r = f >> 4, k4 = (f & 15) * 4;
                kq[i] = k4;
                loff[i] = k4 * LDT + r;
                voff[i] = r0 + r < R ? (unsigned)(((size_t)(r0 + r) * ld + kbeg + k4) * 4) : 0xffffff00u;
            } else {
                const int k = f >> 4, c = (f & 15) * 4;
                kq[i] = 0;
                loff[i] = k * LDT + c;
                voff[i] = r0 + c < R ? (unsigned)(((size_t)(kbeg + k) * ld + r0 + c) * 4) : 0xffffff00u;
            }
        }
        vstep = CONTIG_K ? BK2 * 4u : (unsigned)BK2 * ld * 4;
    }
    __device__ __forceinline__ void load(f32x4& r, int i) {
        r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
        if (voff[i] != 0xffffff00u) voff[i] += vstep;
    }
    // klen: valid k of the step being written (CONTIG_K reads past the row's K into the padding / the next row)
    __device__ __forceinline__ void store(float* T, const f32x4& r, int i, int klen) const {
        if (CONTIG_K) {
            float* d = T + loff[i];
            d[0] = kq[i] + 0 < klen ? r[0] : 0.f;
            d[LDT] = kq[i] + 1 < klen ? r[1] : 0.f;
            d[2 * LDT] = kq[i] + 2 < klen ? r[2] : 0.f;
            d[3 * LDT] = kq[i] + 3 < klen ? r[3] : 0.f;
        } else {
            *reinterpret_cast<f32x4*>(T + loff[i]) = r;
        }
    }
};

template <bool TA, bool TB>
__device__ __forceinline__ void gemm_v2_tile(const GemmArgs& g, int bx, int by, int bz, float (&Ls)[2][2][BK2 * LDT]) {
    const int m0 = by * BM, n0 = bx * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    OperandStage<!TA> sa;
    OperandStage<TB> sb;

    const int fA = (lane >> 5) * LDT + wm * 32 + (lane & 31), fB = (lane >> 5) * LDT + wn * 32 + (lane & 31);
    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    float fa[2][2], fb[2][2];
    auto frag = [&](int buf, int pair2, int set) {       // the two k-pairs 2*pair2, 2*pair2 + 1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            fa[set][u] = Ls[buf][0][fA + (4 * pair2 + 2 * u) * LDT];
            fb[set][u] = Ls[buf][1][fB + (4 * pair2 + 2 * u) * LDT];
        }
    };
    // One 64-wide k-step on buffer `cur` = 16 rounds of [staging | 2 MFMAs]: round r reads the fragments of round
    // r+1, rounds 0-7 issue the loads of step s+2 and write the rows of step s+1 (loaded one step ago) to the other
    // buffer.  klen_next: valid k of step s+1.
    auto step = [&](const f32x4 (&sa_)[4], const f32x4 (&sb_)[4], f32x4 (&la)[4], f32x4 (&lb)[4], int cur, int klen_next) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < 15) frag(cur, r + 1, (r + 1) & 1);
            if (r < 8) {
                if (r & 1) sb.load(lb[r >> 1], r >> 1); else sa.load(la[r >> 1], r >> 1);
                if (r & 1) sb.store(Ls[cur ^ 1][1], sb_[r >> 1], r >> 1, klen_next);
                else sa.store(Ls[cur ^ 1][0], sa_[r >> 1], r >> 1, klen_next);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[r & 1][0]), "v"(fb[r & 1][0]));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[r & 1][1]), "v"(fb[r & 1][1]));
            __builtin_amdgcn_sched_barrier(0);
        }
        ggpm_lds_barrier();
        frag(cur ^ 1, 0, 0);
    };

    if (n0 < g.N) {
        const int nseg = g.nseg ? g.nseg : 1;
        for (int sg = 0; sg < nseg; ++sg) {
            const float* A = g.nseg ? g.segA[sg] : g.A;
            const float* B = g.nseg ? g.segB[sg] : g.B;
            const int lda = g.nseg ? g.seg_lda[sg] : g.lda, ldb = g.nseg ? g.seg_ldb[sg] : g.ldb, K = g.nseg ? g.segK[sg] : g.K;
            const int kbeg = g.nseg ? 0 : bz * g.k_chunk, kend = g.nseg ? K : min(K, kbeg + g.k_chunk);
            sa.init(A, lda, m0, g.M, kbeg, kend, TA ? K : g.M);
            sb.init(B, ldb, n0, g.N, kbeg, kend, TB ? g.N : K);
            if (sg) ggpm_lds_barrier();     // the previous segment's last fragment reads are done
#pragma unroll
            for (int i = 0; i < 4; ++i) { sa.load(ra0[i], i); sb.load(rb0[i], i); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa.store(Ls[0][0], ra0[i], i, kend - kbeg);
                sb.store(Ls[0][1], rb0[i], i, kend - kbeg);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { sa.load(ra1[i], i); sb.load(rb1[i], i); }
            ggpm_lds_barrier();
            frag(0, 0, 0);
            // Whole steps in pairs (the register sets swap roles); loads past the end return zeros without touching
            // memory, so at most one step of the pair multiplies zeros.
            for (int k0 = kbeg; k0 < kend; k0 += 2 * BK2) {
                step(ra1, rb1, ra0, rb0, 0, kend - k0 - BK2);
                step(ra0, rb0, ra1, rb1, 1, kend - k0 - 2 * BK2);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs retire before the accumulator is read
    }

    const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= g.M) continue;
        if (g.ws) {
            if (n < g.N) g.ws[((size_t)bz * g.M + m) * g.N + n] = acc[r];
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

// XCD-contiguous renumbering of the launch's workgroups (see gemm_tn_tall): (x fastest, then y, then z)
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz) {
    const unsigned T = gridDim.x * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xq = T >> 3, xr = T & 7, xcd = L & 7;
    const unsigned logical = xcd * xq + min(xcd, xr) + (L >> 3);
    bx = logical % gridDim.x;
    by = (logical / gridDim.x) % gridDim.y;
    bz = logical / (gridDim.x * gridDim.y);
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel_v2(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float Ls[2][2][BK2 * LDT];      // [buffer][A | B][k][column]
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    gemm_v2_tile<TA, TB>(g, bx, by, bz, Ls);
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_group_v2(GemmGroupArgs gg) {
    __shared__ __attribute__((aligned(16))) float Ls[2][2][BK2 * LDT];
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    gemm_v2_tile<TA, TB>(gg.p[bz], bx, by, 0, Ls);
}

// ---------------------------------------------------------------------------------------------------------
// Small products (the readouts over a few hundred tree nodes, the input gradients of the motif / attachment levels):
// with 64 x 64 tiles such a launch has 5-50 workgroups, each a serial chain of K/2 MFMAs behind a drained prologue per
// K segment -- 27 us for 0.1 GFLOP.  gemm_small_v3 gives it 4x the workgroups and a 4x shorter chain: 32 x 32 output
// tile, the four waves split every 64-wide k-step four ways and sum their partial tiles through LDS at the end (fixed
// order), and the K segments run through ONE pipeline (the staging descriptors are re-pointed when the load cursor
// crosses a segment boundary; no drain, no second prologue).
constexpr int SM = 32, SN = 32, SK = 64, SLD = 36;

template <bool CONTIG_K>
struct SmallStage {
    static constexpr int LD = CONTIG_K ? 33 : SLD;      // 33: conflict-free transposing writes; 36: 16-byte aligned rows
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff[2], vstep;
    int loff[2], kq[2];

    __device__ __forceinline__ void init(const float* P, int ld, int r0, int R, int K, int total_rows) {
        const size_t last = CONTIG_K ? (size_t)(total_rows - 1) * ld + K : (size_t)(K - 1) * ld + R;
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (unsigned)(last * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = threadIdx.x + 256 * i;           // 512 float4 per operand per step
            if (CONTIG_K) {
                const int r = f >> 4, k4 = (f & 15) * 4;
                kq[i] = k4;
                loff[i] = k4 * LD + r;
                voff[i] = r0 + r < R ? (unsigned)(((size_t)(r0 + r) * ld + k4) * 4) : 0xffffff00u;
            } else {
                const int k = f >> 3, c = (f & 7) * 4;
                kq[i] = 0;
                loff[i] = k * LD + c;
                voff[i] = r0 + c < R ? (unsigned)(((size_t)k * ld + r0 + c) * 4) : 0xffffff00u;
            }
        }
        vstep = CONTIG_K ? SK * 4u : (unsigned)SK * ld * 4;
    }
    __device__ __forceinline__ void off() { voff[0] = voff[1] = 0xffffff00u; }
    __device__ __forceinline__ void advance() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (voff[i] != 0xffffff00u) voff[i] += vstep;
    }
    __device__ __forceinline__ void load(f32x4& r, int i) const {
        r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
    }
    __device__ __forceinline__ void store(float* T, const f32x4& r, int i, int klen) const {
        if (CONTIG_K) {
            float* d = T + loff[i];
            d[0] = kq[i] + 0 < klen ? r[0] : 0.f;
            d[LD] = kq[i] + 1 < klen ? r[1] : 0.f;
            d[2 * LD] = kq[i] + 2 < klen ? r[2] : 0.f;
            d[3 * LD] = kq[i] + 3 < klen ? r[3] : 0.f;
        } else {
            *reinterpret_cast<f32x4*>(T + loff[i]) = r;
        }
    }
};

// `chunk`: the K chunk [chunk * k_chunk, ...) of a split-K launch (g.ws set: the partial tile goes to slab `chunk` of the
// workspace, splitk_reduce_group sums the slabs in fixed order and applies the epilogue); 0 with k_chunk >= K otherwise.
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_v3_tile(const GemmArgs& g, int bx, int by, int chunk, float (&Ls)[2][2][SK * SLD]) {
    constexpr int LDA = SmallStage<!TA>::LD, LDB = SmallStage<TB>::LD;
    const int m0 = by * SM, n0 = bx * SN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kbeg = g.nseg ? 0 : chunk * g.k_chunk;
    const int klen = min(g.K, kbeg + g.k_chunk) - kbeg;
    const size_t a_off = TA ? (size_t)kbeg * g.lda : (size_t)kbeg, b_off = TB ? (size_t)kbeg : (size_t)kbeg * g.ldb;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (n0 < g.N) {
        const int nseg = g.nseg ? g.nseg : 1;
        auto seg_A = [&](int sg) { return g.nseg ? g.segA[sg] : g.A + a_off; };
        auto seg_B = [&](int sg) { return g.nseg ? g.segB[sg] : g.B + b_off; };
        auto seg_lda = [&](int sg) { return g.nseg ? g.seg_lda[sg] : g.lda; };
        auto seg_ldb = [&](int sg) { return g.nseg ? g.seg_ldb[sg] : g.ldb; };
        auto seg_K = [&](int sg) { return g.nseg ? g.segK[sg] : klen; };
        int total_steps = 0;
        for (int sg = 0; sg < nseg; ++sg) total_steps += (seg_K(sg) + SK - 1) / SK;

        SmallStage<!TA> sa;
        SmallStage<TB> sb;
        // load cursor (runs two steps ahead of the MFMAs) and store cursor (one step ahead): (segment, first k)
        int lseg = 0, lk = 0, sseg = 0, sk = 0;
        auto point = [&](int sg) {
            const int K = seg_K(sg);
            sa.init(seg_A(sg), seg_lda(sg), m0, g.M, K, TA ? K : g.M);
            sb.init(seg_B(sg), seg_ldb(sg), n0, g.N, K, TB ? g.N : K);
        };
        auto next_load = [&]() {         // after the loads of one step have been issued
            lk += SK;
            if (lseg < nseg && lk >= seg_K(lseg)) {
                ++lseg;
                lk = 0;
                if (lseg < nseg) point(lseg);
                else { sa.off(); sb.off(); }
            } else {
                sa.advance();
                sb.advance();
            }
        };
        auto store_klen = [&]() { return sseg < nseg ? seg_K(sseg) - sk : 0; };
        auto next_store = [&]() {
            sk += SK;
            if (sseg < nseg && sk >= seg_K(sseg)) { ++sseg; sk = 0; }
        };

        const int fA = (16 * wave + (lane >> 5)) * LDA + (lane & 31), fB = (16 * wave + (lane >> 5)) * LDB + (lane & 31);
        f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
        float fa[2][2], fb[2][2];
        auto frag = [&](int buf, int rnd, int set) {       // this wave's k-pairs 2*rnd, 2*rnd + 1 of its quarter step
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                fa[set][u] = Ls[buf][0][fA + (4 * rnd + 2 * u) * LDA];
                fb[set][u] = Ls[buf][1][fB + (4 * rnd + 2 * u) * LDB];
            }
        };
        // one 64-wide k-step on buffer `cur`: 4 rounds of [staging | 2 MFMAs] per wave
        auto step = [&](const f32x4 (&sa_)[2], const f32x4 (&sb_)[2], f32x4 (&la)[2], f32x4 (&lb)[2], int cur) {
            const int klen = store_klen();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < 3) frag(cur, r + 1, (r + 1) & 1);
                if (r & 1) sb.load(lb[r >> 1], r >> 1); else sa.load(la[r >> 1], r >> 1);
                if (r & 1) sb.store(Ls[cur ^ 1][1], sb_[r >> 1], r >> 1, klen);
                else sa.store(Ls[cur ^ 1][0], sa_[r >> 1], r >> 1, klen);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[r & 1][0]), "v"(fb[r & 1][0]));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa[r & 1][1]), "v"(fb[r & 1][1]));
                __builtin_amdgcn_sched_barrier(0);
            }
            next_load();
            next_store();
            ggpm_lds_barrier();
            frag(cur ^ 1, 0, 0);
        };

        point(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) { sa.load(ra0[i], i); sb.load(rb0[i], i); }
        next_load();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sa.store(Ls[0][0], ra0[i], i, store_klen());
            sb.store(Ls[0][1], rb0[i], i, store_klen());
        }
        next_store();
#pragma unroll
        for (int i = 0; i < 2; ++i) { sa.load(ra1[i], i); sb.load(rb1[i], i); }
        next_load();
        ggpm_lds_barrier();
        frag(0, 0, 0);
        for (int s2 = 0; s2 < total_steps; s2 += 2) {      // pairs: the register sets swap roles; a surplus step multiplies zeros
            step(ra1, rb1, ra0, rb0, 0);
            step(ra0, rb0, ra1, rb1, 1);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs retire before the accumulator is read
    }

    // sum the four waves' partial tiles in fixed order, then the usual epilogue (4 elements per thread)
    float* part = &Ls[0][0][0];                                  // 4 x 16 x 64 floats = 16 KB <= one buffer pair
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = threadIdx.x + 256 * j, r = idx >> 6, l = idx & 63;
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = n0 + (l & 31);
        if (m >= g.M) continue;
        if (g.ws) {
            if (n < g.N)
                g.ws[((size_t)chunk * g.M + m) * g.N + n] = (part[(0 * 16 + r) * 64 + l] + part[(1 * 16 + r) * 64 + l]) +
                                                             (part[(2 * 16 + r) * 64 + l] + part[(3 * 16 + r) * 64 + l]);
            continue;
        }
        if (n < g.N) {
            float v = (part[(0 * 16 + r) * 64 + l] + part[(1 * 16 + r) * 64 + l]) +
                      (part[(2 * 16 + r) * 64 + l] + part[(3 * 16 + r) * 64 + l]);
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

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_small_v3(GemmGroupArgs gg, int splits) {      // grid.z = problems x K chunks
    __shared__ __attribute__((aligned(16))) float Ls[2][2][SK * SLD];
    int bx, by, bz;
    xcd_tile(bx, by, bz);
    gemm_v3_tile<TA, TB>(gg.p[bz / splits], bx, by, bz % splits, Ls);
}

// sums the K-chunk slabs of a split gemm_small_v3 launch (fixed order) and applies the epilogue; blockIdx.z: the problem
__global__ void splitk_reduce_group(GemmGroupArgs gg, int splits) {
    const GemmArgs& g = gg.p[blockIdx.z];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= g.n_pad) return;
    float* dst = g.C + (size_t)m * g.ldc + n;
    if (n >= g.N) { *dst = 0.f; return; }
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += g.ws[((size_t)z * g.M + m) * g.N + n];
    if (g.bias) v += g.bias[n];
    if (g.accumulate) v += *dst;
    v = apply_act(v, g.act);
    if (g.zero_row0 && m == 0) v = 0.f;
    *dst = v;
}

// ---------------------------------------------------------------------------------------------------------
// C (or split-K slab) = A^T B for row-major A [K x lda], B [K x ldb]: both operands are contiguous along the
// output dimensions, so tiles go global -> registers -> LDS as float4 without any transposition.
constexpr int TM = 160, TN = 160, TK = 16, TLD = 176;   // TLD % 64 == 48: the 4 k-rows of a fragment read hit disjoint banks
constexpr int T_F4 = (TM + TN) / 4 * TK;                // float4 per k-step (1280) = 5 per thread

__global__ void __launch_bounds__(256) gemm_tn_tall(GemmGroupArgs gg, int splits) {
    __shared__ __attribute__((aligned(16))) float Ls[2][2 * TK * TLD];   // [buffer][A rows | B rows][k][column]
    // workgroups are dealt round-robin over the 8 XCDs: renumber so that the tiles of one K chunk share an L2
    const unsigned T = gridDim.x * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xq = T >> 3, xr = T & 7, xcd = L & 7;
    const unsigned logical = xcd * xq + min(xcd, xr) + (L >> 3);
    const int bx = logical % gridDim.x, by = (logical / gridDim.x) % gridDim.y;
    const int bzz = logical / (gridDim.x * gridDim.y);      // (problem of the group, K chunk)
    const GemmArgs& g = gg.p[bzz / splits];
    const int bz = bzz % splits;
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Waves 0-1 stage the A rows of every k-step, waves 2-3 the B rows: 640 float4 per operand = 5 per thread, fetched
    // with buffer loads.  The descriptor ends at row `kend`, so the K tail reads zeros, and columns beyond M / N get an
    // out-of-range offset (zeros too) -- no branches in the loop.  (The operand choice is wave-uniform and made scalar
    // explicitly: a lane-dependent descriptor would turn every load into a waterfall loop.)
    const int isb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const float* P = isb ? g.B : g.A;
    const int ld = isb ? g.ldb : g.lda, c0 = isb ? n0 : m0, climit = isb ? g.N : g.M;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0,
                                                                        (unsigned)(((size_t)(kend - 1) * ld + climit) * 4), 0x00020000);
    unsigned voff[5];
    int loff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int f = (threadIdx.x & 127) + 128 * i;
        const int k = f / (TM / 4), c = (f % (TM / 4)) * 4;
        loff[i] = isb * TK * TLD + k * TLD + c;
        voff[i] = c0 + c < climit ? (unsigned)(((size_t)(kbeg + k) * ld + c0 + c) * 4) : 0xffffff00u;
    }
    const unsigned vstep = (unsigned)TK * ld * 4;

    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two register sets: the loads of k-step s+2 are issued at the top of step s and written to LDS at the end of
    // step s+1, so every load has two full steps (~2.6 us of MFMAs) to land; LDS holds steps s and s+1.
    f32x4 r0[5], r1[5];
    auto fetch1 = [&](f32x4& r, int i) {
        r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
        if (voff[i] != 0xffffff00u) voff[i] += vstep;
    };
    // One k-step = 4 k-quads of 25 MFMAs on buffer `cur`, software-pipelined so that the matrix pipe never waits.
    // A quad is 5 rounds of [a few staging instructions | 5 MFMAs]: the pipe idles while a wave issues a RUN of LDS or
    // memory instructions (10 fragment reads in front of every quad cost 15 %), so they are dealt out by hand and
    // fenced -- the scheduler otherwise regroups them.
    //   * round i of every quad reads fragment i (one a, one b value) of the NEXT quad into the other fragment set;
    //   * quad 0 also issues the loads of k-step s+2 and writes the rows of step s+1 (loaded a step ago) to the other
    //     buffer: load i is issued right before row i is written, so each write waits for exactly vmcnt(5);
    //   * the step's only barrier sits in front of quad 3: by then every wave has written the other buffer and has
    //     received its last fragments of this one, so the next step's quad 0 is read behind it, under quad 3's MFMAs,
    //     and the next step's writes into `cur` cannot overtake a reader.
    const int fbase = (lane >> 4) * TLD + wm * 80 + (lane & 15), bdelta = TK * TLD + (wn - wm) * 80;
    float fa[2][5], fb[2][5];
    auto quad = [&](int set, const float* next, bool staging, const f32x4 (&rs_)[5], f32x4 (&rl)[5], int sbuf) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (staging) fetch1(rl[i], i);
            fa[set ^ 1][i] = next[16 * i];
            fb[set ^ 1][i] = next[bdelta + 16 * i];
            if (staging) *reinterpret_cast<f32x4*>(&Ls[sbuf][loff[i]]) = rs_[i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                // accumulator tied to one AGPR block: with the builtin the allocator gave 36 of the 200 MFMAs of the
                // unrolled loop a destination different from their addend and rotated 48 registers back every trip
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[set][i]), "v"(fb[set][j]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Branch-free: loads past `kend` return zeros without touching memory, so the loop runs an even number of steps
    // (at most one of them on zeros) and needs no tail conditions.
    auto step = [&](const f32x4 (&rs_)[5], f32x4 (&rl)[5], int cur) {
        const float* F = &Ls[cur][fbase];
        quad(0, F + 4 * TLD, true, rs_, rl, cur ^ 1);
        quad(1, F + 8 * TLD, false, rs_, rl, cur ^ 1);
        quad(0, F + 12 * TLD, false, rs_, rl, cur ^ 1);
        ggpm_lds_barrier();
        quad(1, &Ls[cur ^ 1][fbase], false, rs_, rl, cur ^ 1);
    };

    const bool dbg_on = g.dbg && L == 0 && threadIdx.x == 0;
    if (dbg_on) { g.dbg[0] = clock64(); g.dbg[1] = wall_clock64(); }
#pragma unroll
    for (int i = 0; i < 5; ++i) fetch1(r0[i], i);
#pragma unroll
    for (int i = 0; i < 5; ++i) *reinterpret_cast<f32x4*>(&Ls[0][loff[i]]) = r0[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) fetch1(r1[i], i);
    ggpm_lds_barrier();
    if (dbg_on) { g.dbg[2] = clock64(); g.dbg[3] = wall_clock64(); }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        fa[0][i] = Ls[0][fbase + 16 * i];
        fb[0][i] = Ls[0][fbase + bdelta + 16 * i];
    }
    for (int k0 = kbeg; k0 < kend; k0 += 2 * TK) {
        step(r1, r0, 0);
        step(r0, r1, 1);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs retire before the accumulators are read
    if (dbg_on) { g.dbg[4] = clock64(); g.dbg[5] = wall_clock64(); }

    // lane l holds rows 4*(l>>4) + 0..3, column (l & 15) of every 16 x 16 block
    if (g.ws) {
        // split-K slab in FRAGMENT order [chunk][tile][wave][block][lane][4]: one coalesced 16-byte store per block
        // (row-major slabs meant 400 predicated 4-byte stores per lane, ~15 us of a 125 us launch); tall_reduce
        // undoes the order once, after summing
        f32x4* slab = reinterpret_cast<f32x4*>(g.ws) +
                      ((((size_t)bz * (gridDim.x * gridDim.y) + by * gridDim.x + bx) * 4 + wave) * 25) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) slab[(i * 5 + j) * 64] = acc[i][j];
        return;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int n = n0 + wn * 80 + 16 * j + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 80 + 16 * i + 4 * (lane >> 4) + e;
                if (m >= g.M) continue;
                if (n < g.N) {
                    float v = acc[i][j][e];
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
}

// ---------------------------------------------------------------------------------------------------------
// The same contraction on bf16 operands (BASELINE configs[4]: "bf16"): C slab = rne(A)^T rne(B), fp32 accumulate, on
// v_mfma_f32_16x16x32_bf16.  At 16x the fp32 matrix rate the launch is bound by the operand stream (K x (M + N) floats
// from HBM / L2), not by the matrix pipe, so the structure is the plain one: 32-row k-steps, operands fetched as fp32
// with buffer loads one step ahead (K tail and column overhang read zeros), rounded to bf16 on their way into a
// double-buffered LDS image [k][A columns | B columns], ONE LDS-only barrier per step, several workgroups per CU to
// cover the load latency.  Both MFMA operands want 8 consecutive k per lane while memory is k-major, so the fragments
// are read with the transposing ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group): lane group g takes the
// step's rows 4g..4g+3 and 16+4g..16+4g+3 -- the SAME permutation of k for both operands, so the sum is unchanged --
// which puts the eight rows a 32-lane half reads at distinct bank octets with a row pitch of 168 dwords.
// Same output tile and the same fragment-order slab as gemm_tn_tall, so tall_reduce serves both.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));
typedef short gs16x4 __attribute__((ext_vector_type(4)));
constexpr int BTK = 32, BLD = 336;      // k rows per step; bf16 elements per LDS row (672 B = 168 dwords)

template <bool IN16>      // IN16: the operands are bf16 in memory (bf16 storage of the depth loops' stashes): no conversion
__global__ void __launch_bounds__(256, 2) gemm_tn_tall_bf16(GemmGroupArgs gg, int splits) {
    __shared__ __attribute__((aligned(16))) __bf16 Ls[2][BTK * BLD];
    const unsigned T = gridDim.x * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xq = T >> 3, xr = T & 7, xcd = L & 7;
    const unsigned logical = xcd * xq + min(xcd, xr) + (L >> 3);        // tiles of one K chunk share an XCD's L2
    const int bx = logical % gridDim.x, by = (logical / gridDim.x) % gridDim.y;
    const int bzz = logical / (gridDim.x * gridDim.y);
    const GemmArgs& g = gg.p[bzz / splits];
    const int bz = bzz % splits;
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // waves 0-1 stage the A rows of a step, waves 2-3 the B rows: 32 rows x 40 float4 = 10 per thread
    const int isb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
    const float* P = isb ? g.B : g.A;
    const int ld = isb ? g.ldb : g.lda, c0 = isb ? n0 : m0, climit = isb ? g.N : g.M;
    constexpr unsigned ES = IN16 ? 2 : 4;       // bytes per operand element in memory
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(P), 0, kend > kbeg ? (unsigned)(((size_t)(kend - 1) * ld + climit) * ES) : 0u, 0x00020000);
    unsigned voff[10];
    int loff[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int f = (threadIdx.x & 127) + 128 * i;
        const int k = f / (TM / 4), c = (f % (TM / 4)) * 4;
        loff[i] = k * BLD + isb * TM + c;
        voff[i] = c0 + c < climit ? (unsigned)(((size_t)(kbeg + k) * ld + c0 + c) * ES) : 0xffffff00u;
    }
    const unsigned vstep = (unsigned)BTK * ld * ES;
    f32x4 r[IN16 ? 1 : 10];
    uint2 r16[IN16 ? 10 : 1];
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if constexpr (IN16) r16[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff[i], 0, 0));
            else r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
            if (voff[i] != 0xffffff00u) voff[i] += vstep;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if constexpr (IN16) {
                *reinterpret_cast<uint2*>(&Ls[buf][loff[i]]) = r16[i];
            } else {
                const gbf16x4 w = {(__bf16)r[i][0], (__bf16)r[i][1], (__bf16)r[i][2], (__bf16)r[i][3]};
                *reinterpret_cast<gbf16x4*>(&Ls[buf][loff[i]]) = w;
            }
        }
    };

    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed fragment reads: lane 4q+p of 16-lane group g supplies row 4g+q (resp. 16+4g+q), columns 4p..4p+3 of
    // the 16-column block; lane j of the group receives column j, the four rows in its four elements
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fa0 = (4 * grp + q) * BLD + wm * 80 + 4 * pp;
    const int fb0 = (4 * grp + q) * BLD + TM + wn * 80 + 4 * pp;
    typedef __attribute__((address_space(3))) gs16x4* lds_v4;
    auto frag = [&](const __bf16* base) -> gbf16x8 {
        const gs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
        const gs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 16 * BLD));
        gs16x4 v[2] = {lo, hi};
        return __builtin_bit_cast(gbf16x8, v);
    };

    fetch();
    put(0);
    fetch();
    ggpm_lds_barrier();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BTK, cur ^= 1) {
        gbf16x8 fa[5], fb[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            fa[i] = frag(&Ls[cur][fa0 + 16 * i]);
            fb[i] = frag(&Ls[cur][fb0 + 16 * i]);
        }
        put(cur ^ 1);           // rows of step s+1 (loaded a step ago); every wave finished reading that buffer before the
        fetch();                // barrier that ended step s-1.  Then the loads of step s+2 fly under this step's MFMAs
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        ggpm_lds_barrier();
    }
    // split-K slab in the fragment order of gemm_tn_tall (lane l: rows 4*(l>>4) + 0..3, column l & 15 of every block)
    f32x4* slab = reinterpret_cast<f32x4*>(g.ws) +
                  ((((size_t)bz * (gridDim.x * gridDim.y) + by * gridDim.x + bx) * 4 + wave) * 25) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) slab[(i * 5 + j) * 64] = acc[i][j];
}

// ---------------------------------------------------------------------------------------------------------
// The same contraction at fp32 accuracy on the bf16 matrix pipe ("split" operands).  Every fp32 value is the EXACT sum of
// three bf16 values, x = x1 + x2 + x3 with x1 = rne(x), x2 = rne(x - x1), x3 = x - x1 - x2 (8 + 8 + 8 mantissa bits; both
// subtractions are exact), so a product a*b is the sum of nine bf16 x bf16 products, each exact in the fp32 accumulator.
// The three smallest (a2*b3, a3*b2, a3*b3: below 2^-24 of |a||b|, i.e. below the rounding of the fp32 product itself) are
// dropped; the other six are six v_mfma_f32_16x16x32_bf16 per fragment pair instead of eight v_mfma_f32_16x16x4_f32 --
// at 16x the rate, 2.7x less matrix-pipe time for the same sum to within fp32 rounding.  That makes the launch what the
// bf16 one above is: bound by the operand stream from HBM, not by the pipe (H = 300, K = 54 200: 107 us on fp32 MFMA,
// 49 us with rounded bf16 operands).  Structure as gemm_tn_tall_bf16; the LDS image holds three bf16 planes per operand
// (129 KB for the two buffers: one workgroup per CU, which is enough here because the loads of step s + 2 fly under the six
// products of step s).  Same fragment-order slabs, so tall_reduce serves all three kernels.
__global__ void __launch_bounds__(512, 1) gemm_tn_tall_split(GemmGroupArgs gg, int splits) {
    extern __shared__ __attribute__((aligned(16))) __bf16 Lsp[];      // [2 buffers][3 planes][BTK * BLD]
    constexpr int PLANE = BTK * BLD;
    const unsigned T = gridDim.x * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xq = T >> 3, xr = T & 7, xcd = L & 7;
    const unsigned logical = xcd * xq + min(xcd, xr) + (L >> 3);        // tiles of one K chunk share an XCD's L2
    const int bx = logical % gridDim.x, by = (logical / gridDim.x) % gridDim.y;
    const int bzz = logical / (gridDim.x * gridDim.y);
    const GemmArgs& g = gg.p[bzz / splits];
    const int bz = bzz % splits;
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Eight waves, two roles (wave w sits on SIMD w % 4: every SIMD holds one of each).  Waves 4-7 PRODUCE: they fetch the
    // operands of step s + 2, split the ones of step s + 1 into their three bf16 planes and write the LDS image.  Waves 0-3
    // CONSUME: one quadrant of the 160 x 160 tile each, fragments of step s out of the other buffer, six MFMAs per fragment
    // pair.  The vector instructions of the split (the bulk of a step's issue slots) are then issued by a wave that has no
    // MFMA of its own to wait for, beside the MFMA stream of its SIMD partner; one LDS-only barrier per step hands the buffers over.
    if (wave >= 4) {
        const int ptid = threadIdx.x - 256;
        const int isb = __builtin_amdgcn_readfirstlane(ptid >> 7);      // waves 4-5: A rows, waves 6-7: B rows
        const float* P = isb ? g.B : g.A;
        const int ld = isb ? g.ldb : g.lda, c0 = isb ? n0 : m0, climit = isb ? g.N : g.M;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(P), 0, kend > kbeg ? (unsigned)(((size_t)(kend - 1) * ld + climit) * 4) : 0u, 0x00020000);
        unsigned voff[10];
        int loff[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int f = (ptid & 127) + 128 * i;
            const int k = f / (TM / 4), c = (f % (TM / 4)) * 4;
            loff[i] = k * BLD + isb * TM + c;
            voff[i] = c0 + c < climit ? (unsigned)(((size_t)(kbeg + k) * ld + c0 + c) * 4) : 0xffffff00u;
        }
        const unsigned vstep = (unsigned)BTK * ld * 4;
        f32x4 r[10];
        auto fetch = [&]() {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
                if (voff[i] != 0xffffff00u) voff[i] += vstep;
            }
        };
        // x1 = rne(x) (v_cvt_pk_bf16_f32), r1 = x - x1 (exact, <= 16 significant bits, |r1| <= 2^-9 |x|), x2 = the upper half
        // of r1's bits (truncation: one v_and), r2 = r1 - x2 (exact, <= 8 significant bits: ITS upper half is x3, exactly)
        auto put = [&](int buf) {
            __bf16* base = Lsp + (size_t)buf * 3 * PLANE;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                gbf16x4 w1;
                unsigned r1b[4], r2b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = r[i][e];
                    w1[e] = (__bf16)x;
#if defined(GGPM_ABL_TALL) && GGPM_ABL_TALL == 1      // timing ablation (variant build): no residual arithmetic in the producers
                    r1b[e] = r2b[e] = __builtin_bit_cast(unsigned, x);
#else
                    const float r1 = x - (float)w1[e];
                    r1b[e] = __builtin_bit_cast(unsigned, r1);
                    const float x2f = __builtin_bit_cast(float, r1b[e] & 0xffff0000u);
                    r2b[e] = __builtin_bit_cast(unsigned, r1 - x2f);
#endif
                }
                // upper halves of two registers into one: bytes {lo.2, lo.3, hi.2, hi.3}
                const unsigned w2a = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), w2b = __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u);
                const unsigned w3a = __builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), w3b = __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u);
                *reinterpret_cast<gbf16x4*>(base + loff[i]) = w1;
                *reinterpret_cast<uint2*>(base + PLANE + loff[i]) = make_uint2(w2a, w2b);
                *reinterpret_cast<uint2*>(base + 2 * PLANE + loff[i]) = make_uint2(w3a, w3b);
            }
        };
        fetch();
        put(0);
        fetch();
        ggpm_lds_barrier();
        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += BTK, cur ^= 1) {
            put(cur ^ 1);           // rows of step s+1 (loaded a step ago); the consumers finished reading that buffer before
            fetch();                // the barrier that ended step s-1.  The loads of step s+2 fly under this step
            ggpm_lds_barrier();
        }
        return;
    }

    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fa0 = (4 * grp + q) * BLD + wm * 80 + 4 * pp;
    const int fb0 = (4 * grp + q) * BLD + TM + wn * 80 + 4 * pp;
    typedef __attribute__((address_space(3))) gs16x4* lds_v4;
    auto frag = [&](const __bf16* base) -> gbf16x8 {
        const gs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
        const gs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 16 * BLD));
        gs16x4 v[2] = {lo, hi};
        return __builtin_bit_cast(gbf16x8, v);
    };
    ggpm_lds_barrier();
    int cur = 0;
#if defined(GGPM_ABL_TALL) && GGPM_ABL_TALL == 2          // timing ablation (variant build): fragments read, no products
#define GGPM_TALL_MFMA(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define GGPM_TALL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
    for (int k0 = kbeg; k0 < kend; k0 += BTK, cur ^= 1) {
        const __bf16* img = Lsp + (size_t)cur * 3 * PLANE;
        gbf16x8 fb[3][5], fa[5];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[pl][j] = frag(img + pl * PLANE + fb0 + 16 * j);
        // smallest terms first: a3*b1; a2*b2, a2*b1; a1*b3, a1*b2, a1*b1 -- one A plane in registers at a time
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i] = frag(img + 2 * PLANE + fa0 + 16 * i);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = GGPM_TALL_MFMA(fa[i], fb[0][j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i] = frag(img + PLANE + fa0 + 16 * i);
#pragma unroll
        for (int pl = 1; pl >= 0; --pl)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = GGPM_TALL_MFMA(fa[i], fb[pl][j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 5; ++i) fa[i] = frag(img + fa0 + 16 * i);
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[i][j] = GGPM_TALL_MFMA(fa[i], fb[pl][j], acc[i][j]);
        ggpm_lds_barrier();
    }
    // split-K slab in the fragment order of gemm_tn_tall (lane l: rows 4*(l>>4) + 0..3, column l & 15 of every block)
    f32x4* slab = reinterpret_cast<f32x4*>(g.ws) +
                  ((((size_t)bz * (gridDim.x * gridDim.y) + by * gridDim.x + bx) * 4 + wave) * 25) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) slab[(i * 5 + j) * 64] = acc[i][j];
}
constexpr size_t GGPM_TALL_SPLIT_LDS = (size_t)2 * 3 * BTK * BLD * sizeof(__bf16);
// 0: fp32 MFMA for the fp32 tall contractions (the round-1 kernel); 1 (default): split operands on the bf16 pipe
inline int tall_split_mode() { static const int v = [] { const char* e = ggpm_dev_env("GGPM_TALL_SPLIT"); return e ? atoi(e) : 1; }(); return v; }
inline int tall_split_wgs() { static const int v = [] { const char* e = ggpm_dev_env("GGPM_TALL_SPLIT_WGS"); return e ? atoi(e) : 256; }(); return v; }
inline void launch_tall_split(const GemmGroupArgs& gg, dim3 grid, int splits, hipStream_t s) {
    ggpm_set_lds(gemm_tn_tall_split, GGPM_TALL_SPLIT_LDS);
    gemm_tn_tall_split<<<grid, 512, GGPM_TALL_SPLIT_LDS, s>>>(gg, splits);
}

// Sums the fragment-order slabs of gemm_tn_tall over the K chunks (fixed order: four interleaved partial sums per
// workgroup, combined as (0+1)+(2+3)) and writes C.  One workgroup per (tile, wave, 16 x 16 block).
__global__ void __launch_bounds__(256) tall_reduce(GemmGroupArgs gg, int splits, int tiles_n, int tiles) {
    const GemmArgs& g = gg.p[blockIdx.y];
    __shared__ f32x4 part[4][64];
    const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int blk = blockIdx.x % 25, wave = (blockIdx.x / 25) & 3, tile = blockIdx.x / 100;
    const f32x4* p = reinterpret_cast<const f32x4*>(g.ws) + (((size_t)tile * 4 + wave) * 25 + blk) * 64 + lane;
    const size_t zstride = (size_t)tiles * 4 * 25 * 64;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int z = zg; z < splits; z += 4) v += p[z * zstride];
    part[zg][lane] = v;
    __syncthreads();
    if (zg) return;
    v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int i = blk / 5, j = blk % 5;
    const int n = (tile % tiles_n) * TN + (wave & 1) * 80 + 16 * j + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = (tile / tiles_n) * TM + (wave >> 1) * 80 + 16 * i + 4 * (lane >> 4) + e;
        if (m >= g.M) continue;
        float* dst = g.C + (size_t)m * g.ldc + n;
        if (n < g.N) {
            float x = v[e];
            if (g.bias) x += g.bias[n];
            if (g.accumulate) x += *dst;
            x = apply_act(x, g.act);
            if (g.zero_row0 && m == 0) x = 0.f;
            *dst = x;
        } else if (n < g.n_pad) {
            *dst = 0.f;
        }
    }
}

// The tall kernel pays when the 160 x 160 tiles cover the output without much padding and K is long.
inline bool tall_shape(int M, int N, int K) {
    if (K < 6144 || M < 64 || N < 64) return false;     // measured: 300 x 300 x 2843 is faster on the 64 x 64 kernel
    const double cover = (double)ggpm_round_up(M, TM) * ggpm_round_up(N, TN);
    return (double)M * N >= 0.8 * cover;
}
inline size_t tall_slab_bytes(int M, int N) {
    return (size_t)ggpm_ceil_div(M, TM) * ggpm_ceil_div(N, TN) * 4 * 25 * 64 * sizeof(f32x4);
}
constexpr int GGPM_TALL_BF16_WGS = 768;      // workgroups of a bf16 tall launch (three per CU: it lives on overlapped loads)
inline int tall_splits(int M, int N, int K, int target_wgs = 0) {
    static const int target_env = [] { const char* e = ggpm_dev_env("GGPM_GEMM_TALL_WGS"); return e ? atoi(e) : 256; }();
    const int target = target_wgs > 0 ? target_wgs : target_env;
    const int tiles = ggpm_ceil_div(M, TM) * ggpm_ceil_div(N, TN);
    int s = target / tiles, maxs = K / (8 * TK);
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : s;
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
template <bool IN16>
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
        for (int m = lo + rl; m < hi; m += 4) {
            if constexpr (IN16) v += (float)reinterpret_cast<const __bf16*>(A)[(size_t)m * lda + n];
            else v += A[(size_t)m * lda + n];
        }
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

// short matrices (M <= CS_ONE_MAX rows: the bias gradients of the read-outs and of the small levels): ONE launch, block =
// 64 columns x 16 row lanes, fixed summation order -- the two-stage form spends two launch latencies on a few KB
constexpr int CS_ONE_MAX = 2048;
__global__ void __launch_bounds__(1024) colsum_one(const float* __restrict__ A, int lda, int M, int N,
                                                   float* __restrict__ out) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, n = blockIdx.x * 64 + c;
    const int rl = threadIdx.x >> 6;
    float v = 0.f;
    if (n < N)
        for (int m = rl; m < M; m += 16) v += A[(size_t)m * lda + n];
    red[rl][c] = v;
    __syncthreads();
    if (rl == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        out[n] = t;
    }
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

namespace {
// gemm_small_v3 when the 64 x 64 kernels would launch about one workgroup per CU or fewer (swept: 128 / 300 / 1024
// tiles -> 4.51 / 4.49 / 4.53 ms per GRU step)
inline bool small_launch(int M, int N, int count) {
    static const int use_v3 = [] { const char* e = ggpm_dev_env("GGPM_GEMM_V3"); return e ? atoi(e) : 1; }();
    static const int max_tiles = [] { const char* e = ggpm_dev_env("GGPM_GEMM_V3_TILES"); return e ? atoi(e) : 300; }();
    return use_v3 && (size_t)ggpm_ceil_div(M, BM) * ggpm_ceil_div(N, BN) * count <= (size_t)max_tiles;
}
inline void launch_small(int trans_a, int trans_b, const GemmGroupArgs& gg, int count, int M, int n_pad_max, hipStream_t s,
                         int splits = 1) {
    dim3 grid(ggpm_ceil_div(n_pad_max, SN), ggpm_ceil_div(M, SM), count * splits);
    if (!trans_a && !trans_b) gemm_small_v3<false, false><<<grid, 256, 0, s>>>(gg, splits);
    else if (!trans_a && trans_b) gemm_small_v3<false, true><<<grid, 256, 0, s>>>(gg, splits);
    else if (trans_a && !trans_b) gemm_small_v3<true, false><<<grid, 256, 0, s>>>(gg, splits);
    else gemm_small_v3<true, true><<<grid, 256, 0, s>>>(gg, splits);
}
// K chunks of a grouped small launch whose few 32 x 32 tiles would each walk a long K alone (the input halves of a level's
// gate weight gradients: 60 tiles x 2848 rows = 45 dependent k-steps, 177 us at the very end of the step): enough chunks
// for ~1000 workgroups, never shorter than four k-steps
inline int small_splits(int M, int N, int K, int count, int* k_chunk) {
    const int tiles = ggpm_ceil_div(M, SM) * ggpm_ceil_div(N, SN) * count;
    int want = 1024 / tiles, maxs = K / (4 * SK);
    int sp = want < maxs ? want : maxs;
    if (sp < 2) { *k_chunk = ggpm_round_up(K, BK); return 1; }
    *k_chunk = ggpm_round_up(ggpm_ceil_div(K, sp), SK);
    return ggpm_ceil_div(K, *k_chunk);
}
}  // namespace

extern "C" size_t ggpm_gemm_workspace_bytes(int M, int N, int K) {
    const int s = choose_splits(M, N, K);
    size_t bytes = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
    if (tall_shape(M, N, K)) {      // the caller's transposes are not known here: room for either kernel
        const int ts = max(tall_splits(M, N, K), tall_splits(M, N, K, GGPM_TALL_BF16_WGS));     // (either operand dtype)
        if (ts > 1) bytes = max(bytes, ts * tall_slab_bytes(M, N));
    }
    return bytes;
}

extern "C" int ggpm_gemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda,
                         const float* B, int ldb, float* C, int ldc, int n_pad, const float* bias,
                         int accumulate, int act, int zero_row0, float* splitk_ws, size_t splitk_ws_bytes,
                         ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || n_pad < N || n_pad > ldc) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.n_pad = n_pad; g.bias = bias; g.accumulate = accumulate; g.act = act; g.zero_row0 = zero_row0;
    g.vecA = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
    g.vecB = ((ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);
    static const int use_tall = [] { const char* e = ggpm_dev_env("GGPM_GEMM_TALL"); return e ? atoi(e) : 1; }();
    if (use_tall && trans_a && !trans_b && g.vecA && g.vecB && lda >= ggpm_round_up(M, 4) && ldb >= ggpm_round_up(N, 4) &&
        tall_shape(M, N, K) && n_pad <= ggpm_round_up(N, TN) && (size_t)K * lda * 4 < 0xffffff00ull && (size_t)K * ldb * 4 < 0xffffff00ull) {
        const size_t slab = tall_slab_bytes(M, N);
        if (tall_split_mode() && splitk_ws && splitk_ws_bytes >= slab) {
            // split operands on the bf16 pipe (fp32 accuracy, operand-stream bound): always through slabs + tall_reduce
            // (which applies bias / accumulate / activation)
            int sp = tall_splits(M, N, K, tall_split_wgs());
            if ((size_t)sp * slab > splitk_ws_bytes) sp = (int)(splitk_ws_bytes / slab);
            if (sp < 1) sp = 1;
            g.k_chunk = ggpm_round_up(ggpm_ceil_div(K, sp), BTK);
            sp = ggpm_ceil_div(K, g.k_chunk);
            g.ws = splitk_ws;
            const int tiles_n = ggpm_ceil_div(N, TN), tiles_m = ggpm_ceil_div(M, TM);
            GemmGroupArgs gg;
            for (int i = 0; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = g;
            launch_tall_split(gg, dim3(tiles_n, tiles_m, sp), sp, s);
            tall_reduce<<<tiles_n * tiles_m * 100, 256, 0, s>>>(gg, sp, tiles_n, tiles_n * tiles_m);
            GGPM_CHECK_LAUNCH();
            return GGPM_OK;
        }
        int splits = splitk_ws ? tall_splits(M, N, K) : 1;
        if (splits > 1 && (size_t)splits * slab > splitk_ws_bytes) splits = (int)(splitk_ws_bytes / slab);
        if (splits < 1) splits = 1;
        g.k_chunk = ggpm_round_up(ggpm_ceil_div(K, splits), TK);
        splits = ggpm_ceil_div(K, g.k_chunk);
        g.ws = splits > 1 ? splitk_ws : nullptr;
        const int tiles_n = ggpm_ceil_div(N, TN), tiles_m = ggpm_ceil_div(M, TM);
        dim3 grid(tiles_n, tiles_m, splits);
        static unsigned long long* dbg = [] {
            unsigned long long* p = nullptr;
            if (ggpm_dev_env("GGPM_GEMM_DEBUG")) (void)hipMalloc(&p, 64);
            return p;
        }();
        g.dbg = dbg;
        GemmGroupArgs gg;
        for (int i = 0; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = g;
        gemm_tn_tall<<<grid, 256, 0, s>>>(gg, splits);
        if (dbg) {
            unsigned long long h[6];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            const int steps = ggpm_ceil_div(g.k_chunk, 2 * TK) * 2;
            fprintf(stderr, "[gemm_tn_tall %dx%dx%d grid %dx%dx%d] prologue %.2f us, loop %.2f us = %d steps x %.0f cycles, "
                    "shader clock %.2f GHz\n", M, N, K, grid.x, grid.y, grid.z, (h[3] - h[1]) * 0.01, (h[5] - h[3]) * 0.01,
                    steps, (double)(h[4] - h[2]) / steps, (double)(h[4] - h[2]) / ((h[5] - h[3]) * 10.0));
        }
        if (splits > 1) tall_reduce<<<tiles_n * tiles_m * 100, 256, 0, s>>>(gg, splits, tiles_n, tiles_n * tiles_m);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
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
    static const int use_v2 = [] { const char* e = ggpm_dev_env("GGPM_GEMM_V2"); return e ? atoi(e) : 1; }();
    const size_t rows_a = trans_a ? K : M, rows_b = trans_b ? N : K;
    if (use_v2 && splits == 1 && small_launch(M, N, 1) && g.vecA && g.vecB && rows_a * lda * 4 < 0xffffff00ull &&
        rows_b * ldb * 4 < 0xffffff00ull) {
        GemmGroupArgs gg;
        for (int i = 0; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = g;
        launch_small(trans_a, trans_b, gg, 1, M, n_pad, s);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    // v2 keeps 70 KB of LDS per workgroup (two resident per CU): it wins while the whole grid is resident at once
    // (latency-bound launches: 573 x 600 x 912 35 -> 25 us) and loses beyond (2843 x 912 x 340: 33 -> 37 us)
    const bool resident = (size_t)grid.x * grid.y * grid.z <= 512 || use_v2 == 2;
    if (use_v2 && resident && g.vecA && g.vecB && rows_a * lda * 4 < 0xffffff00ull && rows_b * ldb * 4 < 0xffffff00ull) {
        if (!trans_a && !trans_b) gemm_kernel_v2<false, false><<<grid, 256, 0, s>>>(g);
        else if (!trans_a && trans_b) gemm_kernel_v2<false, true><<<grid, 256, 0, s>>>(g);
        else if (trans_a && !trans_b) gemm_kernel_v2<true, false><<<grid, 256, 0, s>>>(g);
        else gemm_kernel_v2<true, true><<<grid, 256, 0, s>>>(g);
    } else if (!trans_a && !trans_b) gemm_kernel<false, false><<<grid, 256, 0, s>>>(g);
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

namespace {
inline bool v2_operands_ok(const float* A, int lda, size_t rows_a, const float* B, int ldb, size_t rows_b) {
    static const int use_v2 = [] { const char* e = ggpm_dev_env("GGPM_GEMM_V2"); return e ? atoi(e) : 1; }();
    return use_v2 && (lda & 3) == 0 && (ldb & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
           rows_a * lda * 4 < 0xffffff00ull && rows_b * ldb * 4 < 0xffffff00ull;
}
inline void fill_args(GemmArgs& g, int M, int N, int K, const GgpmGemmProblem& p) {
    g = GemmArgs{};
    g.M = M; g.N = N; g.K = K; g.A = p.A; g.lda = p.lda; g.B = p.B; g.ldb = p.ldb; g.C = p.C; g.ldc = p.ldc;
    g.n_pad = p.n_pad; g.bias = p.bias; g.accumulate = p.accumulate; g.act = p.act; g.zero_row0 = p.zero_row0;
    g.vecA = g.vecB = 1;
    g.k_chunk = ggpm_round_up(K, BK);
}
}  // namespace

int ggpm_gemm_tall_grouped(int M, int N, int count, const GgpmGemmProblem* p, const int* K, float* ws, size_t ws_bytes,
                           ggpm_stream_t stream, int bf16) {
    GGPM_CLEAR_STALE_ERROR();
    if (count <= 0 || count > GGPM_GEMM_MAX_GROUP || !p || !K || M <= 0 || N <= 0) return GGPM_ERR_ARG;
    static const int use_tall = [] { const char* e = ggpm_dev_env("GGPM_GEMM_TALL"); return e ? atoi(e) : 1; }();
    const size_t slab = tall_slab_bytes(M, N);
    // (bf16 operands exist in the tall kernel only: a group that does not qualify falls back to fp32 products, which is
    // the more accurate side of the stated tolerance)
    const bool in16 = bf16 == 2;
    const size_t es = in16 ? 2 : 4;
    bool ok = use_tall && (count > 1 || bf16) && ws != nullptr;
    const bool split = !bf16 && tall_split_mode() != 0;      // fp32 accuracy on the bf16 pipe (gemm_tn_tall_split)
    int splits = 1 << 30;
    for (int i = 0; i < count && ok; ++i) {
        ok = (p[i].lda & 3) == 0 && (p[i].ldb & 3) == 0 && ((uintptr_t)p[i].A & 15) == 0 && ((uintptr_t)p[i].B & 15) == 0 &&
             p[i].lda >= ggpm_round_up(M, 4) && p[i].ldb >= ggpm_round_up(N, 4) && tall_shape(M, N, K[i]) &&
             p[i].n_pad <= ggpm_round_up(N, TN) && p[i].n_pad >= N && p[i].n_pad <= p[i].ldc &&
             (size_t)K[i] * p[i].lda * es < 0xffffff00ull && (size_t)K[i] * p[i].ldb * es < 0xffffff00ull;
        splits = min(splits, tall_splits(M, N, K[i], bf16 ? GGPM_TALL_BF16_WGS / count : (split ? tall_split_wgs() / count : 0)));
    }
    const int slabs_per_chunk = 1;
    if (ok) splits = min(splits, (int)(ws_bytes / (count * slab * slabs_per_chunk)));      // the group shares the workspace
    if (!ok || splits < ((bf16 || split) ? 1 : 2)) {
        if (in16) return GGPM_ERR_UNSUPPORTED;      // (bf16 operands in memory exist for the tall kernel only)
        for (int i = 0; i < count; ++i) {
            const int rc = ggpm_gemm(1, 0, M, N, K[i], p[i].A, p[i].lda, p[i].B, p[i].ldb, p[i].C, p[i].ldc, p[i].n_pad,
                                     p[i].bias, p[i].accumulate, p[i].act, p[i].zero_row0, ws, ws_bytes, stream);
            if (rc) return rc;
        }
        return GGPM_OK;
    }
    // one launch: the K chunks of all members share the grid (same number of chunks, chunk length per member), the
    // slabs of member i start at ws + i * splits * slab, one reduce launch sums them all
    GemmGroupArgs gg;
    for (int i = 0; i < count; ++i) {
        fill_args(gg.p[i], M, N, K[i], p[i]);
        gg.p[i].k_chunk = ggpm_round_up(ggpm_ceil_div(K[i], splits), (bf16 || split) ? BTK : TK);
        gg.p[i].ws = ws + (size_t)i * splits * slabs_per_chunk * (slab / sizeof(float));
    }
    for (int i = count; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = gg.p[0];
    hipStream_t s = (hipStream_t)stream;
    const int tiles_n = ggpm_ceil_div(N, TN), tiles_m = ggpm_ceil_div(M, TM);
    if (in16) gemm_tn_tall_bf16<true><<<dim3(tiles_n, tiles_m, splits * count), 256, 0, s>>>(gg, splits);
    else if (bf16) gemm_tn_tall_bf16<false><<<dim3(tiles_n, tiles_m, splits * count), 256, 0, s>>>(gg, splits);
    else if (split) launch_tall_split(gg, dim3(tiles_n, tiles_m, splits * count), splits, s);
    else gemm_tn_tall<<<dim3(tiles_n, tiles_m, splits * count), 256, 0, s>>>(gg, splits);
    tall_reduce<<<dim3(tiles_n * tiles_m * 100, count), 256, 0, s>>>(gg, splits * slabs_per_chunk, tiles_n, tiles_n * tiles_m);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

bool ggpm_bf16_storage_applies(int E1, int H) {
    static const int use_tall = [] { const char* e = ggpm_dev_env("GGPM_GEMM_TALL"); return e ? atoi(e) : 1; }();
    return use_tall && E1 >= 6144 && tall_shape(H, H, E1);
}
extern "C" int ggpm_level_bf16_storage(int E1, int H) { return ggpm_bf16_storage_applies(E1, H) ? 1 : 0; }

extern "C" int ggpm_gemm_tn_bf16_applies(int M, int N, int K) {
    static const int use_tall = [] { const char* e = ggpm_dev_env("GGPM_GEMM_TALL"); return e ? atoi(e) : 1; }();
    return use_tall && M > 0 && N > 0 && K > 0 && tall_shape(M, N, K) ? 1 : 0;
}

extern "C" int ggpm_gemm_tn_bf16(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                 float* ws, size_t ws_bytes, ggpm_stream_t stream) {
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0 || ldc < N) return GGPM_ERR_ARG;
    const GgpmGemmProblem p = {A, lda, B, ldb, C, ldc, N, nullptr, 0, GGPM_ACT_NONE, 0};
    return ggpm_gemm_tall_grouped(M, N, 1, &p, &K, ws, ws_bytes, stream, 1);
}

bool ggpm_gemm_prefers_grouped(int M, int N, int K, int count) {
    return ggpm_gemm_workspace_bytes(M, N, K) == 0 || (small_launch(M, N, count) && K <= 8192);
}

extern "C" int ggpm_gemm_grouped(int trans_a, int trans_b, int M, int N, int K, int count, const ggpm_gemm_problem* p,
                                 ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (count <= 0 || count > GGPM_GEMM_MAX_GROUP || !p || M <= 0 || N <= 0 || K <= 0) return GGPM_ERR_ARG;
    bool ok = count > 1;
    int n_pad_max = 0;
    for (int i = 0; i < count; ++i) {
        if (!p[i].A || !p[i].B || !p[i].C || p[i].n_pad < N || p[i].n_pad > p[i].ldc) return GGPM_ERR_ARG;
        ok = ok && v2_operands_ok(p[i].A, p[i].lda, trans_a ? K : M, p[i].B, p[i].ldb, trans_b ? N : K);
        n_pad_max = max(n_pad_max, p[i].n_pad);
    }
    if (!ok) {
        for (int i = 0; i < count; ++i) {
            const int rc = ggpm_gemm(trans_a, trans_b, M, N, K, p[i].A, p[i].lda, p[i].B, p[i].ldb, p[i].C, p[i].ldc,
                                     p[i].n_pad, p[i].bias, p[i].accumulate, p[i].act, p[i].zero_row0, nullptr, 0, stream);
            if (rc) return rc;
        }
        return GGPM_OK;
    }
    GemmGroupArgs gg;
    for (int i = 0; i < count; ++i) fill_args(gg.p[i], M, N, K, p[i]);
    for (int i = count; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = gg.p[0];
    hipStream_t s = (hipStream_t)stream;
    if (small_launch(M, N, count)) {
        launch_small(trans_a, trans_b, gg, count, M, n_pad_max, s);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    dim3 grid(ggpm_ceil_div(n_pad_max, BN), ggpm_ceil_div(M, BM), count);
    if (!trans_a && !trans_b) gemm_group_v2<false, false><<<grid, 256, 0, s>>>(gg);
    else if (!trans_a && trans_b) gemm_group_v2<false, true><<<grid, 256, 0, s>>>(gg);
    else if (trans_a && !trans_b) gemm_group_v2<true, false><<<grid, 256, 0, s>>>(gg);
    else gemm_group_v2<true, true><<<grid, 256, 0, s>>>(gg);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" size_t ggpm_gemm_grouped_splitk_workspace_bytes(int M, int N, int K, int count) {
    if (M <= 0 || N <= 0 || K <= 0 || count <= 0 || count > GGPM_GEMM_MAX_GROUP || !small_launch(M, N, count)) return 0;
    int kc;
    const int sp = small_splits(M, N, K, count, &kc);
    return sp > 1 ? (size_t)sp * count * M * N * sizeof(float) : 0;
}

extern "C" int ggpm_gemm_grouped_splitk(int trans_a, int trans_b, int M, int N, int K, int count, const ggpm_gemm_problem* p,
                                        float* ws, size_t ws_bytes, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (count <= 0 || count > GGPM_GEMM_MAX_GROUP || !p || M <= 0 || N <= 0 || K <= 0) return GGPM_ERR_ARG;
    const size_t need = ggpm_gemm_grouped_splitk_workspace_bytes(M, N, K, count);
    bool ok = need > 0 && ws && ws_bytes >= need;
    int n_pad_max = 0;
    for (int i = 0; i < count && ok; ++i) {
        if (!p[i].A || !p[i].B || !p[i].C || p[i].n_pad < N || p[i].n_pad > p[i].ldc) return GGPM_ERR_ARG;
        ok = v2_operands_ok(p[i].A, p[i].lda, trans_a ? K : M, p[i].B, p[i].ldb, trans_b ? N : K);
        n_pad_max = max(n_pad_max, p[i].n_pad);
    }
    if (!ok) return ggpm_gemm_grouped(trans_a, trans_b, M, N, K, count, p, stream);
    int kc;
    const int sp = small_splits(M, N, K, count, &kc);
    GemmGroupArgs gg;
    for (int i = 0; i < count; ++i) {
        fill_args(gg.p[i], M, N, K, p[i]);
        gg.p[i].k_chunk = kc;
        gg.p[i].ws = ws + (size_t)i * sp * M * N;
    }
    for (int i = count; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = gg.p[0];
    hipStream_t s = (hipStream_t)stream;
    launch_small(trans_a, trans_b, gg, count, M, N, s, sp);
    splitk_reduce_group<<<dim3(ggpm_ceil_div(n_pad_max, 256), M, count), 256, 0, s>>>(gg, sp);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_gemm_ksegments(int trans_b, int M, int N, int nseg, const float* const* A, const int* lda, const float* const* B,
                        const int* ldb, const int* K, float* C, int ldc, int n_pad, const float* bias, int accumulate,
                        int act, int zero_row0, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (nseg <= 0 || nseg > GGPM_GEMM_MAX_GROUP || !A || !lda || !B || !ldb || !K || !C || M <= 0 || N <= 0 || n_pad < N ||
        n_pad > ldc)
        return GGPM_ERR_ARG;
    bool ok = nseg > 1;
    for (int i = 0; i < nseg; ++i) {
        if (!A[i] || !B[i] || K[i] <= 0) return GGPM_ERR_ARG;
        ok = ok && v2_operands_ok(A[i], lda[i], M, B[i], ldb[i], trans_b ? N : K[i]);
    }
    if (!ok) {      // bias with the first product, activation / row mask with the last
        for (int i = 0; i < nseg; ++i) {
            const bool last = i == nseg - 1;
            const int rc = ggpm_gemm(0, trans_b, M, N, K[i], A[i], lda[i], B[i], ldb[i], C, ldc, i == 0 ? n_pad : N,
                                     i == 0 ? bias : nullptr, i == 0 ? accumulate : 1, last ? act : GGPM_ACT_NONE,
                                     last ? zero_row0 : 0, nullptr, 0, stream);
            if (rc) return rc;
        }
        return GGPM_OK;
    }
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K[0]; g.A = A[0]; g.lda = lda[0]; g.B = B[0]; g.ldb = ldb[0]; g.C = C; g.ldc = ldc; g.n_pad = n_pad;
    g.bias = bias; g.accumulate = accumulate; g.act = act; g.zero_row0 = zero_row0; g.vecA = g.vecB = 1;
    g.k_chunk = ggpm_round_up(K[0], BK);
    g.nseg = nseg;
    for (int i = 0; i < nseg; ++i) { g.segA[i] = A[i]; g.segB[i] = B[i]; g.seg_lda[i] = lda[i]; g.seg_ldb[i] = ldb[i]; g.segK[i] = K[i]; }
    hipStream_t s = (hipStream_t)stream;
    if (small_launch(M, N, 1)) {
        GemmGroupArgs gg;
        for (int i = 0; i < GGPM_GEMM_MAX_GROUP; ++i) gg.p[i] = g;
        launch_small(0, trans_b, gg, 1, M, n_pad, s);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    dim3 grid(ggpm_ceil_div(n_pad, BN), ggpm_ceil_div(M, BM), 1);
    if (trans_b) gemm_kernel_v2<false, true><<<grid, 256, 0, s>>>(g);
    else gemm_kernel_v2<false, false><<<grid, 256, 0, s>>>(g);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_colsum(const float* A, int lda, int M, int N, float* out, float* ws, ggpm_stream_t stream) {
    return ggpm_colsum_any(A, lda, M, N, out, ws, false, stream);
}

int ggpm_colsum_any(const float* A, int lda, int M, int N, float* out, float* ws, bool a_bf16, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!A || !out || !ws || M <= 0 || N <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (a_bf16) {                                // (large stashes only: always the two-stage form)
        int chunks = ggpm_ceil_div(M, 128);
        if (chunks > CS_ROWS) chunks = CS_ROWS;
        colsum_stage1<true><<<dim3(ggpm_ceil_div(N, 64), chunks), 256, 0, s>>>(A, lda, M, N, ws);
        colsum_stage2<<<ggpm_ceil_div(N, 64), 256, 0, s>>>(ws, N, chunks, out);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    if (M <= CS_ONE_MAX) {
        colsum_one<<<ggpm_ceil_div(N, 64), 1024, 0, s>>>(A, lda, M, N, out);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    int chunks = ggpm_ceil_div(M, 128);          // >= 128 rows per chunk, at most CS_ROWS chunks
    if (chunks > CS_ROWS) chunks = CS_ROWS;
    dim3 g1(ggpm_ceil_div(N, 64), chunks);
    colsum_stage1<false><<<g1, 256, 0, s>>>(A, lda, M, N, ws);
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

// ---- a batch of Linear weight gradients in ONE call ----------------------------------------------------------------------------
// Item i:  dW_i[N x K] = dpre_i[M x N]^T x_i[M x K]  (ggpm_gemm(1, 0, N, K, M, ...) with split-K over the M rows where it pays)
// and, when `db` is given, db_i[N] = column sums of dpre_i -- what autograd's mm / sum backward of a Linear computes
// (ggpm/decoder.py:35-58's heads, ggpm/encoder.py:15-19,62-72's read-outs) and what ggpm_amd/functional.py::_defer_flush used to
// issue as ~60 separate ctypes calls per backward pass (10-15 us of host time each).  Same launches, same order, same results;
// `ws` (ws_bytes >= the largest ggpm_gemm_workspace_bytes(N, K, M) of the batch, may be 0 / NULL: no split-K) and `csws`
// (256 * max N floats) are shared by the items, which run one after the other on `stream`.
extern "C" int ggpm_linear_wgrads_batch(int count, const ggpm_wgrad_item* items, float* ws, size_t ws_bytes, float* csws,
                                        ggpm_stream_t stream) {
    if (count < 0 || (count > 0 && !items)) return GGPM_ERR_ARG;
    for (int i = 0; i < count; ++i) {
        const ggpm_wgrad_item& it = items[i];
        if (!it.dpre || !it.x || !it.dW || it.M <= 0 || it.N <= 0 || it.K <= 0) return GGPM_ERR_ARG;
        const size_t need = ggpm_gemm_workspace_bytes(it.N, it.K, it.M);
        const bool split = ws && need && need <= ws_bytes;
        int rc = ggpm_gemm(1, 0, it.N, it.K, it.M, it.dpre, it.ld_dpre, it.x, it.ld_x, it.dW, it.ld_dw, it.K, nullptr, 0, GGPM_ACT_NONE, 0,
                           split ? ws : nullptr, split ? ws_bytes : 0, stream);
        if (rc) return rc;
        if (it.db) {
            if (!csws) return GGPM_ERR_ARG;
            rc = ggpm_colsum(it.dpre, it.ld_dpre, it.M, it.N, it.db, csws, stream);
            if (rc) return rc;
        }
    }
    return GGPM_OK;
}

