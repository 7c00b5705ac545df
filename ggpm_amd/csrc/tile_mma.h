// Workgroup-level "16 message rows x Hp features" tile algebra shared by the fused GRU / LSTM
// depth-step kernels.
//
// Geometry (gfx950, wave64): a workgroup of 4 waves owns R = 16 consecutive message rows for ALL
// Hp feature columns.  Activations live in LDS tiles [16][LD] (LD = Hp + 4 floats); weights are read
// straight from L2 in a pre-packed fragment order (each wave-instruction = one contiguous 1 KiB).
// The contraction runs on v_mfma_f32_16x16x4_f32 with the WEIGHT as the A operand and the
// activation tile as the B operand:
//      D[i][j] += sum_kk  W[out = 16*t + i][k]  *  X[row j][k]
// so that lane l ends up holding 4 consecutive output features (16*t + 4*(l>>4) + 0..3) of message
// row (l & 15): a float4 that lines up with the row-major feature matrices for the fused epilogues.
//
// k order inside a 16-wide chunk: MFMA step s (0..3) takes k = 16*kc + 4*(l>>4) + s from BOTH
// operands (each lane loads one float4 per chunk per operand); the MFMA sums over the four lane
// groups, so all 16 k of the chunk are covered after 4 steps.  MFMA f32 is an exact fmaf chain, so the
// summation order is fixed and results are run-to-run bitwise identical.
#pragma once
#include "common.h"

// Packed weight tile order: [out tile t][k chunk kc][lane 0..63][4 floats].
__device__ __forceinline__ size_t ggpm_pack_index(int t, int kc, int KC, int lane) {
    return (((size_t)t * KC + kc) * 64 + lane) * 4;
}

// Branch-free inner product for the N tiles a wave owns (t = wave + NW*i, all valid).
//   acc[i] += Wp(tile t) x tile^T  over KC chunks of 16 k.
// The weight fragments are software-prefetched PF chunks ahead through a register ring so that enough
// bytes are in flight per CU to stream the packed weights from L2 at the rate the MFMAs consume them
// (16 rows per workgroup = 8 FLOP per weight byte); the prefetch index is clamped instead of branched, so
// the loop body is straight-line: PF*N global_load_dwordx4 + 1 ds_read_b128 + 4*N MFMAs.
template <int N, int NW, int PF>
__device__ __forceinline__ void ggpm_tile_gemm_n(const float* __restrict__ tile, int LD,
                                                 const float* __restrict__ Wp, int KC, int wave, int lane,
                                                 f32x4* __restrict__ acc) {
    const float* brow = tile + (lane & 15) * LD + 4 * (lane >> 4);
    const float* wp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) wp[i] = Wp + ggpm_pack_index(wave + NW * i, 0, KC, lane);
    f32x4 ring[PF][N];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kk = min(d, KC - 1);
#pragma unroll
        for (int i = 0; i < N; ++i) ring[d][i] = *reinterpret_cast<const f32x4*>(wp[i] + (size_t)kk * 256);
    }
    int kc = 0;
    for (; kc + PF <= KC; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(brow + (kc + d) * 16);
            f32x4 a[N];
#pragma unroll
            for (int i = 0; i < N; ++i) a[i] = ring[d][i];
            const int kn = min(kc + d + PF, KC - 1);
#pragma unroll
            for (int i = 0; i < N; ++i) ring[d][i] = *reinterpret_cast<const f32x4*>(wp[i] + (size_t)kn * 256);
            __builtin_amdgcn_sched_barrier(0);   // keep the refill loads HERE (PF chunks ahead of their use)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < N; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[s], acc[i], 0, 0, 0);
            }
        }
    }
    // remainder (KC % PF chunks): their fragments already sit in ring[0 .. rem)
#pragma unroll
    for (int d = 0; d < PF - 1; ++d) {
        if (kc + d < KC) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(brow + (kc + d) * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < N; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[d][i][s], b[s], acc[i], 0, 0, 0);
            }
        }
    }
}

// Number of output tiles (16 columns each) wave `wave` of NW owns out of NT.
template <int NW>
__device__ __forceinline__ int ggpm_tiles_of_wave(int NT, int wave) {
    return wave < NT ? (NT - wave + NW - 1) / NW : 0;
}

// Dispatch on the (wave-uniform) tile count so that every instantiation is branch-free.
template <int TPW, int NW>
__device__ __forceinline__ void ggpm_tile_gemm(const float* __restrict__ tile, int LD,
                                               const float* __restrict__ Wp, int KC, int NT, int wave,
                                               int lane, f32x4 (&acc)[TPW]) {
    constexpr int PF = 4;
    const int n = ggpm_tiles_of_wave<NW>(NT, wave);
    if (n == 1) ggpm_tile_gemm_n<1, NW, PF>(tile, LD, Wp, KC, wave, lane, acc);
    if constexpr (TPW >= 2) { if (n == 2) ggpm_tile_gemm_n<2, NW, PF>(tile, LD, Wp, KC, wave, lane, acc); }
    if constexpr (TPW >= 3) { if (n == 3) ggpm_tile_gemm_n<3, NW, PF>(tile, LD, Wp, KC, wave, lane, acc); }
}

// ---- CSR row walk helpers for the gather phases -------------------------------------------------------
// One wave owns one destination row.  The row's list is loaded ONCE, coalesced (lane j holds entry j), and
// entries are then broadcast with v_readlane; slots past the end read index 0, the all-zero pad row of
// the reference layout, so the 4-way unrolled gather needs no branches and all its loads are independent.
struct GgpmRowList { int lo, n; };

__device__ __forceinline__ GgpmRowList ggpm_row_list(const int32_t* __restrict__ rowptr, int row, int rows) {
    GgpmRowList r;
    r.lo = 0; r.n = 0;
    if (row < rows) {
        r.lo = __builtin_amdgcn_readfirstlane(rowptr[row]);
        r.n = __builtin_amdgcn_readfirstlane(rowptr[row + 1]) - r.lo;
    }
    return r;
}

__device__ __forceinline__ int ggpm_list_chunk(const int32_t* __restrict__ col, GgpmRowList r, int base, int lane) {
    return (base + lane < r.n) ? col[r.lo + base + lane] : 0;
}

__device__ __forceinline__ int ggpm_list_at(int chunk, int j, int m) {
    return (j < m) ? __builtin_amdgcn_readlane(chunk, j) : 0;
}

template <int TPW>
__device__ __forceinline__ void ggpm_zero_acc(f32x4 (&acc)[TPW]) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float4 ggpm_f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// Pack W (or W^T) into fragment order, zero padded to Hp x Hp.
//   src(out, k) = transpose ? W[k*ldw + out] : W[out*ldw + k]     for out, k < H
__global__ void ggpm_pack_weight_kernel(const float* __restrict__ W, int ldw, int H, int Hp, int transpose,
                                        float* __restrict__ dst);
void ggpm_launch_pack(const float* W, int ldw, int H, int Hp, int transpose, float* dst, hipStream_t s);
