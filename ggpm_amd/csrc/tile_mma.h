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

// acc[i] (i-th tile of this wave: t = wave + 4*i) += Wp(tile t) x tile^T  over KC chunks.
template <int TPW>
__device__ __forceinline__ void ggpm_tile_gemm(const float* __restrict__ tile, int LD,
                                               const float* __restrict__ Wp, int KC, int NT, int wave,
                                               int lane, f32x4 (&acc)[TPW]) {
    const float* brow = tile + (lane & 15) * LD + 4 * (lane >> 4);
    f32x4 a_cur[TPW], a_nxt[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + 4 * i;
        a_cur[i] = (t < NT) ? *reinterpret_cast<const f32x4*>(Wp + ggpm_pack_index(t, 0, KC, lane))
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int kc = 0; kc < KC; ++kc) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(brow + kc * 16);
        if (kc + 1 < KC) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + 4 * i;
                if (t < NT) a_nxt[i] = *reinterpret_cast<const f32x4*>(Wp + ggpm_pack_index(t, kc + 1, KC, lane));
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + 4 * i;
                if (t < NT) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i][s], b[s], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) a_cur[i] = a_nxt[i];
    }
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
