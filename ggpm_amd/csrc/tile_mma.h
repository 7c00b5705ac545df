// Workgroup-level "message rows x feature columns" tile algebra shared by the GRU / LSTM depth-step kernels.
//
// Geometry (gfx950, wave64).  A workgroup owns RT*16 consecutive message rows and a COLUMN GROUP of TG
// adjacent 16-wide output tiles (one tile per wave).  The grid is (row tiles) x (column groups) with TG chosen
// per level so that even the small motif/attachment levels (a few hundred messages) spread over all 256 CUs
// while the big atom level keeps the redundant full-row gathers at 2x.  Activations (the contraction operand, full K = Hp wide) sit in LDS tiles
// [RT*16][LD] (LD = Hp + 4 floats); weights are streamed straight from L2 in a pre-packed fragment order
// (one wave instruction = one contiguous 1 KiB) through a PF-deep register prefetch ring.
//
// The contraction runs on v_mfma_f32_16x16x4_f32 with the WEIGHT as the A operand and the activation
// tile as the B operand:   D[i][j] += sum_kk W[out = 16*t + i][k] * X[row j][k]
// so lane l ends up holding 4 consecutive output features (16*t + 4*(l>>4) + 0..3) of message row (l & 15):
// a float4 that lines up with the row-major feature matrices for the fused epilogues.
// k order inside a 16-wide chunk: MFMA step s (0..3) takes k = 16*kc + 4*(l>>4) + s from BOTH operands (one
// float4 per lane per chunk per operand).  MFMA f32 is an exact fmaf chain: the summation order is fixed by
// this code, so results are run-to-run bitwise identical.
#pragma once
#include "common.h"

constexpr int GGPM_NW = 4;        // waves per workgroup of the "B" kernels (one output tile per wave)
constexpr int GGPM_NWA = 16;      // waves per workgroup of the "A" kernels: all 16 gather, the first TG own a tile
constexpr int GGPM_PF = 4;        // weight-fragment prefetch depth (k chunks)

// Packed weight tile order: [out tile t][k chunk kc][lane 0..63][4 floats].
__device__ __forceinline__ size_t ggpm_pack_index(int t, int kc, int KC, int lane) {
    return (((size_t)t * KC + kc) * 64 + lane) * 4;
}

// acc[op][r] += Wp[op](tile t) x tile[op](row tile r)^T   for NOPS independent products sharing the k loop
// (e.g. the z and m gates of a GRU step).  Straight-line body: NOPS refill loads + NOPS*RT ds_read_b128 +
// 4*NOPS*RT MFMAs per k chunk; the refill index is clamped instead of branched.
template <int NOPS, int RT>
__device__ __forceinline__ void ggpm_wave_gemm(const float* const (&tiles)[NOPS], int LD,
                                               const float* const (&wps)[NOPS], int KC, int t, int lane,
                                               f32x4 (&acc)[NOPS][RT]) {
    constexpr int PF = GGPM_PF;       // (a deeper ring for the single-product loops measured no faster)
    const int boff = (lane & 15) * LD + 4 * (lane >> 4);
    const float* wp[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; ++o) wp[o] = wps[o] + ggpm_pack_index(t, 0, KC, lane);
    f32x4 ring[PF][NOPS];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kk = min(d, KC - 1);
#pragma unroll
        for (int o = 0; o < NOPS; ++o) ring[d][o] = *reinterpret_cast<const f32x4*>(wp[o] + (size_t)kk * 256);
    }
    int kc = 0;
    for (; kc + PF <= KC; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            f32x4 a[NOPS], b[NOPS][RT];
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                a[o] = ring[d][o];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    b[o][r] = *reinterpret_cast<const f32x4*>(tiles[o] + r * 16 * LD + boff + (kc + d) * 16);
            }
            const int kn = min(kc + d + PF, KC - 1);
#pragma unroll
            for (int o = 0; o < NOPS; ++o) ring[d][o] = *reinterpret_cast<const f32x4*>(wp[o] + (size_t)kn * 256);
            __builtin_amdgcn_sched_barrier(0);   // keep the refill loads HERE (PF chunks ahead of their use)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int o = 0; o < NOPS; ++o)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[o][s], b[o][r][s], acc[o][r], 0, 0, 0);
        }
    }
    // remainder (KC % PF chunks): their fragments already sit in ring[0 .. rem)
#pragma unroll
    for (int d = 0; d < PF - 1; ++d) {
        if (kc + d < KC) {
#pragma unroll
            for (int o = 0; o < NOPS; ++o)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tiles[o] + r * 16 * LD + boff + (kc + d) * 16);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[d][o][s], b[s], acc[o][r], 0, 0, 0);
                }
        }
    }
}

template <int NOPS, int RT>
__device__ __forceinline__ void ggpm_zero_acc(f32x4 (&acc)[NOPS][RT]) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[o][r] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float4 ggpm_f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// ---- CSR row walk helpers for the gather phases -------------------------------------------------------
// One wave walks one destination row.  The row's list is loaded ONCE, coalesced (lane j holds entry j), and
// entries are then broadcast with v_readlane; slots past the end read index 0, the all-zero pad row of
// the reference layout, so unrolled gathers need no branches and all their loads are independent.
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

// Optional 4-entry neighbour table (ggpm_csr_table4): lanes 0..3 load the row's entries with one 16-byte access that
// does not depend on rowptr; `fast` (wave uniform) tells whether the row fits the table.
__device__ __forceinline__ int ggpm_table_chunk(const int32_t* __restrict__ table, int row, int rows, int lane,
                                                bool& fast) {
    fast = false;
    if (!table) return 0;
    const int v = lane < 4 ? table[(size_t)(row < rows ? row : 0) * 4 + lane] : 0;
    fast = __builtin_amdgcn_readlane(v, 3) >= 0;
    return v;
}

__device__ __forceinline__ int ggpm_list_at(int chunk, int j, int m) {
    return (j < m) ? __builtin_amdgcn_readlane(chunk, j) : 0;
}

// Copy ROWS full feature rows [r0, r0+ROWS) of a [rows][Hp] matrix into an LDS tile [ROWS][LD]
// (rows past the end are zero filled).  All NW waves take part; 16 B per lane, coalesced.
template <int ROWS>
__device__ __forceinline__ void ggpm_load_rows_to_lds(const float* __restrict__ src, int r0, int rows, int Hp,
                                                      int LD, float* __restrict__ tile) {
    const int q = Hp >> 2;    // float4 per row
    for (int it = threadIdx.x; it < ROWS * q; it += blockDim.x) {
        const int lr = it / q, c = (it - lr * q) * 4;
        const int row = r0 + lr;
        const float4 v = row < rows ? ggpm_ld4(src + (size_t)row * Hp + c) : ggpm_zero4();
        ggpm_st4(tile + lr * LD + c, v);
    }
}

// Pack up to 4 gate matrices W (or W^T) into fragment order, zero padded to Hp x Hp, in ONE launch
// (blockIdx.z = matrix; matrix m lands at dst + m*Hp*Hp); optionally pads one bias vector to Hp.
//   src(out, k) = transpose ? W[k*ldw + out] : W[out*ldw + k]     for out, k < H
struct GgpmPackArgs {
    const float* W[4];
    int ldw[4];
    int H, Hp, transpose;
    float* dst;
    const float* bias;
    float* bias_out;
};
__global__ void ggpm_pack_weight_kernel(GgpmPackArgs a);
void ggpm_launch_pack(const GgpmPackArgs& a, int nmat, hipStream_t s);
// Output tiles per column group of the depth-step kernels for a level of E1 messages and NT = Hp/16 tiles.
// One 16-wave workgroup fits a CU at a time, so the grid is kept at <= ~256 workgroups: big levels use ONE
// group (no redundant gathers; waves loop over tiles wave, wave+16, ...), small levels split the columns so
// that a few hundred messages still reach all CUs.
static inline int ggpm_tiles_per_group(int E1, int NT) {
    const int row_tiles = (E1 + 15) / 16;
    int groups = 256 / row_tiles;
    if (groups < 1) groups = 1;
    int tg = (NT + groups - 1) / groups;
    if (tg < 4) tg = 4;
    if (tg > NT) tg = NT;
    return tg;
}
