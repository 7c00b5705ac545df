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
#include <type_traits>

constexpr int GGPM_NW = 4;        // waves per workgroup of the "B" kernels (one output tile per wave)
constexpr int GGPM_NWA = 16;      // waves per workgroup of the "A" kernels: all 16 gather, the first TG own a tile
constexpr int GGPM_PF = 4;        // weight-fragment prefetch depth (k chunks)
#ifndef GGPM_PF3
#define GGPM_PF3 3                // ... of the three-product loops (LSTM): 4 deep spilled 13-17 registers (48 -> 45.6 us)
#endif
#ifndef GGPM_CHAIN_MAX
#define GGPM_CHAIN_MAX 3          // loops with fewer products than this chain their ring across a wave's tiles
#endif
#ifndef GGPM_PF1
#define GGPM_PF1 GGPM_PF          // ... of the single-product loops
#endif
template <int NOPS> struct GgpmPf { static constexpr int value = NOPS >= 3 ? GGPM_PF3 : (NOPS == 1 ? GGPM_PF1 : GGPM_PF); };

// Packed weight tile order: [out tile t][k chunk kc][lane 0..63][4 floats].
__device__ __forceinline__ size_t ggpm_pack_index(int t, int kc, int KC, int lane) {
    return (((size_t)t * KC + kc) * 64 + lane) * 4;
}

// Weight-fragment prefetch ring of one wave: PF k chunks x NOPS products, one 16-byte fragment per lane each.
template <int NOPS>
struct GgpmRing {
    f32x4 r[GgpmPf<NOPS>::value][NOPS];
};

// Loads the first PF chunks of tile t into the ring.  The weights do not depend on anything the kernel computes, so a
// phase issues this BEFORE the barrier that publishes its activation tile: the L2 round trip (~1-2 us, otherwise
// exposed at the head of every phase) then runs while the wave waits for the slower waves of its workgroup.
template <int NOPS>
__device__ __forceinline__ void ggpm_ring_prefetch(const float* const (&wps)[NOPS], int KC, int t, int lane,
                                                   GgpmRing<NOPS>& ring) {
#pragma unroll
    for (int d = 0; d < GgpmPf<NOPS>::value; ++d) {
        const int kk = min(d, KC - 1);
#pragma unroll
        for (int o = 0; o < NOPS; ++o)
            ring.r[d][o] = *reinterpret_cast<const f32x4*>(wps[o] + ggpm_pack_index(t, kk, KC, lane));
    }
}

// acc[op][r] += Wp[op](tile t) x tile[op](row tile r)^T   for NOPS independent products sharing the k loop
// (e.g. the z and m gates of a GRU step).  Straight-line body: NOPS refill loads + NOPS*RT ds_read_b128 +
// 4*NOPS*RT MFMAs per k chunk.  `ring` must hold the first PF chunks of tile t (ggpm_ring_prefetch); refills past the
// end of tile t fetch the head of tile `t_next` (>= 0: the wave's next tile of the same products, whose call then
// finds its ring loaded) or re-read the last chunk (t_next < 0; no branch).
template <int NOPS, int RT, bool CHAIN = (NOPS < GGPM_CHAIN_MAX)>
__device__ __forceinline__ void ggpm_wave_gemm_ring(const float* const (&tiles)[NOPS], int LD,
                                                    const float* const (&wps)[NOPS], int KC, int t, int t_next, int lane,
                                                    f32x4 (&acc)[NOPS][RT], GgpmRing<NOPS>& ring) {
    constexpr int PF = GgpmPf<NOPS>::value;       // (a deeper ring for the single-product loops measured no faster)
    const int boff = (lane & 15) * LD + 4 * (lane >> 4);
    const float* wp[NOPS];
    const float* wn[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
        wp[o] = wps[o] + ggpm_pack_index(t, 0, KC, lane);
        wn[o] = wps[o] + ggpm_pack_index(t_next >= 0 ? t_next : t, 0, KC, lane);
    }
    const bool chain = CHAIN && t_next >= 0 && KC >= PF;      // (KC < PF: the loop below never refills: re-prefetch)
    int kc = 0;
    for (; kc + PF <= KC; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            f32x4 a[NOPS], b[NOPS][RT];
#pragma unroll
            for (int o = 0; o < NOPS; ++o) {
                a[o] = ring.r[d][o];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    b[o][r] = *reinterpret_cast<const f32x4*>(tiles[o] + r * 16 * LD + boff + (kc + d) * 16);
            }
            const int kq = kc + d + PF;
            const bool over = kq >= KC;
            const int kn = over ? (chain ? kq - KC : KC - 1) : kq;
#ifndef GGPM_ABL_NOLOAD            // (dev ablations, timing only: the weight stream / the matrix pipe alone)
#pragma unroll
            for (int o = 0; o < NOPS; ++o)
                ring.r[d][o] = *reinterpret_cast<const f32x4*>((over && chain ? wn[o] : wp[o]) + (size_t)kn * 256);
#else
            (void)kn; (void)wn;
#endif
            __builtin_amdgcn_sched_barrier(0);   // keep the refill loads HERE (PF chunks ahead of their use)
#ifndef GGPM_ABL_NOMFMA
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int o = 0; o < NOPS; ++o)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[o][s], b[o][r][s], acc[o][r], 0, 0, 0);
#else
#pragma unroll
            for (int o = 0; o < NOPS; ++o)
#pragma unroll
                for (int r = 0; r < RT; ++r) asm volatile("" :: "v"(a[o]), "v"(b[o][r]));
#endif
        }
    }
    // remainder (rem = KC % PF chunks): their fragments sit in ring slots [0, rem)
    const int rem = KC - kc;
#pragma unroll
    for (int d = 0; d < PF - 1; ++d) {
        if (d < rem) {
#pragma unroll
            for (int o = 0; o < NOPS; ++o)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(tiles[o] + r * 16 * LD + boff + (kc + d) * 16);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.r[d][o][s], b[s], acc[o][r], 0, 0, 0);
                }
        }
    }
    if (t_next < 0) return;
    if (!CHAIN || !chain) {
        ggpm_ring_prefetch<NOPS>(wps, KC, t_next, lane, ring);
        return;
    }
    // chained: the slots [rem, PF) already hold chunks 0 .. PF-rem-1 of tile t_next (refilled by the last trips of the
    // loop above); rotate them to the front and fetch the missing rem chunks, so that the ring looks as
    // ggpm_ring_prefetch(t_next) would have left it -- but with most of it loaded a whole tile ago
    auto rotate = [&](auto R) {
        constexpr int r0 = decltype(R)::value;
        if constexpr (r0 > 0) {
#pragma unroll
            for (int d = 0; d < PF; ++d)
#pragma unroll
                for (int o = 0; o < NOPS; ++o) {
                    if (d + r0 < PF) ring.r[d][o] = ring.r[d + r0][o];
                    else ring.r[d][o] = *reinterpret_cast<const f32x4*>(wn[o] + (size_t)d * 256);
                }
        }
    };
    switch (rem) {
        case 1: rotate(std::integral_constant<int, 1>{}); break;
        case 2: rotate(std::integral_constant<int, 2>{}); break;
        case 3: rotate(std::integral_constant<int, 3>{}); break;
        case 4: if constexpr (PF > 4) rotate(std::integral_constant<int, 4>{}); break;
        case 5: if constexpr (PF > 5) rotate(std::integral_constant<int, 5>{}); break;
        case 6: if constexpr (PF > 6) rotate(std::integral_constant<int, 6>{}); break;
        case 7: if constexpr (PF > 7) rotate(std::integral_constant<int, 7>{}); break;
        default: break;
    }
}

// One product set on one tile with the ring loaded on the spot (phases that have nothing to overlap the load with).
template <int NOPS, int RT>
__device__ __forceinline__ void ggpm_wave_gemm(const float* const (&tiles)[NOPS], int LD,
                                               const float* const (&wps)[NOPS], int KC, int t, int lane,
                                               f32x4 (&acc)[NOPS][RT]) {
    GgpmRing<NOPS> ring;
    ggpm_ring_prefetch<NOPS>(wps, KC, t, lane, ring);
    ggpm_wave_gemm_ring<NOPS, RT>(tiles, LD, wps, KC, t, -1, lane, acc, ring);
}

// ---- bf16 gate products (BASELINE configs[4]: "bf16" -- operands rounded to bf16, fp32 accumulate) -----------------------
// v_mfma_f32_16x16x32_bf16: one instruction contracts 32 k values; lane l supplies k = 32*kc + 8*(l>>4) + 0..7 of BOTH
// operands (weight row 16*t + (l&15), activation row l&15) and receives the same 4 output features of row (l&15) as
// the fp32 path, so the epilogues are shared.  Weights are packed once per call as bf16 in fragment order
// [out tile][k chunk of 32][lane][8] (16 bytes per lane per instruction, zero padded to a multiple of 32 columns);
// activations stay fp32 in the LDS tiles and are rounded (RNE, v_cvt_pk_bf16_f32) as they are read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ size_t ggpm_pack_index_bf16(int t, int kc, int KC32, int lane) {
    return (((size_t)t * KC32 + kc) * 64 + lane) * 8;      // in bf16 elements
}
static inline int ggpm_kc32(int Hp) { return (Hp + 31) / 32; }
__device__ __forceinline__ int ggpm_kc32_dev(int Hp) { return (Hp + 31) >> 5; }

template <int NOPS, int RT>
__device__ __forceinline__ void ggpm_wave_gemm_bf16(const float* const (&tiles)[NOPS], int LD,
                                                    const float* const (&wps_f32)[NOPS], int Hp, int t, int lane,
                                                    f32x4 (&acc)[NOPS][RT]) {
    constexpr int PF = 2;
    const int KC32 = ggpm_kc32_dev(Hp);
    const int boff = (lane & 15) * LD + 8 * (lane >> 4);
    const bf16x8* wp[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; ++o)
        wp[o] = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(wps_f32[o]) +
                                                 ggpm_pack_index_bf16(t, 0, KC32, lane));
    bf16x8 ring[PF][NOPS];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int o = 0; o < NOPS; ++o) ring[d][o] = wp[o][(size_t)min(d, KC32 - 1) * 64];
    for (int kc = 0; kc < KC32; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (kc + d < KC32) {
                const int k0 = 32 * (kc + d) + 8 * (lane >> 4);
                const bool live = k0 < Hp;              // Hp is a multiple of 16: a group of 8 columns is all in or all out
                bf16x8 a[NOPS], b[NOPS][RT];
#pragma unroll
                for (int o = 0; o < NOPS; ++o) {
                    a[o] = ring[d][o];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                        if (live) {
                            const float* src = tiles[o] + r * 16 * LD + boff + 32 * (kc + d);
                            lo = *reinterpret_cast<const f32x4*>(src);
                            hi = *reinterpret_cast<const f32x4*>(src + 4);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) { b[o][r][i] = (__bf16)lo[i]; b[o][r][4 + i] = (__bf16)hi[i]; }
                    }
                }
                const int kn = min(kc + d + PF, KC32 - 1);
#pragma unroll
                for (int o = 0; o < NOPS; ++o) ring[d][o] = wp[o][(size_t)kn * 64];
#pragma unroll
                for (int o = 0; o < NOPS; ++o)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[o][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[o], b[o][r], acc[o][r], 0, 0, 0);
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

// The same tile through an index: row r0 + i comes from src[idx[r0 + i]] (zeros when the id is negative); `copy`
// (optional) receives the gathered rows at their own position -- the start state of a sparse forward, fetched and
// materialised by the launch that first needs it.
template <int ROWS>
__device__ __forceinline__ void ggpm_gather_rows_to_lds(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                        int r0, int rows, int Hp, int LD, float* __restrict__ tile,
                                                        float* __restrict__ copy) {
    const int q = Hp >> 2;
    for (int it = threadIdx.x; it < ROWS * q; it += blockDim.x) {
        const int lr = it / q, c = (it - lr * q) * 4;
        const int row = r0 + lr;
        float4 v = ggpm_zero4();
        if (row < rows) {
            const int id = idx[row];
            if (id >= 0) v = ggpm_ld4(src + (size_t)id * Hp + c);
            if (copy) ggpm_st4(copy + (size_t)row * Hp + c, v);
        }
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
    int bf16;          // pack as bf16 fragments (ggpm_wave_gemm_bf16): matrix m at dst + m * Hp * 32 * kc32(Hp) bf16 elements
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
