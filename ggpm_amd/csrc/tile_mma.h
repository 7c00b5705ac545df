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
#ifndef GGPM_NWA_VALUE
#define GGPM_NWA_VALUE 16         // (variant builds: 8 -> two 8-wave workgroups per CU where the LDS tiles allow, 128 VGPRs each)
#endif
constexpr int GGPM_NWA = GGPM_NWA_VALUE;      // waves per workgroup of the "A" kernels: all gather, the first TG own a tile
#if GGPM_NWA_VALUE == 16
#define GGPM_A_BOUNDS __launch_bounds__(1024)
#else
#define GGPM_A_BOUNDS __launch_bounds__(GGPM_NWA_VALUE * 64, 4)
#endif
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

// ---- fp32-accurate gate products on the bf16 matrix pipe ("split" operands; gate mode 2) -----------------------------------
// Every fp32 value is the EXACT sum of three bf16 values (x1 = rne(x), x2 = the upper 8 significant bits of x - x1, x3 = the
// rest; gemm.hip: gemm_tn_tall_split), a product a*b the sum of nine bf16 x bf16 products that are exact in the fp32
// accumulator; the three below 2^-24 |a||b| are dropped, the other six are six v_mfma_f32_16x16x32_bf16 (16 cycles each)
// per 16 x 16 x 32 fragment pair instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each): 2.7x less matrix-pipe time for
// the same sum to within fp32 rounding.
//   weights     packed ONCE per call as three bf16 planes in fragment order [out tile][k chunk of 32][plane][lane][8]
//               (3 KiB contiguous per tile and chunk; zero padded to 32 * kc32 columns);
//   activations split where they are PRODUCED (gather / epilogue: once per element, not once per reading wave) into an LDS
//               image of three row-major bf16 planes [rows][LDH], LDH = Hp + 8 halves: the row pitch in dwords is
//               4 * (Hp / 8 + 1) -- an odd multiple of four, Hp being a multiple of 16 -- so 16 rows read at ONE k offset
//               land on 16 distinct bank quads.  (A ds_read_b128 lane group is {0-3, 12-15, 20-27}: eight of its rows are
//               read one quad further, and rows r / r+5 then share a quad for every padded pitch -- the 2-way conflict the
//               fp32 tile has as well; measured under the weight stream, profiles/r04_gru_sq_counters.txt.)  When Hp is not a multiple of 32 the last k
//               chunk's upper lanes (k = Hp .. Hp + 15) read the row's 8 pad halves and the first 8 halves of what follows
//               -- the next row, the next plane, or the 8-half gap that ends every image.  Their weights are zero; what
//               must not reach the pipe is a stale NaN bit pattern, so pads and gaps are zeroed once per launch
//               (ggpm_split_init) and everything else a fragment can touch is data of the same image.
// Summation: the five small products go into one accumulator, a1*b1 into another (added at the end): the small terms are not
// rounded against the large partial sums, and the two chains are independent for the pipe.
__device__ __forceinline__ int ggpm_split_ldh(int Hp) { return Hp + 8; }
static inline int ggpm_split_ldh_host(int Hp) { return Hp + 8; }
// one image = three planes + an 8-half gap; bytes / halves
static inline size_t ggpm_split_image_bytes(int rows, int Hp) { return ((size_t)3 * rows * ggpm_split_ldh_host(Hp) + 8) * 2; }
__device__ __forceinline__ int ggpm_split_image_halves(int rows, int Hp) { return 3 * rows * (Hp + 8) + 8; }
// floats per packed matrix, by gate mode (0 fp32 fragments, 1 bf16, 2 three bf16 planes); the largest is what callers reserve
__host__ __device__ static inline size_t ggpm_packed_matrix_floats(int Hp, int mode) {
    const size_t kc32 = (size_t)((Hp + 31) >> 5);
    return mode == 2 ? 48 * Hp * kc32 : mode == 1 ? 16 * Hp * kc32 : (size_t)Hp * Hp;
}
static inline size_t ggpm_packed_matrix_slot(int Hp) { return ggpm_packed_matrix_floats(Hp, 2); }      // >= every mode's
struct GgpmNoRing {};

__device__ __forceinline__ size_t ggpm_pack_index_split(int t, int kc, int KC32, int plane, int lane) {
    return ((((size_t)t * KC32 + kc) * 3 + plane) * 64 + lane) * 8;      // in bf16 elements
}

// x -> (x1, x2, x3) for four values; each plane as two dwords of packed bf16 (element 0 in the low half)
__device__ __forceinline__ void ggpm_split3(float4 x, uint2& p1, uint2& p2, uint2& p3) {
    const float xv[4] = {x.x, x.y, x.z, x.w};
    unsigned b1[4], r1b[4], r2b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)xv[e];                                   // RNE (v_cvt_pk_bf16_f32)
        b1[e] = (unsigned)__builtin_bit_cast(unsigned short, h);
        const float r1 = xv[e] - __builtin_bit_cast(float, b1[e] << 16);  // exact
        r1b[e] = __builtin_bit_cast(unsigned, r1);
        const float x2 = __builtin_bit_cast(float, r1b[e] & 0xffff0000u);
        r2b[e] = __builtin_bit_cast(unsigned, r1 - x2);                   // exact, <= 8 significant bits
    }
    p1 = make_uint2(b1[0] | (b1[1] << 16), b1[2] | (b1[3] << 16));
    p2 = make_uint2(__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u));
    p3 = make_uint2(__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u));
}

// store four consecutive columns [c, c+4) of local row lr into the three planes of an LDS image (plane stride PLANE halves)
__device__ __forceinline__ void ggpm_split_store(__bf16* img, int PLANE, int LDH, int lr, int c, float4 x) {
    uint2 p1, p2, p3;
    ggpm_split3(x, p1, p2, p3);
    __bf16* d = img + lr * LDH + c;
    *reinterpret_cast<uint2*>(d) = p1;
    *reinterpret_cast<uint2*>(d + PLANE) = p2;
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
}

// zero the row pads and the closing gaps of `nimg` consecutive images of `rows` rows (all threads of the workgroup take part;
// the data columns are written by the producers, the barrier that publishes them publishes these too)
__device__ __forceinline__ void ggpm_split_init(__bf16* img0, int nimg, int rows, int Hp) {
    const int LDH = Hp + 8, IMG = 3 * rows * LDH + 8;
    for (int it = threadIdx.x; it < nimg * (3 * rows + 1); it += blockDim.x) {
        const int im = it / (3 * rows + 1), r = it - im * (3 * rows + 1);
        __bf16* d = img0 + (size_t)im * IMG + (r < 3 * rows ? r * LDH + Hp : 3 * rows * LDH);
        *reinterpret_cast<uint4*>(d) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// Rows [r0, r0 + ROWS) of a [rows][Hp] fp32 matrix -> the three planes of an LDS image (B kernels); rows past the end: zeros.
template <int ROWS>
__device__ __forceinline__ void ggpm_load_rows_to_lds_split(const float* __restrict__ src, int r0, int rows, int Hp,
                                                            __bf16* img, int PLANE, int LDH) {
    const int q = Hp >> 2;
    for (int it = threadIdx.x; it < ROWS * q; it += blockDim.x) {
        const int lr = it / q, c = (it - lr * q) * 4;
        const int row = r0 + lr;
        const float4 v = row < rows ? ggpm_ld4(src + (size_t)row * Hp + c) : ggpm_zero4();
        ggpm_split_store(img, PLANE, LDH, lr, c, v);
    }
}

template <int ROWS>
__device__ __forceinline__ void ggpm_gather_rows_to_lds_split(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                              int r0, int rows, int Hp, __bf16* img, int PLANE, int LDH,
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
        ggpm_split_store(img, PLANE, LDH, lr, c, v);
    }
}

#ifndef GGPM_SPLIT_PF
#define GGPM_SPLIT_PF 2           // weight-plane ring depth in k chunks of 32 for the two-product loops (3 x 16 B per lane,
#endif                            // product and chunk); the one-product loops take twice as many chunks (same bytes in flight)
#ifndef GGPM_SPLIT_PF1
#define GGPM_SPLIT_PF1 2          // (3 deep spilled 9 registers in lstm_bwd_a, 4 deep 7 in gru_bwd_a)
#endif
#ifndef GGPM_SPLIT_PF3
#define GGPM_SPLIT_PF3 1          // three-product loops: 9 x 16 B per lane and chunk; 2 deep spilled 16-56 registers (the
#endif                            // load of the next chunk then lands under the other three waves of the SIMD)
template <int NOPS> struct GgpmSplitPf { static constexpr int value = NOPS == 1 ? GGPM_SPLIT_PF1 : NOPS >= 3 ? GGPM_SPLIT_PF3 : GGPM_SPLIT_PF; };
template <int NOPS>
struct GgpmSplitRing {
    bf16x8 r[GgpmSplitPf<NOPS>::value][NOPS][3];
};

template <int NOPS>
__device__ __forceinline__ void ggpm_split_ring_prefetch(const float* const (&wps_f32)[NOPS], int KC32, int t, int lane,
                                                         GgpmSplitRing<NOPS>& ring) {
    constexpr int PF = GgpmSplitPf<NOPS>::value;
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kk = min(d, KC32 - 1);
#pragma unroll
        for (int o = 0; o < NOPS; ++o)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                ring.r[d][o][p] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(wps_f32[o]) +
                                                                   ggpm_pack_index_split(t, kk, KC32, p, lane));
    }
}

// acc[op] += W[op](tile t) x image[op]^T over the full K.  `ring` holds the first PF chunks of tile t.  The MFMAs read the
// ring's registers themselves and the slot is refilled (chunk kc + PF) right behind them -- no second copy of the fragments
// in registers; the load then has PF - 1 chunk times to land.  Refills past the end of the tile fetch the head of tile
// t_next (>= 0, and only when KC32 is a multiple of PF: the ring then arrives loaded at the next call) or re-read the last
// chunk.  SHARED: every product reads the SAME image (the LSTM's three gates over s): its fragments are fetched once per
// chunk.  SUM: the products are summed (dS = sum_g dg_pre . W_g): one accumulator pair, acc[0], for all of them.
template <int NOPS, bool SHARED = false, bool SUM = false>
__device__ __forceinline__ void ggpm_wave_gemm_split(const __bf16* const (&imgs)[NOPS], int PLANE, int LDH,
                                                     const float* const (&wps_f32)[NOPS], int KC32, int t, int t_next,
                                                     int lane, f32x4 (&acc)[NOPS][1], GgpmSplitRing<NOPS>& ring) {
    constexpr int PF = GgpmSplitPf<NOPS>::value;
    constexpr int NACC = SUM ? 1 : NOPS;
#ifdef GGPM_ABL_LDS_BROADCAST      // timing ablation (wrong results): every lane reads row 0 of the image -- one address per 16-lane
    const int boff = 8 * (lane >> 4);      // group, no bank conflict possible: what the measured 46 % conflict ratio costs
#else
    const int boff = (lane & 15) * LDH + 8 * (lane >> 4);
#endif
    const bool chain = t_next >= 0 && (KC32 % PF) == 0;
    const __bf16* wp[NOPS];
    const __bf16* wn[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
        wp[o] = reinterpret_cast<const __bf16*>(wps_f32[o]) + ggpm_pack_index_split(t, 0, KC32, 0, lane);
        wn[o] = reinterpret_cast<const __bf16*>(wps_f32[o]) + ggpm_pack_index_split(chain ? t_next : t, 0, KC32, 0, lane);
    }
    f32x4 lo[NACC];
#pragma unroll
    for (int o = 0; o < NACC; ++o) lo[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < KC32; kc += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if (kc + d < KC32) {
                const int kq = kc + d + PF;
                const bool over = kq >= KC32;
                const int kn = over ? (chain ? kq - KC32 : KC32 - 1) : kq;
                bf16x8 b[3];
#pragma unroll
                for (int o = 0; o < NOPS; ++o) {
                    if (o == 0 || !SHARED) {
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            b[p] = *reinterpret_cast<const bf16x8*>(imgs[o] + p * PLANE + boff + 32 * (kc + d));
                    }
                    constexpr int Z = 0;
                    f32x4& hi_ = acc[SUM ? Z : o][0];
                    f32x4& lo_ = lo[SUM ? Z : o];
                    // smallest terms first: a3*b1, a1*b3, a2*b2, a2*b1, a1*b2 into `lo`; a1*b1 into the main accumulator
                    lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][2], b[0], lo_, 0, 0, 0);
                    lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][0], b[2], lo_, 0, 0, 0);
                    hi_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][0], b[0], hi_, 0, 0, 0);
                    lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][1], b[1], lo_, 0, 0, 0);
                    lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][1], b[0], lo_, 0, 0, 0);
                    lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring.r[d][o][0], b[1], lo_, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);      // the refills stay BEHIND the products that read the slot
#ifndef GGPM_ABL_NOLOAD_SPLIT      // (timing / traffic ablation: no weight-plane refills -- the ring keeps its first chunks)
#pragma unroll
                for (int o = 0; o < NOPS; ++o)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        ring.r[d][o][p] = *reinterpret_cast<const bf16x8*>((over && chain ? wn[o] : wp[o]) +
                                                                           ((size_t)kn * 3 + p) * 512);
#else
                (void)kn; (void)over;
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NACC; ++o) acc[o][0] = acc[o][0] + lo[o];
    if (t_next >= 0 && !chain) ggpm_split_ring_prefetch<NOPS>(wps_f32, KC32, t_next, lane, ring);
}

template <int NOPS, int RT>
__device__ __forceinline__ void ggpm_zero_acc(f32x4 (&acc)[NOPS][RT]) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[o][r] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float4 ggpm_f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// ---- bf16 STORAGE of the depth loop's arrays (gate mode 1 on large dense levels; BASELINE configs[4]) ----------------------
// An array kept in bf16 lives in the FIRST HALF of the fp32 buffer the caller reserved for it (same slot count, half the
// bytes), so no allocation, driver or binding changes with the storage type; values are rounded (RNE) where they are written
// and every reader -- the writing kernel's own epilogue included -- sees the rounded value (oracle/ref_encoder.py: "bf16s").
__device__ __forceinline__ float4 ggpm_bf16x4_to_f4(uint2 v) {
    return make_float4(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                       __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 ggpm_f4_to_bf16x4(float4 v) {      // RNE (v_cvt_pk_bf16_f32)
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    const b4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ float4 ggpm_rne4(float4 v) { return ggpm_bf16x4_to_f4(ggpm_f4_to_bf16x4(v)); }
// element i (a multiple of 4) of an array that is fp32 or, with B16, bf16 in the first half of the same buffer
template <bool B16>
__device__ __forceinline__ float4 ggpm_ldx(const float* base, size_t i) {
    if constexpr (B16) return ggpm_bf16x4_to_f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(base) + i));
    else return ggpm_ld4(base + i);
}
template <bool B16>
__device__ __forceinline__ void ggpm_stx(float* base, size_t i, float4 v) {
    if constexpr (B16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(base) + i) = ggpm_f4_to_bf16x4(v);
    else ggpm_st4(base + i, v);
}
// the same through a BYTE offset computed for fp32 elements (ggpm_ld4o / ggpm_st4o): halved for a bf16 array
template <bool B16>
__device__ __forceinline__ float4 ggpm_ldxo(const float* base, unsigned byte_off_f32) {
    if constexpr (B16)
        return ggpm_bf16x4_to_f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + (byte_off_f32 >> 1)));
    else return ggpm_ld4o(base, byte_off_f32);
}
template <bool B16>
__device__ __forceinline__ void ggpm_stxo(float* base, unsigned byte_off_f32, float4 v) {
    if constexpr (B16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + (byte_off_f32 >> 1)) = ggpm_f4_to_bf16x4(v);
    else ggpm_st4o(base, byte_off_f32, v);
}
// start of slot t (of `slot` elements) of such an array
static inline float* ggpm_slot_ptr(float* base, size_t t, size_t slot, bool b16) {
    return b16 ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(base) + t * slot) : base + t * slot;
}
static inline const float* ggpm_slot_ptr(const float* base, size_t t, size_t slot, bool b16) {
    return b16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(base) + t * slot) : base + t * slot;
}

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

__device__ __forceinline__ int ggpm_list_at(int chunk, int j, int m) {
    return (j < m) ? __builtin_amdgcn_readlane(chunk, j) : 0;
}

// Copy ROWS full feature rows [r0, r0+ROWS) of a [rows][Hp] matrix into an LDS tile [ROWS][LD]
// (rows past the end are zero filled).  All NW waves take part; 16 B per lane, coalesced.
template <int ROWS, bool B16 = false>
__device__ __forceinline__ void ggpm_load_rows_to_lds(const float* __restrict__ src, int r0, int rows, int Hp,
                                                      int LD, float* __restrict__ tile) {
    const int q = Hp >> 2;    // float4 per row
    for (int it = threadIdx.x; it < ROWS * q; it += blockDim.x) {
        const int lr = it / q, c = (it - lr * q) * 4;
        const int row = r0 + lr;
        const float4 v = row < rows ? ggpm_ldx<B16>(src, (size_t)row * Hp + c) : ggpm_zero4();
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
