// Fused GRU message step for gfx950 -- the hot loop of the hierarchical encoder.
//
// Reference arithmetic: GRU.forward / GRU.GRU (ggpm/rnn.py:41-50, 25-39) + index_select_ND
// (ggpm/nnutils.py:65-70).  Restated over CSR predecessor lists with the x-halves of W_z/W_r/W_h hoisted
// out of the depth loop and U_r applied once per message (q = U_r h + b_u) instead of once per padded slot.
//
// Geometry (tile_mma.h): grid = (16-row message tiles) x (column groups of `tg` output tiles), 16 waves per workgroup.
// Per depth, forward:
//   kernel A  P1  CSR gather of predecessor rows (h_p, q_p) -> s, g tiles in LDS (one wave per row, 16 B per lane,
//                 hardware exp2/rcp sigmoid; the stash stores drain under the GEMM: LDS-only barrier)
//             P2  gate GEMMs on MFMA f32 16x16x4:  Wz_h . s  and  Wh_h . g  (weights streamed packed from L2)
//                 + fused gate math -> h'
//             P3  only when the level has ONE column group (the workgroup then holds the complete h' rows):
//                 q' = U_r h' + b_u from an LDS tile -- no second launch
//   kernel B  otherwise: q' = U_r h' + b_u with the h' rows re-read from L2
// Backward mirrors it (gather over SUCCESSORS through the transposed CSR, so no atomics):
//   kernel A  P1  dq, dh-partial tiles from successors   P2  dh = partial + dq.U_r ; gate derivatives
//             P3  (one column group) dG = dm_pre.Wh_h, dS = ds_dir + dz_pre.Wz_h, dXr += dG * R
//   kernel B  otherwise the same products from rows re-read from L2
// Weight gradients are three tall split-K GEMMs over the [depth*E1, Hp] stashes (gemm.hip), issued by the caller
// on a second stream.  sparse_forward (rows with a `frozen` mask) reuses the same kernels.
#include "tile_mma.h"
#include <cstdlib>
#include <cstdio>

__global__ void ggpm_pack_weight_kernel(GgpmPackArgs a) {
    const int Hp = a.Hp, H = a.H, KC = Hp / 16;
    const int lane = threadIdx.x;            // 64
    const int kc = blockIdx.x, t = blockIdx.y, m = blockIdx.z;
    const float* __restrict__ W = a.W[m];
    const int ldw = a.ldw[m];
    const int out = 16 * t + (lane & 15);
    if (a.bf16 == 2) {                       // three bf16 planes (ggpm_wave_gemm_split); grid.x = kc32(Hp) chunks of 32 columns
        const int KC32 = ggpm_kc32_dev(Hp);
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 32 * kc + 8 * (lane >> 4) + i;
            x[i] = (out < H && k < H) ? (a.transpose ? W[(size_t)k * ldw + out] : W[(size_t)out * ldw + k]) : 0.f;
        }
        uint2 lo[3], hi[3];
        ggpm_split3(make_float4(x[0], x[1], x[2], x[3]), lo[0], lo[1], lo[2]);
        ggpm_split3(make_float4(x[4], x[5], x[6], x[7]), hi[0], hi[1], hi[2]);
        __bf16* dst = reinterpret_cast<__bf16*>(a.dst) + (size_t)m * 2 * ggpm_packed_matrix_floats(Hp, 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(dst + ggpm_pack_index_split(t, kc, KC32, pl, lane)) = make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y);
        if (a.bias && m == 0 && t == 0 && kc * 64 + lane < Hp) {
            const int c = kc * 64 + lane;
            a.bias_out[c] = (c < H) ? a.bias[c] : 0.f;
        }
        return;
    }
    if (a.bf16) {                            // grid.x = kc32(Hp) chunks of 32 columns
        const int KC32 = ggpm_kc32_dev(Hp);
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 32 * kc + 8 * (lane >> 4) + i;
            float x = 0.f;
            if (out < H && k < H) x = a.transpose ? W[(size_t)k * ldw + out] : W[(size_t)out * ldw + k];
            v[i] = (__bf16)x;
        }
        __bf16* dst = reinterpret_cast<__bf16*>(a.dst) + (size_t)m * Hp * 32 * KC32;
        *reinterpret_cast<bf16x8*>(dst + ggpm_pack_index_bf16(t, kc, KC32, lane)) = v;
        if (a.bias && m == 0 && t == 0 && kc * 64 + lane < Hp) {
            const int c = kc * 64 + lane;
            a.bias_out[c] = (c < H) ? a.bias[c] : 0.f;
        }
        return;                              // (kc32 * 64 >= Hp: the blocks above cover every bias column)
    }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 16 * kc + 4 * (lane >> 4) + i;
        float x = 0.f;
        if (out < H && k < H) x = a.transpose ? W[(size_t)k * ldw + out] : W[(size_t)out * ldw + k];
        v[i] = x;
    }
    ggpm_st4(a.dst + (size_t)m * Hp * Hp + ggpm_pack_index(t, kc, KC, lane), make_float4(v[0], v[1], v[2], v[3]));
    // optional bias pad (once, by the first block row of matrix 0)
    if (a.bias && m == 0 && t == 0 && kc * 64 + lane < Hp) {
        const int c = kc * 64 + lane;
        a.bias_out[c] = (c < H) ? a.bias[c] : 0.f;
    }
}

void ggpm_launch_pack(const GgpmPackArgs& a, int nmat, hipStream_t s) {
    dim3 grid(a.bf16 ? ggpm_kc32(a.Hp) : a.Hp / 16, a.Hp / 16, nmat);
    ggpm_pack_weight_kernel<<<grid, 64, 0, s>>>(a);
}

namespace { thread_local int g_gate_dtype = 0; }
// 0: fp32 (gate products on split operands where the level's shape allows, see gate_mode below); 1: bf16 operands;
// 2: fp32 on v_mfma_f32_16x16x4_f32 only, 3: fp32 on split operands wherever they fit the LDS (the two forms every fp32
// call chooses between, selectable so that they can be compared on one shape)
void ggpm_set_gate_dtype(int dtype) { g_gate_dtype = (dtype >= 1 && dtype <= 3) ? dtype : 0; }
int ggpm_gate_dtype() { return g_gate_dtype; }
extern "C" int ggpm_level_gate_dtype(int dtype) {
    const int prev = g_gate_dtype;
    if (dtype >= 0 && dtype <= 3) g_gate_dtype = dtype;
    return prev;
}

namespace {

#ifndef GGPM_GATHER_U
#define GGPM_GATHER_U 2            // predecessor rows gathered per trip of the forward gather (null slots read row 0;
                                   // 4 per trip measured equal on the atom level, 1.3 us slower on the tree levels)
#endif
constexpr int RT = 1;              // row tiles (of 16 messages) per workgroup (default; the kernels take it as RTT)
constexpr int ROWS = RT * 16;
// RTT = 2 (32 message rows per workgroup): every weight fragment a wave streams from L2 then feeds TWO row tiles, which
// halves the weight stream of a level -- at H = 600 with ~950 row tiles (configs[4]) every workgroup reads all 3 Hp^2 packed
// weights per launch, 4.2 GB in total, and that stream, not the matrix pipe, bounds the launch.  Two tiles of 32 rows
// fill the LDS, so there is no room for the fused q' / dS,dG phase: those levels use the B launches.
constexpr int GGPM_RT2_MIN_ROW_TILES = 512;

struct GruFwdArgs {
    int E1, Hp, tg;                // tg: output tiles per column group of kernel A
    const float *Xz, *Xr, *Xh;
    const float *Hprev, *Qprev;
    float *Hnew, *Qnew;
    float *S, *G, *Z, *M, *R;      // stash slot of this depth (nullptr when not saving)
    const float *Wz, *Wh, *Ur;     // packed
    const float* bu;               // [Hp] zero padded
    const int32_t *rowptr, *col;
    int ablate;                    // timing experiments only (GGPM_ABLATE): 1 no gather, 2 no GEMM
    const unsigned char* frozen;   // sparse_forward only: rows with frozen[row] != 0 keep their state (h' = h)
    int fuse_b;                    // single column group: kernel A also forms q' = U_r h' + b_u (no B launch)
    unsigned long long* dbg;       // optional phase stamps of workgroup (0,0) (GGPM_ADEBUG; dev only)
    int bf16;                      // gate mode: 0 fp32 MFMA, 1 bf16 operands (packed weights are bf16 fragments), 2 fp32 on
                                   // split operands (three bf16 planes per operand, tile_mma.h)
    int st16;                      // bf16 storage of Hs / Qs / S / G / Z / M (gate mode 1, large dense training levels; tile_mma.h)
    int h0_zero;                   // first depth of a dense level: h^0 = 0, so s = g = 0 without a gather and the gate
                                   // products vanish (h^1 = sigmoid(x_z) tanh(x_h)); H^0 / Q^0 are neither built nor read
    float* Hout;                   // bf16 storage only: where the LAST depth also writes h' in fp32 (the level's result)
    const float* src_h;            // kernel B of a sparse forward's q^0 launch (ggpm_forward_gather_state): row r of the
    const int32_t* src_idx;        // start state is src_h[src_idx[r]] (zero when < 0); the launch writes it to Hnew itself
};

__global__ void pad_bias(const float* __restrict__ b, int H, int Hp, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Hp) out[c] = (c < H) ? b[c] : 0.f;
}

__device__ __forceinline__ float4 one_minus(float4 r) { return make_float4(1.f - r.x, 1.f - r.y, 1.f - r.z, 1.f - r.w); }

// ---------------------------------------------------------------------------------------------- forward
// Kernel A (16 waves): every wave gathers one message row at a time (full Hp width: two 256-column sweeps
// and 4 predecessor rows in flight -> 16 independent 16-byte loads per lane), then the first `tg` waves run
// the gate GEMMs of their output tile and the gate math.
template <bool STASH, int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS gru_fwd_a(GruFwdArgs a) {
    constexpr int ROWS = RTT * 16;
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    static_assert(!ST16 || GM == 1, "bf16 storage goes with bf16 gate products");      // Hs, Qs, S, G, Z, M in bf16 (tile_mma.h)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* Ts = lds;
    float* Tg = lds + ROWS * LD;
    // split operands: Ts stays (the epilogue reads s in fp32); behind it the bf16 images of s, g and (fused q') h'
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp);
    const int IMG = ggpm_split_image_halves(ROWS, Hp);
    __bf16* Is = reinterpret_cast<__bf16*>(lds + ROWS * LD);
    __bf16* Ig = Is + IMG;
    __bf16* Ih = Ig + IMG;
    if constexpr (SPLIT) ggpm_split_init(Is, a.fuse_b ? 3 : 2, ROWS, Hp);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t = grp * a.tg + wave;              // this wave's output tile (if wave < tg)
    const bool dbg_on = a.dbg && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0;
    if (dbg_on) a.dbg[0] = wall_clock64();
    const bool dbg15 = a.dbg && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 15 * 64;
    if (dbg15) a.dbg[5] = wall_clock64();

    // ---- P1: gather
    for (int lr = wave; lr < ROWS; lr += GGPM_NWA) {
        const int row = r0 + lr;
        GgpmRowList rl;
        if (a.h0_zero) { rl.lo = 0; rl.n = 0; } else rl = ggpm_row_list(a.rowptr, row, a.E1);
        if (a.ablate & 1) rl.n = 0;
        const size_t rowo = (size_t)(row < a.E1 ? row : 0) * Hp;
        for (int c0 = 0; c0 < Hp; c0 += 512) {
            int c[2], cs[2];
            bool on[2];
            float4 s[2], g[2], rc[2], xr[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                c[k] = c0 + 256 * k + lane * 4;
                on[k] = c[k] < Hp;
                cs[k] = on[k] ? c[k] : 0;      // lanes past the row end re-read column 0 (no branch), store nothing
                s[k] = ggpm_zero4(); g[k] = ggpm_zero4(); rc[k] = ggpm_zero4();
                xr[k] = ggpm_ld4(a.Xr + rowo + cs[k]);
            }
            for (int base = 0; base < rl.n; base += 64) {
                const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                const int m = min(64, rl.n - base);
                for (int j = 0; j < m; j += GGPM_GATHER_U) {
                    float4 h[GGPM_GATHER_U][2], q[GGPM_GATHER_U][2];
#pragma unroll
                    for (int u = 0; u < GGPM_GATHER_U; ++u) {
                        const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            h[u][k] = ggpm_ldx<ST16>(a.Hprev, p + cs[k]);
                            q[u][k] = ggpm_ldx<ST16>(a.Qprev, p + cs[k]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < GGPM_GATHER_U; ++u)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {          // null slots: h[0] == 0 contributes nothing
                            const float4 r = ggpm_fsigmoid4<BF16>(xr[k] + q[u][k]);
                            const float4 rh = r * h[u][k];
                            s[k] = s[k] + h[u][k];
                            g[k] = g[k] + rh;
                            rc[k] = rc[k] + rh * one_minus(r);
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!on[k]) continue;
                if constexpr (ST16) s[k] = ggpm_rne4(s[k]);      // what the stash holds is what the epilogue below uses
                ggpm_st4(Ts + lr * LD + c[k], s[k]);
                if constexpr (SPLIT) {
                    ggpm_split_store(Is, PLANE, LDH, lr, c[k], s[k]);
                    ggpm_split_store(Ig, PLANE, LDH, lr, c[k], g[k]);
                } else {
                    ggpm_st4(Tg + lr * LD + c[k], g[k]);
                }
                if (STASH && row < a.E1 && (c[k] >> 4) / a.tg == grp) {
                    const size_t o = (size_t)row * Hp + c[k];
                    ggpm_stx<ST16>(a.S, o, s[k]);
                    ggpm_stx<ST16>(a.G, o, g[k]);
                    ggpm_st4(a.R + o, rc[k]);     // sum_p h_p r(1-r): lets the backward form dXr without a gather
                }
            }
        }
    }

    if (dbg_on) a.dbg[1] = wall_clock64();
    if (dbg15) a.dbg[6] = wall_clock64();
    // the first weight fragments of P2 travel from L2 while this wave waits for the slower gatherers
    const int t_end = min(NT, (grp + 1) * a.tg);
    const bool p2_gemm = !(a.ablate & 2) && !a.h0_zero;
    const float* const wps2[2] = {a.Wz, a.Wh};
    std::conditional_t<GM == 0, GgpmRing<2>, GgpmNoRing> ring2;
    std::conditional_t<SPLIT, GgpmSplitRing<2>, GgpmNoRing> sring2;
    if constexpr (GM == 0)
        if (p2_gemm && t < t_end) ggpm_ring_prefetch<2>(wps2, KC, t, lane, ring2);
    if constexpr (SPLIT)
        if (p2_gemm && t < t_end) ggpm_split_ring_prefetch<2>(wps2, KC32, t, lane, sring2);
    ggpm_lds_barrier();      // LDS tiles only: the stash stores above finish under the GEMM
    if (dbg_on) a.dbg[2] = wall_clock64();

    // ---- P2: gate GEMMs + gate math for this wave's tiles (wave, wave+16, ... inside the column group)
    const int lr = lane & 15;
    for (int tt = t; tt < t_end; tt += GGPM_NWA) {
        const int c = 16 * tt + 4 * (lane >> 4);
        float4 xz[RTT], xh[RTT];
        auto load_inputs = [&]() {
#pragma unroll
            for (int r = 0; r < RTT; ++r) {
                const int row = r0 + 16 * r + lr;
                const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
                xz[r] = ggpm_ld4(a.Xz + o);
                xh[r] = ggpm_ld4(a.Xh + o);
            }
        };
        if constexpr (RTT == 1) load_inputs();      // in flight under the GEMM (two row tiles: the registers go to the GEMM)
        f32x4 acc[2][RTT];
        ggpm_zero_acc<2, RTT>(acc);
        if (p2_gemm) {
            const float* const tiles[2] = {Ts, Tg};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[2] = {Is, Ig};
                ggpm_wave_gemm_split<2>(imgs, PLANE, LDH, wps2, KC32, tt, tn, lane, acc, sring2);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<2, RTT>(tiles, LD, wps2, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<2, RTT>(tiles, LD, wps2, KC, tt, tn, lane, acc, ring2);
        }
        if constexpr (RTT != 1) load_inputs();
        if (dbg_on) a.dbg[3] = wall_clock64();
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int lrow = 16 * r + lr, row = r0 + lrow;
            const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
            float4 h = ggpm_zero4(), z = ggpm_zero4(), m = ggpm_zero4();
            auto keep_h = [&](float4 v) {          // the complete h' rows for the fused q' phase
                if constexpr (SPLIT) ggpm_split_store(Ih, PLANE, LDH, lrow, c, v);
                else ggpm_st4(lds + 2 * ROWS * LD + lrow * LD + c, v);
            };
            if (row >= a.E1) {
                if (a.fuse_b) keep_h(h);
                continue;
            }
            if (a.frozen && a.frozen[row]) {
                h = ggpm_ld4(a.Hprev + o);             // z = m = 0 in the stash => the backward passes dh through
            } else if (row != 0 || a.frozen) {
                const float4 s = ggpm_ld4(Ts + lrow * LD + c);
                const float4 pz = ggpm_f4(acc[0][r]) + xz[r], pm = ggpm_f4(acc[1][r]) + xh[r];
                z = ggpm_sigmoid4(pz);
                m = make_float4(tanhf(pm.x), tanhf(pm.y), tanhf(pm.z), tanhf(pm.w));
                if constexpr (ST16) { z = ggpm_rne4(z); m = ggpm_rne4(m); }
                h = make_float4((1.f - z.x) * s.x + z.x * m.x, (1.f - z.y) * s.y + z.y * m.y,
                                (1.f - z.z) * s.z + z.z * m.z, (1.f - z.w) * s.w + z.w * m.w);
                if constexpr (ST16) h = ggpm_rne4(h);
            }
            ggpm_stx<ST16>(a.Hnew, o, h);
            if constexpr (ST16) if (a.Hout) ggpm_st4(a.Hout + o, h);      // the level's result (last depth) also in fp32
            if (a.fuse_b) keep_h(h);
            if (STASH) {
                ggpm_stx<ST16>(a.Z, o, z);
                ggpm_stx<ST16>(a.M, o, m);
            }
        }
    }
    if (dbg_on) a.dbg[4] = wall_clock64();
    if (!a.fuse_b) return;

    // ---- P3 (single column group, RTT = 1 only): the workgroup holds the complete h' rows -> q' = U_r h' + b_u
    if constexpr (RTT == 1) {
        const int row = r0 + lr;
        const float* const wps3[1] = {a.Ur};
        std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring3;
        std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring3;
        if constexpr (GM == 0)
            if (wave < NT) ggpm_ring_prefetch<1>(wps3, KC, wave, lane, ring3);
        if constexpr (SPLIT)
            if (wave < NT) ggpm_split_ring_prefetch<1>(wps3, KC32, wave, lane, sring3);
        ggpm_lds_barrier();
        const float* Th = lds + 2 * ROWS * LD;
        for (int tt = wave; tt < NT; tt += GGPM_NWA) {
            const int c = 16 * tt + 4 * (lane >> 4);
            const float4 b = ggpm_ld4(a.bu + c);
            f32x4 acc[1][1];
            ggpm_zero_acc<1, 1>(acc);
            const float* const tiles[1] = {Th};
            const int tn = tt + GGPM_NWA < NT ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[1] = {Ih};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps3, KC32, tt, tn, lane, acc, sring3);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, 1>(tiles, LD, wps3, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, 1>(tiles, LD, wps3, KC, tt, tn, lane, acc, ring3);
            if (row < a.E1) ggpm_stx<ST16>(a.Qnew, (size_t)row * Hp + c, ggpm_f4(acc[0][0]) + b);
        }
    }
}

// Kernel B (same geometry as A): q' = U_r h' + b_u (h' rows come back from L2).
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS gru_fwd_b(GruFwdArgs a) {
    constexpr int ROWS = RTT * 16;
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* Th = lds;
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp);
    __bf16* Ih = reinterpret_cast<__bf16*>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps[1] = {a.Ur};
    std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring;
    std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring;
    if constexpr (GM == 0)
        if (grp * a.tg + wave < t_end) ggpm_ring_prefetch<1>(wps, KC, grp * a.tg + wave, lane, ring);     // under the row copy
    if constexpr (SPLIT) {
        if (grp * a.tg + wave < t_end) ggpm_split_ring_prefetch<1>(wps, KC32, grp * a.tg + wave, lane, sring);
        ggpm_split_init(Ih, 1, ROWS, Hp);
        if (a.src_idx) ggpm_gather_rows_to_lds_split<ROWS>(a.src_h, a.src_idx, r0, a.E1, Hp, Ih, PLANE, LDH, grp == 0 ? a.Hnew : nullptr);
        else ggpm_load_rows_to_lds_split<ROWS>(a.Hnew, r0, a.E1, Hp, Ih, PLANE, LDH);
    } else {
        if (a.src_idx) ggpm_gather_rows_to_lds<ROWS>(a.src_h, a.src_idx, r0, a.E1, Hp, LD, Th, grp == 0 ? a.Hnew : nullptr);
        else ggpm_load_rows_to_lds<ROWS, ST16>(a.Hnew, r0, a.E1, Hp, LD, Th);
    }
    __syncthreads();
    for (int tt = grp * a.tg + wave; tt < t_end; tt += GGPM_NWA) {
        const int c = 16 * tt + 4 * (lane >> 4);
        const float4 b = ggpm_ld4(a.bu + c);
        f32x4 acc[1][RTT];
        ggpm_zero_acc<1, RTT>(acc);
        if (!(a.ablate & 2)) {
            const float* const tiles[1] = {Th};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[1] = {Ih};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps, KC32, tt, tn, lane, acc, sring);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, RTT>(tiles, LD, wps, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, RTT>(tiles, LD, wps, KC, tt, tn, lane, acc, ring);
        }
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int row = r0 + 16 * r + (lane & 15);
            if (row < a.E1) ggpm_stx<ST16>(a.Qnew, (size_t)row * Hp + c, ggpm_f4(acc[0][r]) + b);
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
struct GruBwdArgs {
    int E1, Hp, tg;
    int first;                     // t == depth: dH comes from dHD, no successor gather
    const float* Xr;
    const float *Hcur, *Qcur;      // Hs[t],   Qs[t]      (kernel A gather; unused when first)
    const float *S, *Z, *M, *R;    // stash slot t-1
    const float* dHD;              // [E1,Hp], used when first
    const float *dSin, *dGin;      // from depth t+1
    float *dSout, *dGout;          // for depth t-1
    float* DQ;                     // stash slot for dq^t (nullptr when first)
    float *DMP, *DZP, *DSD;        // dm_pre / dz_pre stash slot t-1, ds_dir scratch
    float *dXz, *dXr, *dXh;        // running sums (zeroed by the driver)
    const float *WzT, *WhT, *UrT;  // packed transposes
    const int32_t *srowptr, *scol; // successors
    // sparse_forward only: frozen rows pass their gradient straight through the depth loop (carry), and a final
    // gather-only launch (t = 0) yields the gradient of the incoming state for them.
    const unsigned char* frozen;
    float* carry;                  // [E1,Hp] running dh of frozen rows (started by the first backward depth)
    int final_pass;                // t == 0: P1 + dq.U_r only, result to dHin
    float* dHin;                   // [E1,Hp]
    float* scat_h;                 // ggpm_backward_scatter_state: the final pass ADDS row r's result to
    const int32_t* scat_idx;       // scat_h[scat_idx[r]] (unique ids; < 0: dropped) instead of writing dHin
    int fuse_b;                    // single column group: kernel A also forms dS, dG for depth t-1 (no B launch)
    unsigned long long* dbg;       // optional phase stamps (GGPM_ADEBUG; dev only)
    int bf16;                      // gate products on bf16 operands
    int skip_xsum;                 // dXz / dXh are NOT accumulated here: the caller sums the DZP / DMP stash slots afterwards
    int st16;                      // bf16 storage of Hs / Qs / S / Z / M / dS / dG / DQ / DZP / DMP (see GruFwdArgs)
};

// Kernel A (16 waves): gather over successors (dq full rows, dh partial) -> dh = partial + dq.U_r ->
// gate derivatives for this workgroup's column group.
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS gru_bwd_a(GruBwdArgs a) {
    constexpr int ROWS = RTT * 16;      // ST16: Hs, Qs, S, Z, M, dS / dG, DQ, DZP, DMP in bf16 (tile_mma.h)
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* T0 = lds;                  // dh partial
    float* T1 = lds + ROWS * LD;      // dq
    // split operands: T0 stays fp32 (epilogue only); behind it the bf16 images of dq and (fused dS / dG) dz_pre, dm_pre
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp);
    const int IMG = ggpm_split_image_halves(ROWS, Hp);
    __bf16* I1 = reinterpret_cast<__bf16*>(lds + ROWS * LD);
    __bf16* Iz = I1 + IMG;
    __bf16* Im = Iz + IMG;
    if constexpr (SPLIT) ggpm_split_init(I1, a.fuse_b ? 3 : 1, ROWS, Hp);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t = grp * a.tg + wave;

    const bool dbg_on = a.dbg && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0;
    if (dbg_on) a.dbg[0] = wall_clock64();
    // ---- P1: dh_p += dS_e + dG_e*r ; dq_p += dG_e * h_p * r(1-r) over successors e (null slots: row 0,
    // where dS = dG = 0)
    if (!a.first) {
        for (int lr = wave; lr < ROWS; lr += GGPM_NWA) {
            const int p = r0 + lr;
            const GgpmRowList rl = ggpm_row_list(a.srowptr, p, a.E1);
            const size_t po = (size_t)(p < a.E1 ? p : 0) * Hp;
            for (int c0 = 0; c0 < Hp; c0 += 512) {
                int c[2], cs[2];
                bool on[2];
                float4 dh[2], dq[2], hp[2], qp[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    c[k] = c0 + 256 * k + lane * 4;
                    on[k] = c[k] < Hp;
                    cs[k] = on[k] ? c[k] : 0;
                    dh[k] = ggpm_zero4(); dq[k] = ggpm_zero4();
                    hp[k] = ggpm_ldx<ST16>(a.Hcur, po + cs[k]);
                    qp[k] = ggpm_ldx<ST16>(a.Qcur, po + cs[k]);
                }
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.scol, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 2) {
                        float4 xr[2][2], dg[2][2], ds[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const size_t e = (size_t)ggpm_list_at(chunk, j + u, m) * Hp;
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                xr[u][k] = ggpm_ld4(a.Xr + e + cs[k]);
                                dg[u][k] = ggpm_ldx<ST16>(a.dGin, e + cs[k]);
                                ds[u][k] = ggpm_ldx<ST16>(a.dSin, e + cs[k]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float4 r = ggpm_fsigmoid4<BF16>(xr[u][k] + qp[k]);
                                const float4 dgr = dg[u][k] * r;
                                dh[k] = dh[k] + ds[u][k] + dgr;
                                dq[k] = dq[k] + dgr * hp[k] * one_minus(r);
                            }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (!on[k]) continue;
                    ggpm_st4(T0 + lr * LD + c[k], dh[k]);
                    if constexpr (SPLIT) ggpm_split_store(I1, PLANE, LDH, lr, c[k], dq[k]);
                    else ggpm_st4(T1 + lr * LD + c[k], dq[k]);
                    if (p < a.E1 && (c[k] >> 4) / a.tg == grp) ggpm_stx<ST16>(a.DQ, (size_t)p * Hp + c[k], dq[k]);
                }
            }
        }
    }

    if (dbg_on) a.dbg[1] = wall_clock64();
    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps2[1] = {a.UrT};
    std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring2;
    std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring2;
    if constexpr (GM == 0)
        if (!a.first && t < t_end) ggpm_ring_prefetch<1>(wps2, KC, t, lane, ring2);      // under the wait for the gatherers
    if constexpr (SPLIT)
        if (!a.first && t < t_end) ggpm_split_ring_prefetch<1>(wps2, KC32, t, lane, sring2);
    if (!a.first) ggpm_lds_barrier();      // LDS tiles only: the dq stash stores finish under the GEMM
    if (dbg_on) a.dbg[2] = wall_clock64();

    // ---- P2: dh = partial + dq . U_r ; gate derivatives, for this wave's tiles
    const int lr = lane & 15;
    float4 dsd_keep[2] = {ggpm_zero4(), ggpm_zero4()};      // fused P3 (RTT = 1): ds_dir of this wave's (at most two) tiles
    int it = 0;
    for (int tt = t; tt < t_end; tt += GGPM_NWA, ++it) {
        const int c = 16 * tt + 4 * (lane >> 4);
        float4 s[RTT], z[RTT], m[RTT], oxz[RTT], oxh[RTT], dhd[RTT];
        auto load_inputs = [&]() {
#pragma unroll
            for (int r = 0; r < RTT; ++r) {
                const int row = r0 + 16 * r + lr;
                const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
                s[r] = z[r] = m[r] = oxz[r] = oxh[r] = ggpm_zero4();
                if (!a.final_pass) {
                    s[r] = ggpm_ldx<ST16>(a.S, o); z[r] = ggpm_ldx<ST16>(a.Z, o); m[r] = ggpm_ldx<ST16>(a.M, o);
                    if (!a.first && !a.skip_xsum) { oxz[r] = ggpm_ld4(a.dXz + o); oxh[r] = ggpm_ld4(a.dXh + o); }    // depth D starts the sums
                }
                dhd[r] = a.first ? ggpm_ld4(a.dHD + o) : ggpm_zero4();
            }
        };
        if constexpr (RTT == 1) load_inputs();      // in flight under the GEMM (two row tiles: the registers go to the GEMM)
        f32x4 acc[1][RTT];
        ggpm_zero_acc<1, RTT>(acc);
        if (!a.first) {
            const float* const tiles[1] = {T1};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[1] = {I1};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps2, KC32, tt, tn, lane, acc, sring2);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, RTT>(tiles, LD, wps2, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, RTT>(tiles, LD, wps2, KC, tt, tn, lane, acc, ring2);
        }
        if constexpr (RTT != 1) load_inputs();
        if (dbg_on) a.dbg[3] = wall_clock64();
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int lrow = 16 * r + lr, row = r0 + lrow;
            const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
            auto keep_zm = [&](float4 vz, float4 vm) {      // the complete dz_pre / dm_pre rows for the fused dS / dG phase
                if constexpr (SPLIT) {
                    ggpm_split_store(Iz, PLANE, LDH, lrow, c, vz);
                    ggpm_split_store(Im, PLANE, LDH, lrow, c, vm);
                } else {
                    ggpm_st4(lds + 2 * ROWS * LD + lrow * LD + c, vz);
                    ggpm_st4(lds + 3 * ROWS * LD + lrow * LD + c, vm);
                }
            };
            if (row >= a.E1) {
                if (a.fuse_b) keep_zm(ggpm_zero4(), ggpm_zero4());
                continue;
            }
            const bool frz = a.frozen && a.frozen[row];
            if (a.final_pass) {        // gradient of the incoming state: frozen rows only (active rows started from 0)
                float4 dh0 = ggpm_zero4();
                if (frz) dh0 = ggpm_f4(acc[0][r]) + ggpm_ld4(T0 + lrow * LD + c) + ggpm_ld4(a.carry + o);
                if (a.scat_idx) {
                    const int id = frz ? a.scat_idx[row] : -1;
                    if (id >= 0) {
                        float* d = a.scat_h + (size_t)id * Hp + c;
                        ggpm_st4(d, ggpm_ld4(d) + dh0);
                    }
                } else {
                    ggpm_st4(a.dHin + o, dh0);
                }
                continue;
            }
            float4 dsdir = ggpm_zero4(), dzp = ggpm_zero4(), dmp = ggpm_zero4();
            if (row != 0 || a.frozen) {
                float4 dh = a.first ? dhd[r] : (ggpm_f4(acc[0][r]) + ggpm_ld4(T0 + lrow * LD + c));
                if (frz) {             // h_t = h_{t-1} for frozen rows: carry the whole dh to the previous depth
                    if (!a.first) dh = dh + ggpm_ld4(a.carry + o);      // (the first backward depth starts the carry: no memset)
                    ggpm_st4(a.carry + o, dh);
                    dh = ggpm_zero4();   // nothing flows through gates (their stash is 0 anyway)
                }
                const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, sv[4] = {s[r].x, s[r].y, s[r].z, s[r].w};
                const float zv[4] = {z[r].x, z[r].y, z[r].z, z[r].w}, mv[4] = {m[r].x, m[r].y, m[r].z, m[r].w};
                float o_ds[4], o_dz[4], o_dm[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o_ds[k] = dhv[k] * (1.f - zv[k]);
                    o_dz[k] = dhv[k] * (mv[k] - sv[k]) * zv[k] * (1.f - zv[k]);
                    o_dm[k] = dhv[k] * zv[k] * (1.f - mv[k] * mv[k]);
                }
                dsdir = make_float4(o_ds[0], o_ds[1], o_ds[2], o_ds[3]);
                dzp = make_float4(o_dz[0], o_dz[1], o_dz[2], o_dz[3]);
                dmp = make_float4(o_dm[0], o_dm[1], o_dm[2], o_dm[3]);
                if constexpr (ST16) { dzp = ggpm_rne4(dzp); dmp = ggpm_rne4(dmp); }      // (the stored values: every user sees them)
            }
            ggpm_stx<ST16>(a.DZP, o, dzp);
            ggpm_stx<ST16>(a.DMP, o, dmp);
            if (!a.skip_xsum) {
                ggpm_st4(a.dXz + o, oxz[r] + dzp);
                ggpm_st4(a.dXh + o, oxh[r] + dmp);
            }
            if (a.fuse_b) {
                keep_zm(dzp, dmp);
                if (it == 0) dsd_keep[0] = dsdir; else dsd_keep[1] = dsdir;
            } else {
                ggpm_st4(a.DSD + o, dsdir);
            }
        }
    }
    if (dbg_on) a.dbg[4] = wall_clock64();
    if (!a.fuse_b) return;

    // ---- P3 (single column group, RTT = 1 only): the workgroup holds the complete dz_pre / dm_pre rows ->
    // dG = dm_pre . Wh_h ; dS = ds_dir + dz_pre . Wz_h ; dXr += dG * R   (the body of kernel B)
    if constexpr (RTT == 1) {
        const int row = r0 + lr;
        const float* const wps3[2] = {a.WhT, a.WzT};
        std::conditional_t<GM == 0, GgpmRing<2>, GgpmNoRing> ring3;
        std::conditional_t<SPLIT, GgpmSplitRing<2>, GgpmNoRing> sring3;
        if constexpr (GM == 0)
            if (wave < NT) ggpm_ring_prefetch<2>(wps3, KC, wave, lane, ring3);
        if constexpr (SPLIT)
            if (wave < NT) ggpm_split_ring_prefetch<2>(wps3, KC32, wave, lane, sring3);
        ggpm_lds_barrier();
        it = 0;
        for (int tt = wave; tt < NT; tt += GGPM_NWA, ++it) {
            const int c = 16 * tt + 4 * (lane >> 4);
            const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
            const float4 dsd = it == 0 ? dsd_keep[0] : dsd_keep[1];
            const float4 rco = ggpm_ld4(a.R + o), oxr = a.first ? ggpm_zero4() : ggpm_ld4(a.dXr + o);
            f32x4 acc[2][1];
            ggpm_zero_acc<2, 1>(acc);
            {
                const float* const tiles[2] = {lds + 3 * ROWS * LD, lds + 2 * ROWS * LD};
                const int tn = tt + GGPM_NWA < NT ? tt + GGPM_NWA : -1;
                if constexpr (SPLIT) {
                    const __bf16* const imgs[2] = {Im, Iz};
                    ggpm_wave_gemm_split<2>(imgs, PLANE, LDH, wps3, KC32, tt, tn, lane, acc, sring3);
                } else if constexpr (BF16) ggpm_wave_gemm_bf16<2, 1>(tiles, LD, wps3, Hp, tt, lane, acc);
                else ggpm_wave_gemm_ring<2, 1>(tiles, LD, wps3, KC, tt, tn, lane, acc, ring3);
            }
            if (row >= a.E1) continue;
            float4 dg = ggpm_f4(acc[0][0]);
            if constexpr (ST16) dg = ggpm_rne4(dg);
            ggpm_stx<ST16>(a.dGout, o, dg);
            ggpm_stx<ST16>(a.dSout, o, ggpm_f4(acc[1][0]) + dsd);
            ggpm_st4(a.dXr + o, oxr + dg * rco);
        }
    }
}

// Kernel B (same geometry as A): dG = dm_pre . Wh_h ; dS = ds_dir + dz_pre . Wz_h (for depth t-1) ;
// dXr += dG * R with R = sum_p h_p r(1-r) stashed by the forward gather.
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS gru_bwd_b(GruBwdArgs a) {
    constexpr int ROWS = RTT * 16;
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* T1 = lds;                  // dz_pre rows
    float* T2 = lds + ROWS * LD;      // dm_pre rows
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp);
    __bf16* Iz = reinterpret_cast<__bf16*>(lds);
    __bf16* Im = Iz + ggpm_split_image_halves(ROWS, Hp);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps[2] = {a.WhT, a.WzT};
    std::conditional_t<GM == 0, GgpmRing<2>, GgpmNoRing> ring;
    std::conditional_t<SPLIT, GgpmSplitRing<2>, GgpmNoRing> sring;
    if constexpr (GM == 0)
        if (grp * a.tg + wave < t_end) ggpm_ring_prefetch<2>(wps, KC, grp * a.tg + wave, lane, ring);     // under the row copies
    if constexpr (SPLIT) {
        if (grp * a.tg + wave < t_end) ggpm_split_ring_prefetch<2>(wps, KC32, grp * a.tg + wave, lane, sring);
        ggpm_split_init(Iz, 2, ROWS, Hp);
        ggpm_load_rows_to_lds_split<ROWS>(a.DZP, r0, a.E1, Hp, Iz, PLANE, LDH);
        ggpm_load_rows_to_lds_split<ROWS>(a.DMP, r0, a.E1, Hp, Im, PLANE, LDH);
    } else {
        ggpm_load_rows_to_lds<ROWS, ST16>(a.DZP, r0, a.E1, Hp, LD, T1);
        ggpm_load_rows_to_lds<ROWS, ST16>(a.DMP, r0, a.E1, Hp, LD, T2);
    }
    __syncthreads();
    for (int tt = grp * a.tg + wave; tt < t_end; tt += GGPM_NWA) {
        const int c = 16 * tt + 4 * (lane >> 4);
        float4 dsd[RTT], rco[RTT], oxr[RTT];
        auto load_inputs = [&]() {
#pragma unroll
            for (int r = 0; r < RTT; ++r) {
                const int e = r0 + 16 * r + (lane & 15);
                const size_t o = (size_t)(e < a.E1 ? e : 0) * Hp + c;
                dsd[r] = ggpm_ld4(a.DSD + o); rco[r] = ggpm_ld4(a.R + o);
                oxr[r] = a.first ? ggpm_zero4() : ggpm_ld4(a.dXr + o);
            }
        };
        if constexpr (RTT == 1) load_inputs();
        f32x4 acc[2][RTT];
        ggpm_zero_acc<2, RTT>(acc);
        {
            const float* const tiles[2] = {T2, T1};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[2] = {Im, Iz};
                ggpm_wave_gemm_split<2>(imgs, PLANE, LDH, wps, KC32, tt, tn, lane, acc, sring);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<2, RTT>(tiles, LD, wps, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<2, RTT>(tiles, LD, wps, KC, tt, tn, lane, acc, ring);
        }
        if constexpr (RTT != 1) load_inputs();
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int e = r0 + 16 * r + (lane & 15);
            if (e >= a.E1) continue;
            const size_t o = (size_t)e * Hp + c;
            float4 dg = ggpm_f4(acc[0][r]);
            if constexpr (ST16) dg = ggpm_rne4(dg);
            ggpm_stx<ST16>(a.dGout, o, dg);
            ggpm_stx<ST16>(a.dSout, o, ggpm_f4(acc[1][r]) + dsd[r]);
            ggpm_st4(a.dXr + o, oxr[r] + dg * rco[r]);
        }
    }
}

template <typename K>
inline void set_lds(K kernel, size_t bytes) { ggpm_set_lds(kernel, bytes); }      // (common.h)

// Environment switches are read once: getenv walks the whole environment (~0.5 us) and the launch helpers below run
// ~120 times per training step.
inline bool env_no_fuse_b() { static const bool v = ggpm_dev_env("GGPM_NO_FUSE_B") != nullptr; return v; }
inline bool env_adebug() { static const bool v = ggpm_dev_env("GGPM_ADEBUG") != nullptr; return v; }

inline int pick_tg(int E1, int NT) {
    static const char* const tg_env = ggpm_dev_env("GGPM_TG");
    if (const char* e = tg_env) { int v = atoi(e); if (v >= 1 && v <= 64) return v; }   // tuning override
    static const char* const tg_small_env = ggpm_dev_env("GGPM_TG_SMALL");
    if (const char* e = tg_small_env) {      // tuning override for the small (motif / attachment) levels only
        int v = atoi(e);
        if (v >= 1 && v <= 64 && (E1 + 15) / 16 <= 64) return v < NT ? v : NT;
    }
    static const char* const tg_large_env = ggpm_dev_env("GGPM_TG_LARGE");
    if (const char* e = tg_large_env) {      // tuning override for the large (atom) levels only
        int v = atoi(e);
        if (v >= 1 && v <= 64 && (E1 + 15) / 16 > 64) return v < NT ? v : NT;
    }
    return ggpm_tiles_per_group(E1, NT);
}

// two row tiles per workgroup where the level is large enough to be bound by the weight stream (see GGPM_RT2_MIN_ROW_TILES)
thread_local bool g_prefer_narrow = false;       // ggpm_level_prefer_narrow
inline bool use_rt2(int E1, int Hp, bool sparse) {
    static const int mode = [] { const char* e = ggpm_dev_env("GGPM_RT2"); return e ? atoi(e) : 1; }();     // 0 off, 2 always
    if (mode == 0 || sparse) return false;
    if ((size_t)2 * 32 * (Hp + 4) * sizeof(float) > 160 * 1024) return false;
    return mode == 2 || g_prefer_narrow || ggpm_ceil_div(E1, 16) >= GGPM_RT2_MIN_ROW_TILES;
}

// Gate mode of a level call (GruFwdArgs.bf16): the caller's dtype 1 (bf16 operands) stays; an fp32 call runs its gate products
// on split operands (mode 2: fp32 accuracy on the bf16 matrix pipe, tile_mma.h) where that form is the faster one -- DENSE
// levels whose row tiles alone fill the chip (one column group: one 16-wave workgroup per 16 messages owns all gate columns,
// four waves per SIMD take turns on the pipe; the atom level: 33.9 -> 31.4 / 39.1 -> 35.8 us per launch, LSTM 46.2 -> 36.7 /
// 49.7 -> 42.0) and whose fp32 tile + two bf16 images fit the LDS -- and on fp32 MFMA (mode 0) otherwise: with column groups
// (the tree-side levels, the decode steps) ONE wave per SIMD streams three weight planes with nothing to hide their latency
// behind and the launch gets slower (attachment level gru_bwd_a 11.7 -> 16.7 us).  Sparse calls are always mode 0, so a
// sequence of them sharing one packed weight set (ggpm_weights_packed) agrees on it.  dtype 3 forces mode 2 wherever it
// fits (tests); GGPM_GATE_SPLIT=0: mode 0 everywhere (A/B runs).
inline int gate_mode(int dtype, int Hp, bool rt2, bool single_group, bool sparse) {
    if (dtype == 1) return 1;
    if (dtype == 2) return 0;
    static const bool on = [] { const char* e = ggpm_dev_env("GGPM_GATE_SPLIT"); return !e || atoi(e) != 0; }();
    if (!on || rt2) return 0;
    if (dtype != 3 && (!single_group || sparse)) return 0;
    const size_t need = (size_t)16 * (Hp + 4) * sizeof(float) + 2 * ggpm_split_image_bytes(16, Hp);
    return need <= 160 * 1024 ? 2 : 0;
}
inline bool single_group(int E1, int Hp) { return pick_tg(E1, Hp / 16) >= Hp / 16; }

void launch_fwd(GruFwdArgs a, bool stash, bool with_b, double flops1, hipStream_t s) {
    const int Hp = a.Hp, NT = Hp / 16;
    const bool rt2 = use_rt2(a.E1, Hp, a.frozen != nullptr);
    const int rows = rt2 ? 32 : 16;
    dim3 grid_a(ggpm_ceil_div(a.E1, rows), ggpm_ceil_div(NT, a.tg));
    const bool split = a.bf16 == 2;
    const size_t tile_b = (size_t)rows * (Hp + 4) * sizeof(float), img_b = ggpm_split_image_bytes(rows, Hp);
    const size_t lds_b = split ? img_b : tile_b;
    const size_t lds_fused = split ? tile_b + 3 * img_b : 3 * tile_b;
    a.fuse_b = (!rt2 && with_b && grid_a.y == 1 && lds_fused <= 160 * 1024 && !env_no_fuse_b()) ? 1 : 0;
    static unsigned long long* dbg_buf = nullptr;
    static int dbg_count = 0;
    a.dbg = nullptr;
    if (env_adebug()) {
        if (!dbg_buf) (void)hipMalloc(&dbg_buf, 64);
        a.dbg = dbg_buf;
    }
    if (a.fuse_b) with_b = false;
    const size_t lds_a = a.fuse_b ? lds_fused : split ? tile_b + 2 * img_b : 2 * tile_b;
    ggpm_timing_begin(0, s, ((a.fuse_b ? 1 : 0) + (a.h0_zero ? 0 : 2)) * flops1);     // the first depth has no gate products
    auto go = [&](auto kernel) {
        set_lds(kernel, lds_a);
        kernel<<<grid_a, GGPM_NWA * 64, lds_a, s>>>(a);
    };
    if (a.st16) {      // (training levels only: always with stashes)
        if (rt2) go(gru_fwd_a<true, 1, 2, true>); else go(gru_fwd_a<true, 1, 1, true>);
    } else if (rt2) {
        if (a.bf16 == 1) { if (stash) go(gru_fwd_a<true, 1, 2>); else go(gru_fwd_a<false, 1, 2>); }
        else { if (stash) go(gru_fwd_a<true, 0, 2>); else go(gru_fwd_a<false, 0, 2>); }
    } else {
        if (a.bf16 == 2) { if (stash) go(gru_fwd_a<true, 2, 1>); else go(gru_fwd_a<false, 2, 1>); }
        else if (a.bf16 == 1) { if (stash) go(gru_fwd_a<true, 1, 1>); else go(gru_fwd_a<false, 1, 1>); }
        else { if (stash) go(gru_fwd_a<true, 0, 1>); else go(gru_fwd_a<false, 0, 1>); }
    }
    ggpm_timing_end(0, s);
    if (a.dbg && (++dbg_count % 97) == 0) {
        unsigned long long h[7];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[adebug E1=%d grid=%dx%d fuse=%d] gather %.2f barrier %.2f gemm %.2f epilogue %.2f us; wave 15 starts "
                "+%.2f, gathers %.2f\n", a.E1, grid_a.x, grid_a.y, a.fuse_b, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01,
                (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, ((double)h[5] - (double)h[0]) * 0.01, (h[6] - h[5]) * 0.01);
    }
    if (with_b) {
        auto gob = [&](auto kernel) {
            set_lds(kernel, lds_b);
            kernel<<<grid_a, GGPM_NWA * 64, lds_b, s>>>(a);
        };
        ggpm_timing_begin(4, s, 1 * flops1);
        if (a.st16) { if (rt2) gob(gru_fwd_b<1, 2, true>); else gob(gru_fwd_b<1, 1, true>); }
        else if (rt2) { if (a.bf16 == 1) gob(gru_fwd_b<1, 2>); else gob(gru_fwd_b<0, 2>); }
        else { if (a.bf16 == 2) gob(gru_fwd_b<2, 1>); else if (a.bf16 == 1) gob(gru_fwd_b<1, 1>); else gob(gru_fwd_b<0, 1>); }
        ggpm_timing_end(4, s);
    }
}

void launch_bwd(GruBwdArgs a, bool with_b, double flops1, hipStream_t s) {
    const int Hp = a.Hp, NT = Hp / 16;
    const bool rt2 = use_rt2(a.E1, Hp, a.frozen != nullptr);
    const int rows = rt2 ? 32 : 16;
    dim3 grid_a(ggpm_ceil_div(a.E1, rows), ggpm_ceil_div(NT, a.tg));
    const bool split = a.bf16 == 2;
    const size_t tile_b = (size_t)rows * (Hp + 4) * sizeof(float), img_b = ggpm_split_image_bytes(rows, Hp);
    const size_t lds = split ? 2 * img_b : 2 * tile_b;                        // kernel B: the dz_pre and dm_pre rows
    const size_t lds_fused = split ? tile_b + 3 * img_b : 4 * tile_b;
    a.fuse_b = (!rt2 && with_b && !a.final_pass && grid_a.y == 1 && NT <= 2 * GGPM_NWA && lds_fused <= 160 * 1024 &&
                !env_no_fuse_b()) ? 1 : 0;
    if (a.fuse_b) with_b = false;
    const size_t lds_a = a.fuse_b ? lds_fused : split ? tile_b + img_b : 2 * tile_b;
    static unsigned long long* dbg_buf = nullptr;
    static int dbg_count = 0;
    a.dbg = nullptr;
    if (env_adebug()) {
        if (!dbg_buf) (void)hipMalloc(&dbg_buf, 64);
        a.dbg = dbg_buf;
    }
    auto go = [&](auto kernel, size_t bytes) {
        set_lds(kernel, bytes);
        kernel<<<grid_a, GGPM_NWA * 64, bytes, s>>>(a);
    };
    ggpm_timing_begin(1, s, (a.fuse_b ? 3 : 1) * flops1);
    if (a.st16) { if (rt2) go(gru_bwd_a<1, 2, true>, lds_a); else go(gru_bwd_a<1, 1, true>, lds_a); }
    else if (rt2) { if (a.bf16 == 1) go(gru_bwd_a<1, 2>, lds_a); else go(gru_bwd_a<0, 2>, lds_a); }
    else { if (a.bf16 == 2) go(gru_bwd_a<2, 1>, lds_a); else if (a.bf16 == 1) go(gru_bwd_a<1, 1>, lds_a); else go(gru_bwd_a<0, 1>, lds_a); }
    ggpm_timing_end(1, s);
    if (a.dbg && !a.first && (++dbg_count % 89) == 0) {
        unsigned long long h[5];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[adebug bwd E1=%d grid=%dx%d fuse=%d] gather %.2f barrier %.2f gemm %.2f epilogue %.2f us\n", a.E1,
                grid_a.x, grid_a.y, a.fuse_b, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01,
                (h[4] - h[3]) * 0.01);
    }
    if (with_b) {
        ggpm_timing_begin(5, s, 2 * flops1);
        if (a.st16) { if (rt2) go(gru_bwd_b<1, 2, true>, lds); else go(gru_bwd_b<1, 1, true>, lds); }
        else if (rt2) { if (a.bf16 == 1) go(gru_bwd_b<1, 2>, lds); else go(gru_bwd_b<0, 2>, lds); }
        else { if (a.bf16 == 2) go(gru_bwd_b<2, 1>, lds); else if (a.bf16 == 1) go(gru_bwd_b<1, 1>, lds); else go(gru_bwd_b<0, 1>, lds); }
        ggpm_timing_end(5, s);
    }
}

}  // namespace

extern "C" size_t ggpm_gru_pack_floats(int H) {
    const int Hp = ggpm_padded_hidden(H);
    return 3 * ggpm_packed_matrix_slot(Hp) + (size_t)Hp;
}

static int gru_shape_ok(int Hp) {
    // one LDS tile pair of 16 rows must fit a CU
    return (size_t)2 * 16 * (Hp + 4) * sizeof(float) <= 160 * 1024;
}

namespace {
__global__ void sparse_init_state(const float* __restrict__ h_in, const unsigned char* __restrict__ frozen,
                                  float* __restrict__ H0, int Hp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= Hp) return;
    H0[(size_t)r * Hp + c] = frozen[r] ? h_in[(size_t)r * Hp + c] : 0.f;    // rows being recomputed start from 0
}
}  // namespace

namespace { thread_local int g_run_depth = 0, g_bwd_lo = 0, g_wgrad_lo = 0; }
void ggpm_forward_run_depth(int run_depth) { g_run_depth = run_depth; }
int ggpm_take_run_depth() { const int v = g_run_depth; g_run_depth = 0; return v; }
void ggpm_backward_lo_depth(int lo) { g_bwd_lo = lo; }
int ggpm_take_backward_lo() { const int v = g_bwd_lo; g_bwd_lo = 0; return v; }
void ggpm_wgrad_lo_depth(int lo) { g_wgrad_lo = lo; }
int ggpm_take_wgrad_lo() { const int v = g_wgrad_lo; g_wgrad_lo = 0; return v; }
namespace { thread_local bool g_wgrad_skip_bu = false; }
void ggpm_wgrad_skip_bias_u(int yes) { g_wgrad_skip_bu = yes != 0; }
namespace { thread_local float* g_defer[4] = {nullptr, nullptr, nullptr, nullptr}; thread_local bool g_defer_set = false; }
extern "C" void ggpm_backward_defer_stash(float* s0, float* s1, float* s2, float* s3) {
    g_defer[0] = s0; g_defer[1] = s1; g_defer[2] = s2; g_defer[3] = s3;
    g_defer_set = s0 != nullptr;
}
namespace { thread_local bool g_sparse_skip_wgrads = false; }
void ggpm_sparse_backward_skip_wgrads(int yes) { g_sparse_skip_wgrads = yes != 0; }
bool ggpm_take_sparse_skip_wgrads() { const bool v = g_sparse_skip_wgrads; g_sparse_skip_wgrads = false; return v; }
namespace { thread_local bool g_skip_xsum = false; }
extern "C" void ggpm_backward_skip_x_sums(int yes) { g_skip_xsum = yes != 0; }
bool ggpm_take_skip_x_sums() { const bool v = g_skip_xsum; g_skip_xsum = false; return v; }
namespace {
thread_local const float* g_gs_h = nullptr; thread_local const float* g_gs_c = nullptr; thread_local const int32_t* g_gs_idx = nullptr;
thread_local float* g_ss_h = nullptr; thread_local float* g_ss_c = nullptr; thread_local const int32_t* g_ss_idx = nullptr;
}
extern "C" void ggpm_forward_gather_state(const float* src_h, const float* src_c, const int32_t* idx) {
    g_gs_h = src_h; g_gs_c = src_c; g_gs_idx = idx;
}
bool ggpm_take_gather_state(const float** src_h, const float** src_c, const int32_t** idx) {
    *src_h = g_gs_h; *src_c = g_gs_c; *idx = g_gs_idx;
    const bool v = g_gs_idx != nullptr && g_gs_h != nullptr;
    g_gs_h = g_gs_c = nullptr; g_gs_idx = nullptr;
    return v;
}
extern "C" void ggpm_backward_scatter_state(float* dst_h, float* dst_c, const int32_t* idx) {
    g_ss_h = dst_h; g_ss_c = dst_c; g_ss_idx = idx;
}
bool ggpm_take_scatter_state(float** dst_h, float** dst_c, const int32_t** idx) {
    *dst_h = g_ss_h; *dst_c = g_ss_c; *idx = g_ss_idx;
    const bool v = g_ss_idx != nullptr && g_ss_h != nullptr;
    g_ss_h = g_ss_c = nullptr; g_ss_idx = nullptr;
    return v;
}
extern "C" void ggpm_level_prefer_narrow(int yes) { g_prefer_narrow = yes != 0; }
bool ggpm_prefer_narrow() { return g_prefer_narrow; }
namespace { thread_local bool g_packed = false; }
extern "C" void ggpm_weights_packed(int yes) { g_packed = yes != 0; }
bool ggpm_take_weights_packed() { const bool v = g_packed; g_packed = false; return v; }
bool ggpm_take_defer_stash(float* (&out)[4]) {
    const bool v = g_defer_set;
    for (int i = 0; i < 4; ++i) { out[i] = g_defer[i]; g_defer[i] = nullptr; }
    g_defer_set = false;
    return v;
}

static int gru_forward_impl(int E1, int H, int depth, const float* Xz, const float* Xr, const float* Xh,
                            const float* Wz_h, int ld_wz, const float* Ur, int ld_ur, const float* bu,
                            const float* Wh_h, int ld_wh, const int32_t* pred_rowptr, const int32_t* pred_col,
                            float* Hs, float* Qs, float* Ss, float* Gs, float* Zs, float* Ms, float* Rs,
                            float* wpack, int save_for_backward, const float* h_in, const unsigned char* frozen,
                            ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    const bool weights_packed = ggpm_take_weights_packed();      // (consumed on every path)
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xz || !Xr || !Xh || !Wz_h || !Ur || !bu || !Wh_h || !pred_rowptr ||
        !pred_col || !Hs || !Qs || !wpack)
        return GGPM_ERR_ARG;
    if (save_for_backward && (!Ss || !Gs || !Zs || !Ms || !Rs)) return GGPM_ERR_ARG;
    const int Hp = ggpm_padded_hidden(H);
    if (!gru_shape_ok(Hp)) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    const int bf16 = gate_mode(ggpm_gate_dtype(), Hp, use_rt2(E1, Hp, frozen != nullptr), single_group(E1, Hp), frozen != nullptr);      // gate mode 0 / 1 / 2
    const size_t mstep = ggpm_packed_matrix_floats(Hp, bf16);
    float* pWz = wpack; float* pWh = wpack + mstep; float* pUr = wpack + 2 * mstep;
    float* pbu = wpack + 3 * ggpm_packed_matrix_slot(Hp);
    {
        GgpmPackArgs pk = {};
        pk.W[0] = Wz_h; pk.ldw[0] = ld_wz; pk.W[1] = Wh_h; pk.ldw[1] = ld_wh; pk.W[2] = Ur; pk.ldw[2] = ld_ur;
        pk.H = H; pk.Hp = Hp; pk.transpose = 0; pk.dst = wpack; pk.bias = bu; pk.bias_out = pbu; pk.bf16 = bf16;
        if (!weights_packed) ggpm_launch_pack(pk, 3, s);
    }
    dim3 ig(ggpm_ceil_div(Hp, 256), E1);
    const int tg0 = pick_tg(E1, Hp / 16);
    const float *gs_h = nullptr, *gs_c = nullptr;
    const int32_t* gs_idx = nullptr;
    const bool gathered = ggpm_take_gather_state(&gs_h, &gs_c, &gs_idx) && frozen;      // (consumed on every path)
    if (frozen) {      // sparse_forward: start from the caller's state, q^0 = U_r h^0 + b_u by one B launch
        // (h_in == Hs: the caller put the masked start state -- frozen rows' states, zero elsewhere -- into slot 0 itself;
        // ggpm_forward_gather_state: the q^0 launch fetches it through the index and writes slot 0 on the way)
        if (h_in != Hs && !gathered) sparse_init_state<<<ig, 256, 0, s>>>(h_in, frozen, Hs, Hp);
        GruFwdArgs a0 = {};
        a0.E1 = E1; a0.Hp = Hp; a0.tg = tg0; a0.Hnew = Hs; a0.Qnew = Qs; a0.Ur = pUr; a0.bu = pbu; a0.bf16 = bf16;
        if (gathered) { a0.src_h = gs_h; a0.src_idx = gs_idx; }
        const size_t lds_b = bf16 == 2 ? ggpm_split_image_bytes(ROWS, Hp) : (size_t)ROWS * (Hp + 4) * sizeof(float);
        dim3 grid_a(ggpm_ceil_div(E1, ROWS), ggpm_ceil_div(Hp / 16, tg0));
        if (bf16 == 2) { set_lds(gru_fwd_b<2, 1>, lds_b); gru_fwd_b<2, 1><<<grid_a, GGPM_NWA * 64, lds_b, s>>>(a0); }
        else if (bf16 == 1) { set_lds(gru_fwd_b<1, 1>, lds_b); gru_fwd_b<1, 1><<<grid_a, GGPM_NWA * 64, lds_b, s>>>(a0); }
        else { set_lds(gru_fwd_b<0, 1>, lds_b); gru_fwd_b<0, 1><<<grid_a, GGPM_NWA * 64, lds_b, s>>>(a0); }
    }
    // dense levels start from h^0 = 0: the first depth launch knows that (h0_zero), so H^0 / Q^0 are never materialised

    const int tg = pick_tg(E1, Hp / 16);
    const double flops1 = 2.0 * (double)(E1 - 1) * H * H;   // algorithmic flops of ONE gate product
    static const char* const abl = ggpm_dev_env("GGPM_ABLATE");
    // bf16 storage (tile_mma.h): bf16 gate products, dense, training, every stash contraction on the bf16 tall kernel
    const bool st16 = bf16 == 1 && !frozen && save_for_backward && ggpm_bf16_storage_applies(E1, H);
    int run_depth = ggpm_take_run_depth();
    if (run_depth <= 0 || run_depth > depth || frozen || !save_for_backward) run_depth = depth;
    for (int t = 1; t <= run_depth; ++t) {
        GruFwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.Xz = Xz; a.Xr = Xr; a.Xh = Xh;
        a.Wz = pWz; a.Wh = pWh; a.Ur = pUr; a.bu = pbu; a.rowptr = pred_rowptr; a.col = pred_col;
        a.ablate = abl ? atoi(abl) : 0;
        a.frozen = frozen;
        a.bf16 = bf16;
        a.st16 = st16 ? 1 : 0;
        a.h0_zero = (t == 1 && !frozen) ? 1 : 0;
        if (save_for_backward) {
            a.Hprev = ggpm_slot_ptr(Hs, t - 1, slot, st16); a.Hnew = ggpm_slot_ptr(Hs, t, slot, st16);
            a.Hout = (st16 && t == depth) ? Hs + (size_t)depth * slot : nullptr;      // the level's result stays fp32, at its usual place
            a.Qprev = ggpm_slot_ptr(Qs, t - 1, slot, st16);
            a.Qnew = (t < depth) ? ggpm_slot_ptr(Qs, t, slot, st16) : nullptr;   // q^depth is never consumed
            a.S = ggpm_slot_ptr(Ss, t - 1, slot, st16); a.G = ggpm_slot_ptr(Gs, t - 1, slot, st16);
            a.Z = ggpm_slot_ptr(Zs, t - 1, slot, st16); a.M = ggpm_slot_ptr(Ms, t - 1, slot, st16);
            a.R = Rs + (size_t)(t - 1) * slot;
        } else {
            a.Hprev = Hs + (size_t)((t - 1) & 1) * slot; a.Hnew = Hs + (size_t)(t & 1) * slot;
            a.Qprev = Qs + (size_t)((t - 1) & 1) * slot; a.Qnew = Qs + (size_t)(t & 1) * slot;
            a.S = a.G = a.Z = a.M = a.R = nullptr;
        }
        launch_fwd(a, save_for_backward != 0, t < depth, flops1, s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_gru_forward(int E1, int H, int depth, const float* Xz, const float* Xr, const float* Xh,
                                const float* Wz_h, int ld_wz, const float* Ur, int ld_ur, const float* bu,
                                const float* Wh_h, int ld_wh, const int32_t* pred_rowptr,
                                const int32_t* pred_col, float* Hs, float* Qs, float* Ss, float* Gs, float* Zs,
                                float* Ms, float* Rs, float* wpack, int save_for_backward,
                                ggpm_stream_t stream) {
    return gru_forward_impl(E1, H, depth, Xz, Xr, Xh, Wz_h, ld_wz, Ur, ld_ur, bu, Wh_h, ld_wh, pred_rowptr, pred_col,
                            Hs, Qs, Ss, Gs, Zs, Ms, Rs, wpack, save_for_backward, nullptr, nullptr, stream);
}

extern "C" int ggpm_gru_sparse_forward(int E1, int H, int depth, const float* h_in, const unsigned char* frozen,
                                       const float* Xz, const float* Xr, const float* Xh, const float* Wz_h,
                                       int ld_wz, const float* Ur, int ld_ur, const float* bu, const float* Wh_h,
                                       int ld_wh, const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs,
                                       float* Qs, float* Ss, float* Gs, float* Zs, float* Ms, float* Rs,
                                       float* wpack, int save_for_backward, ggpm_stream_t stream) {
    if (!h_in || !frozen) return GGPM_ERR_ARG;
    return gru_forward_impl(E1, H, depth, Xz, Xr, Xh, Wz_h, ld_wz, Ur, ld_ur, bu, Wh_h, ld_wh, pred_rowptr, pred_col,
                            Hs, Qs, Ss, Gs, Zs, Ms, Rs, wpack, save_for_backward, h_in, frozen, stream);
}

extern "C" size_t ggpm_gru_backward_workspace_bytes(int E1, int H, int depth) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    const size_t slot = (size_t)E1 * Hp;
    size_t f = 0;
    f += 2 * (size_t)depth * slot;                     // DMP, DZP
    f += (size_t)depth * slot;                         // DQ (slot t = dq^t; slot 0 only used by sparse_forward)
    f += 6 * slot;                                     // dS/dG double buffers + ds_dir scratch + carry
    f += 3 * ggpm_packed_matrix_slot((int)Hp);         // packed transposes (the largest gate mode's)
    f += 256 * Hp;                                     // colsum scratch
    size_t bytes = f * sizeof(float);
    bytes += ggpm_gemm_workspace_bytes(H, H, depth * E1);   // split-K slabs (largest contraction)
    return bytes + 256;
}

static int gru_weight_grads_impl(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs,
                                 float* work, size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr, int ld_dur,
                                 float* dbu, float* dWh_h, int ld_dwh, bool with_slot0, int lo, ggpm_stream_t stream);

// a small pool of re-recordable events (the encoder drivers' stream ordering, encoder.hip)
hipEvent_t ggpm_wgrad_event(int i) {
    static thread_local hipEvent_t pool[64] = {};
    i &= 63;
    if (!pool[i] && hipEventCreateWithFlags(&pool[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    return pool[i];
}

static int gru_backward_impl(int E1, int H, int depth, const float* Xr, const float* Wz_h, int ld_wz,
                                 const float* Ur, int ld_ur, const float* Wh_h, int ld_wh,
                                 const int32_t* pred_rowptr, const int32_t* pred_col,
                                 const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                 const float* Qs, const float* Ss, const float* Gs, const float* Zs,
                                 const float* Ms, const float* Rs, const float* dHD, float* dXz, float* dXr, float* dXh,
                                 float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                 int ld_dwh, float* work, size_t work_bytes, int weight_grads,
                                 const unsigned char* frozen, float* dHin, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    const bool weights_packed = ggpm_take_weights_packed();      // (consumed on every path)
    if (ggpm_take_sparse_skip_wgrads() && frozen) weight_grads = 0;      // (ggpm_gru_sparse_weight_grads follows, on the caller's choice of stream)
    const bool skip_xsum = ggpm_take_skip_x_sums() && !frozen;
    float *ss_h = nullptr, *ss_c = nullptr;
    const int32_t* ss_idx = nullptr;
    const bool scattered = ggpm_take_scatter_state(&ss_h, &ss_c, &ss_idx) && frozen;      // (consumed on every path)
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xr || !Wz_h || !Ur || !Wh_h || !pred_rowptr || !pred_col ||
        !succ_rowptr || !succ_col || !Hs || !Qs || !Ss || !Gs || !Zs || !Ms || !Rs || !dHD || !dXz || !dXr || !dXh ||
        !dWz_h || !dUr || !dbu || !dWh_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_gru_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    if (!gru_shape_ok(Hp)) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;

    // (the packed transposes come first: their place does not depend on E1, so a sequence of calls that shares one
    // `work` buffer and one set of weights packs them once -- ggpm_weights_packed)
    float* w = work;
    const int bf16 = gate_mode(ggpm_gate_dtype(), Hp, use_rt2(E1, Hp, frozen != nullptr), single_group(E1, Hp), frozen != nullptr);      // gate mode 0 / 1 / 2
    const size_t mstep = ggpm_packed_matrix_floats(Hp, bf16);
    float* pWzT = w; float* pWhT = w + mstep; float* pUrT = w + 2 * mstep; w += 3 * ggpm_packed_matrix_slot(Hp);
    float* DMP = w; w += (size_t)depth * slot;
    float* DZP = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)depth * slot;
    float* dSb[2]; float* dGb[2];
    dSb[0] = w; w += slot; dSb[1] = w; w += slot; dGb[0] = w; w += slot; dGb[1] = w; w += slot;
    float* DSD = w; w += slot;
    float* carry = w; w += slot;
    float* csws = w; w += (size_t)256 * Hp;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);
    {       // deferred weight gradients (ggpm_backward_defer_stash): the stashes go to the caller's stacked buffers
        float* ext[4];
        if (ggpm_take_defer_stash(ext)) {
            if (!frozen || !ext[1] || !ext[2]) return GGPM_ERR_ARG;
            DMP = ext[0]; DZP = ext[1]; DQ = ext[2];
            weight_grads = 0;
        }
    }

    {
        GgpmPackArgs pk = {};
        pk.W[0] = Wz_h; pk.ldw[0] = ld_wz; pk.W[1] = Wh_h; pk.ldw[1] = ld_wh; pk.W[2] = Ur; pk.ldw[2] = ld_ur;
        pk.H = H; pk.Hp = Hp; pk.transpose = 1; pk.dst = pWzT; pk.bias = nullptr; pk.bias_out = nullptr; pk.bf16 = bf16;
        if (!weights_packed) ggpm_launch_pack(pk, 3, s);
    }
    // dXz / dXh are started (not accumulated) by the first backward depth; so is dXr when that depth has a dS/dG product
    if (depth == 1 && !frozen) (void)hipMemsetAsync(dXr, 0, slot * sizeof(float), s);

    const int tg = pick_tg(E1, Hp / 16);
    const double flops1 = 2.0 * (double)(E1 - 1) * H * H;   // algorithmic flops of ONE gate product
    const bool st16 = bf16 == 1 && !frozen && ggpm_bf16_storage_applies(E1, H);      // (as the forward decided)
    // tree-side levels: d(h^t) vanishes below step `lo` (nilpotent Jacobian, common.h); sparse runs go all the way
    int lo = ggpm_take_backward_lo();
    if (lo < 1 || lo > depth || frozen) lo = 1;
    for (int t = depth; t >= lo; --t) {
        GruBwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.first = (t == depth);
        a.Xr = Xr;
        a.st16 = st16 ? 1 : 0;
        a.Hcur = ggpm_slot_ptr(Hs, t, slot, st16);
        a.Qcur = (t < depth) ? ggpm_slot_ptr(Qs, t, slot, st16) : nullptr;
        a.S = ggpm_slot_ptr(Ss, t - 1, slot, st16); a.Z = ggpm_slot_ptr(Zs, t - 1, slot, st16);
        a.M = ggpm_slot_ptr(Ms, t - 1, slot, st16);
        a.R = Rs + (size_t)(t - 1) * slot;
        a.dHD = dHD;
        a.dSin = dSb[(t + 1) & 1]; a.dGin = dGb[(t + 1) & 1];      // (bf16 storage: bf16 in the first half of each buffer)
        a.dSout = dSb[t & 1]; a.dGout = dGb[t & 1];
        a.DQ = (t < depth) ? ggpm_slot_ptr(DQ, t, slot, st16) : nullptr;
        a.frozen = frozen; a.carry = carry; a.final_pass = 0; a.dHin = nullptr;
        a.DMP = ggpm_slot_ptr(DMP, t - 1, slot, st16); a.DZP = ggpm_slot_ptr(DZP, t - 1, slot, st16); a.DSD = DSD;
        a.dXz = dXz; a.dXr = dXr; a.dXh = dXh;
        a.WzT = pWzT; a.WhT = pWhT; a.UrT = pUrT; a.bf16 = bf16;
        a.srowptr = succ_rowptr; a.scol = succ_col;
        a.skip_xsum = skip_xsum ? 1 : 0;
        launch_bwd(a, t > 1 || frozen != nullptr, flops1, s);     // (the dS/dG launch of step lo > 1 still forms dXr)
    }
    GGPM_CHECK_LAUNCH();

    if (frozen) {      // gradient of the incoming state: one more gather + dq.U_r launch at t = 0
        GruBwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.first = 0; a.final_pass = 1;
        a.Xr = Xr; a.Hcur = Hs; a.Qcur = Qs;
        a.dSin = dSb[1]; a.dGin = dGb[1];          // written by the B launch of depth 1
        a.DQ = DQ; a.UrT = pUrT; a.srowptr = succ_rowptr; a.scol = succ_col; a.bf16 = bf16;
        a.frozen = frozen; a.carry = carry; a.dHin = dHin;
        if (scattered) { a.scat_h = ss_h; a.scat_idx = ss_idx; }
        launch_bwd(a, false, flops1, s);
        GGPM_CHECK_LAUNCH();
    }
    if (!weight_grads) return GGPM_OK;
    return gru_weight_grads_impl(E1, H, depth, Hs, Ss, Gs, work, work_bytes, dWz_h, ld_dwz, dUr, ld_dur, dbu, dWh_h,
                                 ld_dwh, frozen != nullptr, lo, stream);
}

extern "C" int ggpm_gru_backward(int E1, int H, int depth, const float* Xr, const float* Wz_h, int ld_wz,
                                 const float* Ur, int ld_ur, const float* Wh_h, int ld_wh,
                                 const int32_t* pred_rowptr, const int32_t* pred_col,
                                 const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                 const float* Qs, const float* Ss, const float* Gs, const float* Zs,
                                 const float* Ms, const float* Rs, const float* dHD, float* dXz, float* dXr,
                                 float* dXh, float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu,
                                 float* dWh_h, int ld_dwh, float* work, size_t work_bytes, int weight_grads,
                                 ggpm_stream_t stream) {
    return gru_backward_impl(E1, H, depth, Xr, Wz_h, ld_wz, Ur, ld_ur, Wh_h, ld_wh, pred_rowptr, pred_col, succ_rowptr,
                             succ_col, Hs, Qs, Ss, Gs, Zs, Ms, Rs, dHD, dXz, dXr, dXh, dWz_h, ld_dwz, dUr, ld_dur, dbu,
                             dWh_h, ld_dwh, work, work_bytes, weight_grads, nullptr, nullptr, stream);
}

// sparse_forward backward: additionally returns dHin (gradient of the incoming state; zero on the recomputed rows)
extern "C" int ggpm_gru_sparse_backward(int E1, int H, int depth, const unsigned char* frozen, const float* Xr,
                                        const float* Wz_h, int ld_wz, const float* Ur, int ld_ur,
                                        const float* Wh_h, int ld_wh, const int32_t* pred_rowptr,
                                        const int32_t* pred_col, const int32_t* succ_rowptr,
                                        const int32_t* succ_col, const float* Hs, const float* Qs, const float* Ss,
                                        const float* Gs, const float* Zs, const float* Ms, const float* Rs,
                                        const float* dHD, float* dHin, float* dXz, float* dXr, float* dXh,
                                        float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                        int ld_dwh, float* work, size_t work_bytes, ggpm_stream_t stream) {
    if (!frozen || !dHin) return GGPM_ERR_ARG;
    return gru_backward_impl(E1, H, depth, Xr, Wz_h, ld_wz, Ur, ld_ur, Wh_h, ld_wh, pred_rowptr, pred_col, succ_rowptr,
                             succ_col, Hs, Qs, Ss, Gs, Zs, Ms, Rs, dHD, dXz, dXr, dXh, dWz_h, ld_dwz, dUr, ld_dur, dbu,
                             dWh_h, ld_dwh, work, work_bytes, 1, frozen, dHin, stream);
}

// Weight gradients of the GRU message function: tall contractions over every (depth, message) row of the
// stashes ggpm_gru_backward left in `work`.  Separate entry point so that the host can run them on a second
// stream while the next level's (latency-bound) depth loop occupies the main one.
static int gru_weight_grads_impl(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs,
                                 float* work, size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr, int ld_dur,
                                 float* dbu, float* dWh_h, int ld_dwh, bool with_slot0, int lo, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Hs || !Ss || !Gs || !work || !dWz_h || !dUr || !dbu || !dWh_h)
        return GGPM_ERR_ARG;
    const bool skip_bu = g_wgrad_skip_bu;                 // (consumed: ggpm_gru_bias_u_grad forms db_u elsewhere)
    g_wgrad_skip_bu = false;
    if (lo < 1 || lo > depth || with_slot0) lo = 1;       // backward steps depth .. lo ran (stash slots lo-1 .. depth-1)
    if (work_bytes < ggpm_gru_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    float* w = work + 3 * ggpm_packed_matrix_slot(Hp);             // (layout of gru_backward_impl)
    float* DMP = w; w += (size_t)depth * slot;
    float* DZP = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)depth * slot;
    w += 6 * slot;
    float* csws = w; w += (size_t)256 * Hp;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);
    const int KD = (depth - lo + 1) * E1;
    // bf16 storage (as gru_forward_impl / gru_backward_impl decided): the stashes are bf16 in the first half of their buffers
    // and the bf16 tall kernel reads them as they are
    const bool st16 = ggpm_gate_dtype() == 1 && !with_slot0 && ggpm_bf16_storage_applies(E1, H);
    const int tall_mode = st16 ? 2 : (ggpm_gate_dtype() == 1 ? 1 : 0);
    int rc;
    // the two or three contractions in ONE launch and one reduce (they share the split-K workspace)
    ggpm_gemm_problem gp[3] = {{ggpm_slot_ptr(DMP, lo - 1, slot, st16), Hp, ggpm_slot_ptr(Gs, lo - 1, slot, st16), Hp, dWh_h, ld_dwh, H,
                                nullptr, 0, GGPM_ACT_NONE, 0},
                               {ggpm_slot_ptr(DZP, lo - 1, slot, st16), Hp, ggpm_slot_ptr(Ss, lo - 1, slot, st16), Hp, dWz_h, ld_dwz, H,
                                nullptr, 0, GGPM_ACT_NONE, 0},
                               {nullptr, Hp, nullptr, Hp, dUr, ld_dur, H, nullptr, 0, GGPM_ACT_NONE, 0}};
    int Ks[3] = {KD, KD, 0};
    if (depth > lo || with_slot0) {
        // dq^t pairs with h^t; the dense level never produces dq^0 (h^0 = 0), sparse_forward does
        const int first_slot = with_slot0 ? 0 : lo;
        const int KQ = (depth - first_slot) * E1;
        const float* dq0 = ggpm_slot_ptr(DQ, first_slot, slot, st16);
        gp[2].A = dq0;
        gp[2].B = ggpm_slot_ptr(Hs, first_slot, slot, st16);
        Ks[2] = KQ;
        if (!skip_bu) {      // (the light column sum first, the contraction last)
            rc = ggpm_colsum_any(dq0, Hp, KQ, H, dbu, csws, st16, stream);
            if (rc) return rc;
        }
        rc = ggpm_gemm_tall_grouped(H, H, 3, gp, Ks, skws, skbytes, stream, tall_mode);
        if (rc) return rc;
    } else {
        rc = ggpm_gemm_tall_grouped(H, H, 2, gp, Ks, skws, skbytes, stream, tall_mode);
        if (rc) return rc;
        for (int r = 0; r < H; ++r) (void)hipMemsetAsync(dUr + (size_t)r * ld_dur, 0, H * sizeof(float), s);
        if (!skip_bu) (void)hipMemsetAsync(dbu, 0, H * sizeof(float), s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// db_u of a dense level by itself (what gru_weight_grads_impl does last unless ggpm_wgrad_skip_bias_u was set): the column
// sum of the dq stash slots lo .. depth-1 (dq^t pairs with h^t; the dense level never produces dq^0).
int ggpm_gru_bias_u_grad(int E1, int H, int depth, int lo, float* work, float* dbu, float* csws, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !work || !dbu || !csws) return GGPM_ERR_ARG;
    if (lo < 1 || lo > depth) lo = 1;
    const int Hp = ggpm_padded_hidden(H);
    const size_t slot = (size_t)E1 * Hp;
    const float* DQ = work + 3 * ggpm_packed_matrix_slot(Hp) + 2 * (size_t)depth * slot;      // (layout of gru_backward_impl)
    if (depth <= lo) {
        (void)hipMemsetAsync(dbu, 0, H * sizeof(float), (hipStream_t)stream);
        return GGPM_OK;
    }
    const bool st16 = ggpm_gate_dtype() == 1 && ggpm_bf16_storage_applies(E1, H);
    return ggpm_colsum_any(ggpm_slot_ptr(DQ, lo, slot, st16), Hp, (depth - lo) * E1, H, dbu, csws, st16, stream);
}

namespace {
template <bool B16>
__global__ void __launch_bounds__(256) sum_slots_k(const float* __restrict__ src, int slots, size_t slot4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= slot4) return;
    float4 acc = ggpm_ldx<B16>(src, 4 * i);
    for (int t = 1; t < slots; ++t) acc = acc + ggpm_ldx<B16>(src, 4 * ((size_t)t * slot4 + i));          // fixed order
    reinterpret_cast<float4*>(out)[i] = acc;
}
}  // namespace

extern "C" int ggpm_sum_slots(const float* src, int slots, size_t slot_floats, float* out, ggpm_stream_t stream) {
    return ggpm_sum_slots_any(src, slots, slot_floats, out, false, stream);
}

int ggpm_sum_slots_any(const float* src, int slots, size_t slot_floats, float* out, bool src_bf16, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!src || !out || slots <= 0 || slot_floats == 0 || (slot_floats & 3)) return GGPM_ERR_ARG;
    const size_t slot4 = slot_floats / 4;
    if (src_bf16) sum_slots_k<true><<<(unsigned)((slot4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, slots, slot4, out);
    else sum_slots_k<false><<<(unsigned)((slot4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, slots, slot4, out);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// where ggpm_gru_backward left its dm_pre / dz_pre stashes inside `work` (slot t-1 of each = backward step t)
extern "C" int ggpm_gru_backward_stashes(float* work, int E1, int H, int depth, float** DMP, float** DZP) {
    if (!work || !DMP || !DZP || E1 <= 0 || H <= 0 || depth <= 0) return GGPM_ERR_ARG;
    const size_t Hp = (size_t)ggpm_padded_hidden(H), slot = (size_t)E1 * Hp;
    *DMP = work + 3 * ggpm_packed_matrix_slot((int)Hp);
    *DZP = *DMP + (size_t)depth * slot;
    return GGPM_OK;
}

extern "C" size_t ggpm_weight_grads_stacked_workspace_bytes(int H, int rows) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    return 256 * Hp * sizeof(float) + ggpm_gemm_workspace_bytes(H, H, rows) + 256;
}

// The hidden-half weight gradients of MANY sparse backward calls at once: their stashes stacked row-wise (one block per
// call, the same block order in every buffer), see ggpm_backward_defer_stash.
extern "C" int ggpm_gru_weight_grads_stacked(int rows, int rows_q, int H, const float* DMP, const float* Gs,
                                             const float* DZP, const float* Ss, const float* DQ, const float* Hs,
                                             float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                             int ld_dwh, float* work, size_t work_bytes, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (rows <= 0 || rows_q <= 0 || H <= 0 || !DMP || !Gs || !DZP || !Ss || !DQ || !Hs || !dWz_h || !dUr || !dbu ||
        !dWh_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_weight_grads_stacked_workspace_bytes(H, rows > rows_q ? rows : rows_q)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    float* csws = work;
    float* skws = work + (size_t)256 * Hp;
    const size_t skbytes = work_bytes - (size_t)256 * Hp * sizeof(float);
    const ggpm_gemm_problem gp[3] = {{DMP, Hp, Gs, Hp, dWh_h, ld_dwh, H, nullptr, 0, GGPM_ACT_NONE, 0},
                                     {DZP, Hp, Ss, Hp, dWz_h, ld_dwz, H, nullptr, 0, GGPM_ACT_NONE, 0},
                                     {DQ, Hp, Hs, Hp, dUr, ld_dur, H, nullptr, 0, GGPM_ACT_NONE, 0}};
    const int Ks[3] = {rows, rows, rows_q};
    // (the light column sum first: whatever else shares the GPU at the end of a pass, the stream ends with the contraction)
    int rc = ggpm_colsum(DQ, Hp, rows_q, H, dbu, csws, stream);
    if (rc) return rc;
    rc = ggpm_gemm_tall_grouped(H, H, 3, gp, Ks, skws, skbytes, stream);
    if (rc) return rc;
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

int ggpm_gru_sparse_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs, float* work,
                                 size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                 int ld_dwh, ggpm_stream_t stream) {
    return gru_weight_grads_impl(E1, H, depth, Hs, Ss, Gs, work, work_bytes, dWz_h, ld_dwz, dUr, ld_dur, dbu, dWh_h, ld_dwh,
                                 true, 1, stream);
}

extern "C" int ggpm_gru_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs,
                                     float* work, size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr,
                                     int ld_dur, float* dbu, float* dWh_h, int ld_dwh, ggpm_stream_t stream) {
    return gru_weight_grads_impl(E1, H, depth, Hs, Ss, Gs, work, work_bytes, dWz_h, ld_dwz, dUr, ld_dur, dbu, dWh_h,
                                 ld_dwh, false, ggpm_take_wgrad_lo(), stream);
}
