// Fused LSTM message step for gfx950 (all 32 shipped reference configs use rnn_type = LSTM).
//
// Reference arithmetic: LSTM.forward / LSTM.LSTM (ggpm/rnn.py:96-108, 85-94).  Restated over CSR
// predecessors; the x-halves of W_i/W_o/W/W_f (+ their biases) are hoisted (Xi, Xo, Xu, Xf) and the hidden
// half of the forget gate is applied once per message (qf = Wf_h h) instead of once per padded slot.
//
// Same geometry as mpn_gru.hip (16-row message tiles x column groups, 16 waves, MFMA f32 16x16x4):
//   fwd  A: P1 gather (h_p, c_p, qf_p) -> s, fc tiles | P2 [Wi_h;Wo_h;Wu_h].s + gate math -> h', c'
//           P3 (one column group) qf' = Wf_h h'          B: otherwise the same product from rows re-read from L2
//   bwd  A: P1 successors -> dh partial, dqf, dc tiles | P2 dh += dqf.Wf_h ; gate derivatives, dXf += dFC * F
//           P3 (one column group) dS = di_pre.Wi_h + do_pre.Wo_h + du_pre.Wu_h      B: otherwise, from L2
#include "tile_mma.h"
#include <cstdlib>

namespace {

constexpr int ROWS = 16;      // rows of a row tile (the kernels take one or two per workgroup: template parameter RTT)

struct LstmFwdArgs {
    int E1, Hp, tg;
    const float *Xi, *Xo, *Xu, *Xf;
    const float *Hprev, *Cprev, *Qprev;
    float *Hnew, *Cnew, *Qnew;
    float *S, *I, *O, *U, *F;        // stash slot (nullptr when not saving); F = sum_p c_p f(1-f)
    const float *Wi, *Wo, *Wu, *Wf;  // packed
    const int32_t *rowptr, *col;
    const unsigned char* frozen;     // sparse_forward only: rows that keep their (h, c)
    int fuse_b;                      // single column group: kernel A also forms qf' = Wf_h h' (no B launch)
    int h0_zero;                     // first depth of a dense level: h^0 = c^0 = 0 -> no gather, no gate products
    int bf16;                        // gate products on bf16 operands (packed weights are bf16 fragments then)
    int st16;                        // bf16 storage of Hs / Qs / S / I / O / U (gate mode 1, large dense training levels; tile_mma.h)
    float* Hout;                     // ... then the LAST depth also writes h' in fp32 here (the level's result)
    const float *src_h, *src_c;      // kernel B of a sparse forward's qf^0 launch (ggpm_forward_gather_state): the start
    const int32_t* src_idx;          // (h, c) of row r is (src_h, src_c)[src_idx[r]] (zero when < 0), written to Hnew / Cnew
};

__device__ __forceinline__ float4 one_minus(float4 r) { return make_float4(1.f - r.x, 1.f - r.y, 1.f - r.z, 1.f - r.w); }

// ---------------------------------------------------------------------------------------------- forward
// Kernel A (16 waves): every wave gathers one message row at a time: s over the full row (GEMM operand), the
// forget sum fc (and its backward coefficient) only over this workgroup's column group; then the first `tg`
// waves run [Wi_h; Wo_h; Wu_h] . s for their output tile and the gate math.
template <bool STASH, int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS lstm_fwd_a(LstmFwdArgs a) {
    static_assert(!ST16 || GM == 1, "bf16 storage goes with bf16 gate products");      // the cell state c and F stay fp32
    constexpr int ROWS = RTT * 16;      // RTT = 2: two row tiles per workgroup (ggpm_level_prefer_narrow), no fused P3
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;      // gate mode (LstmFwdArgs.bf16): 0 fp32 MFMA, 1 bf16, 2 split operands
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    // split operands (tile_mma.h): fc keeps its fp32 tile (epilogue only) at the head; behind it the bf16 images of s and,
    // for the fused qf' phase, h' (s itself is a GEMM operand only)
    float* Ts = SPLIT ? nullptr : lds;
    float* Tf = SPLIT ? lds : lds + ROWS * LD;      // fc, full-width tile of which only the group's columns are written/read
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp), IMG = ggpm_split_image_halves(ROWS, Hp);
    __bf16* Is = reinterpret_cast<__bf16*>(lds + ROWS * LD);
    __bf16* Ih = Is + IMG;
    if constexpr (SPLIT) ggpm_split_init(Is, a.fuse_b ? 2 : 1, ROWS, Hp);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t = grp * a.tg + wave;
    const int g_lo = 16 * a.tg * grp, g_hi = min(Hp, g_lo + 16 * a.tg);   // column range of this group

    for (int lr = wave; lr < ROWS; lr += GGPM_NWA) {
        const int row = r0 + lr;
        GgpmRowList rl;
        if (a.h0_zero) { rl.lo = 0; rl.n = 0; } else rl = ggpm_row_list(a.rowptr, row, a.E1);
        const size_t rowo = (size_t)(row < a.E1 ? row : 0) * Hp;
        for (int c0 = 0; c0 < Hp; c0 += 512) {
            int c[2], cs[2], cf[2];
            bool on[2], mine[2];
            float4 s[2], fc[2], fco[2], xf[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                c[k] = c0 + 256 * k + lane * 4;
                on[k] = c[k] < Hp;
                cs[k] = on[k] ? c[k] : 0;
                mine[k] = c[k] >= g_lo && c[k] < g_hi;
                cf[k] = mine[k] ? c[k] : g_lo;           // other lanes re-read one in-group column (no branch)
                s[k] = ggpm_zero4(); fc[k] = ggpm_zero4(); fco[k] = ggpm_zero4();
                xf[k] = ggpm_ld4(a.Xf + rowo + cf[k]);
            }
            for (int base = 0; base < rl.n; base += 64) {
                const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                const int m = min(64, rl.n - base);
                for (int j = 0; j < m; j += 2) {
                    float4 h[2][2], cc[2][2], q[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            h[u][k] = ggpm_ldx<ST16>(a.Hprev, p + cs[k]);
                            cc[u][k] = ggpm_ld4(a.Cprev + p + cf[k]);
                            q[u][k] = ggpm_ldx<ST16>(a.Qprev, p + cf[k]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {          // null slots: h[0] = c[0] = 0
                            const float4 f = ggpm_fsigmoid4<BF16>(xf[k] + q[u][k]);
                            const float4 fcc = f * cc[u][k];
                            s[k] = s[k] + h[u][k];
                            fc[k] = fc[k] + fcc;
                            fco[k] = fco[k] + fcc * one_minus(f);
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!on[k]) continue;
                if constexpr (ST16) s[k] = ggpm_rne4(s[k]);
                if constexpr (SPLIT) ggpm_split_store(Is, PLANE, LDH, lr, c[k], s[k]);
                else ggpm_st4(Ts + lr * LD + c[k], s[k]);
                if (mine[k]) {
                    ggpm_st4(Tf + lr * LD + c[k], fc[k]);
                    if (STASH && row < a.E1) {
                        const size_t o = (size_t)row * Hp + c[k];
                        ggpm_stx<ST16>(a.S, o, s[k]);
                        ggpm_st4(a.F + o, fco[k]);
                    }
                }
            }
        }
    }

    // the first weight fragments of P2 travel from L2 while this wave waits for the slower gatherers
    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps2[3] = {a.Wi, a.Wo, a.Wu};
    std::conditional_t<GM == 0, GgpmRing<3>, GgpmNoRing> ring2;
    std::conditional_t<SPLIT, GgpmSplitRing<3>, GgpmNoRing> sring2;
    if constexpr (GM == 0)
        if (!a.h0_zero && t < t_end) ggpm_ring_prefetch<3>(wps2, KC, t, lane, ring2);
    if constexpr (SPLIT)
        if (!a.h0_zero && t < t_end) ggpm_split_ring_prefetch<3>(wps2, KC32, t, lane, sring2);
    ggpm_lds_barrier();      // LDS tiles only: the stash stores above finish under the GEMM

    const int lr = lane & 15;
    float* Th = lds + 2 * ROWS * LD;      // fused P3 only: the complete h' rows of this workgroup
    for (int tt = t; tt < t_end; tt += GGPM_NWA) {
        const int c = 16 * tt + 4 * (lane >> 4);
        float4 xi[RTT], xo[RTT], xu[RTT];
        auto load_inputs = [&](int r) {
            const int row = r0 + 16 * r + lr;
            const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
            xi[r] = ggpm_ld4(a.Xi + o); xo[r] = ggpm_ld4(a.Xo + o); xu[r] = ggpm_ld4(a.Xu + o);
        };
        if constexpr (RTT == 1 && !SPLIT) load_inputs(0);     // in flight under the GEMM (two row tiles / split operands: the
                                                              // registers go to the GEMM)
        f32x4 acc[3][RTT];
        ggpm_zero_acc<3, RTT>(acc);
        if (!a.h0_zero) {
            const float* const tiles[3] = {Ts, Ts, Ts};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[3] = {Is, Is, Is};
                ggpm_wave_gemm_split<3, true>(imgs, PLANE, LDH, wps2, KC32, tt, tn, lane, acc, sring2);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<3, RTT>(tiles, LD, wps2, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<3, RTT>(tiles, LD, wps2, KC, tt, tn, lane, acc, ring2);
        }
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            if constexpr (RTT != 1 || SPLIT) load_inputs(r);
            const int lrow = 16 * r + lr, row = r0 + lrow;
            const size_t o = (size_t)(row < a.E1 ? row : 0) * Hp + c;
            float4 h = ggpm_zero4(), cn = ggpm_zero4(), gi = ggpm_zero4(), go = ggpm_zero4(), gu = ggpm_zero4();
            auto keep_h = [&](float4 v) {          // the complete h' rows for the fused qf' phase
                if constexpr (SPLIT) ggpm_split_store(Ih, PLANE, LDH, lrow, c, v);
                else ggpm_st4(Th + lrow * LD + c, v);
            };
            if (row >= a.E1) {
                if (a.fuse_b) keep_h(h);
                continue;
            }
            if (a.frozen && a.frozen[row]) {
                h = ggpm_ld4(a.Hprev + o);             // gates stashed as 0 => the backward passes dh, dc through
                cn = ggpm_ld4(a.Cprev + o);
            } else if (row != 0 || a.frozen) {
                const float4 pi = ggpm_f4(acc[0][r]) + xi[r];
                const float4 po = ggpm_f4(acc[1][r]) + xo[r];
                const float4 pu = ggpm_f4(acc[2][r]) + xu[r];
                const float4 fc = ggpm_ld4(Tf + lrow * LD + c);
                gi = ggpm_sigmoid4(pi);
                go = ggpm_sigmoid4(po);
                gu = make_float4(tanhf(pu.x), tanhf(pu.y), tanhf(pu.z), tanhf(pu.w));
                if constexpr (ST16) { gi = ggpm_rne4(gi); go = ggpm_rne4(go); gu = ggpm_rne4(gu); }      // (as stashed)
                cn = gi * gu + fc;
                h = go * make_float4(tanhf(cn.x), tanhf(cn.y), tanhf(cn.z), tanhf(cn.w));
                if constexpr (ST16) h = ggpm_rne4(h);
            }
            ggpm_stx<ST16>(a.Hnew, o, h);
            if constexpr (ST16) if (a.Hout) ggpm_st4(a.Hout + o, h);
            ggpm_st4(a.Cnew + o, cn);
            if (a.fuse_b) keep_h(h);
            if (STASH) {
                ggpm_stx<ST16>(a.I, o, gi);
                ggpm_stx<ST16>(a.O, o, go);
                ggpm_stx<ST16>(a.U, o, gu);
            }
        }
    }
    if (!a.fuse_b) return;

    // ---- P3 (single column group, RTT = 1 only): qf' = Wf_h h' from the rows this workgroup already holds
    if constexpr (RTT == 1) {
        const int row = r0 + lr;
        const float* const wps3[1] = {a.Wf};
        std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring3;
        std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring3;
        if constexpr (GM == 0)
            if (wave < NT) ggpm_ring_prefetch<1>(wps3, KC, wave, lane, ring3);
        if constexpr (SPLIT)
            if (wave < NT) ggpm_split_ring_prefetch<1>(wps3, KC32, wave, lane, sring3);
        ggpm_lds_barrier();
        for (int tt = wave; tt < NT; tt += GGPM_NWA) {
            f32x4 acc[1][1];
            ggpm_zero_acc<1, 1>(acc);
            const float* const tiles[1] = {Th};
            const int tn = tt + GGPM_NWA < NT ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[1] = {Ih};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps3, KC32, tt, tn, lane, acc, sring3);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, 1>(tiles, LD, wps3, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, 1>(tiles, LD, wps3, KC, tt, tn, lane, acc, ring3);
            const int c = 16 * tt + 4 * (lane >> 4);
            if (row < a.E1) ggpm_stx<ST16>(a.Qnew, (size_t)row * Hp + c, ggpm_f4(acc[0][0]));
        }
    }
}

// Kernel B (same geometry as A): qf' = Wf_h h'.
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS lstm_fwd_b(LstmFwdArgs a) {
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
    const float* const wps[1] = {a.Wf};
    std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring;
    std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring;
    if constexpr (GM == 0)
        if (grp * a.tg + wave < t_end) ggpm_ring_prefetch<1>(wps, KC, grp * a.tg + wave, lane, ring);     // under the row copy
    if constexpr (SPLIT) {
        if (grp * a.tg + wave < t_end) ggpm_split_ring_prefetch<1>(wps, KC32, grp * a.tg + wave, lane, sring);
        ggpm_split_init(Ih, 1, ROWS, Hp);
    }
    if (a.src_idx) {
        if constexpr (SPLIT) ggpm_gather_rows_to_lds_split<ROWS>(a.src_h, a.src_idx, r0, a.E1, Hp, Ih, PLANE, LDH, grp == 0 ? a.Hnew : nullptr);
        else ggpm_gather_rows_to_lds<ROWS>(a.src_h, a.src_idx, r0, a.E1, Hp, LD, Th, grp == 0 ? a.Hnew : nullptr);
        if (grp == 0) {            // the cell state of the same rows: copied through, not a GEMM operand
            const int q = Hp >> 2;
            for (int it = threadIdx.x; it < ROWS * q; it += blockDim.x) {
                const int lr = it / q, c = (it - lr * q) * 4, r = r0 + lr;
                if (r >= a.E1) continue;
                const int id = a.src_idx[r];
                ggpm_st4(a.Cnew + (size_t)r * Hp + c, id >= 0 ? ggpm_ld4(a.src_c + (size_t)id * Hp + c) : ggpm_zero4());
            }
        }
    } else {
        if constexpr (SPLIT) ggpm_load_rows_to_lds_split<ROWS>(a.Hnew, r0, a.E1, Hp, Ih, PLANE, LDH);
        else ggpm_load_rows_to_lds<ROWS, ST16>(a.Hnew, r0, a.E1, Hp, LD, Th);
    }
    __syncthreads();
    for (int tt = grp * a.tg + wave; tt < t_end; tt += GGPM_NWA) {
        f32x4 acc[1][RTT];
        ggpm_zero_acc<1, RTT>(acc);
        {
            const float* const tiles[1] = {Th};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[1] = {Ih};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps, KC32, tt, tn, lane, acc, sring);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, RTT>(tiles, LD, wps, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, RTT>(tiles, LD, wps, KC, tt, tn, lane, acc, ring);
        }
        const int c = 16 * tt + 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int row = r0 + 16 * r + (lane & 15);
            if (row < a.E1) ggpm_stx<ST16>(a.Qnew, (size_t)row * Hp + c, ggpm_f4(acc[0][r]));
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
struct LstmBwdArgs {
    int E1, Hp, tg;
    int first;
    const float* Xf;
    const float *Ccur, *Qcur;        // Cs[t], Qs[t]
    const float *I, *O, *U, *F;      // stash slot t-1
    const float* dHD;
    const float *dSin, *dFCin;
    float *dSout, *dFCout;
    float* DQ;                       // dqf^t stash slot (nullptr when first)
    float *DI, *DO, *DU;             // stash slot t-1
    float *dXi, *dXo, *dXu, *dXf;    // running sums (zeroed by the driver)
    const float *WiT, *WoT, *WuT, *WfT;
    const int32_t *srowptr, *scol;
    // sparse_forward only (see mpn_gru.hip): frozen rows carry dh / dc through the loop; final gather-only launch
    const unsigned char* frozen;
    const float* dCD;                // gradient of c_D (sparse_forward returns the cell state too)
    float *carry_h, *carry_c;        // [E1,Hp] each, started by the first backward depth
    int final_pass;
    float *dHin, *dCin;
    float *scat_h, *scat_c;          // ggpm_backward_scatter_state: the final pass ADDS row r's results to
    const int32_t* scat_idx;         // (scat_h, scat_c)[scat_idx[r]] (unique ids; < 0: dropped) instead of writing dHin / dCin
    int fuse_b;                      // single column group: kernel A also forms dS for depth t-1 (no B launch)
    int bf16;                        // gate products on bf16 operands
    int skip_xsum;                   // dXi / dXo / dXu are NOT accumulated here (the caller sums the DI / DO / DU stash slots)
    int st16;                        // bf16 storage of Hs / Qs / I / O / U / dS / DQ / DI / DO / DU (see LstmFwdArgs)
};

// Kernel A (16 waves): successors -> dqf (full rows), dh partial / dc (own columns) -> dh += dqf.Wf_h ->
// gate derivatives; dXf += dFC * F.
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS lstm_bwd_a(LstmBwdArgs a) {
    constexpr int ROWS = RTT * 16;      // RTT = 2: two row tiles per workgroup (ggpm_level_prefer_narrow), no fused P3
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    // split operands: the fp32 tiles of dh partial and dc (epilogue only) at the head, behind them the bf16 images of dqf
    // and, for the fused dS phase, di_pre / do_pre / du_pre
    float* T1 = lds;                      // dqf  (full rows)
    float* T0 = SPLIT ? lds : lds + ROWS * LD;                    // dh partial (group columns)
    float* T2 = SPLIT ? lds + ROWS * LD : lds + 2 * ROWS * LD;    // dc         (group columns)
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp), IMG = ggpm_split_image_halves(ROWS, Hp);
    __bf16* I1 = reinterpret_cast<__bf16*>(lds + 2 * ROWS * LD);
    __bf16* Ia = I1 + IMG;
    __bf16* Ib = Ia + IMG;
    __bf16* Ic = Ib + IMG;
    if constexpr (SPLIT) ggpm_split_init(I1, a.fuse_b ? 4 : 1, ROWS, Hp);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t = grp * a.tg + wave;
    const int g_lo = 16 * a.tg * grp, g_hi = min(Hp, g_lo + 16 * a.tg);

    if (!a.first) {
        for (int lr = wave; lr < ROWS; lr += GGPM_NWA) {
            const int p = r0 + lr;
            const GgpmRowList rl = ggpm_row_list(a.srowptr, p, a.E1);
            const size_t po = (size_t)(p < a.E1 ? p : 0) * Hp;
            for (int c0 = 0; c0 < Hp; c0 += 512) {
                int c[2], cs[2], cf[2];
                bool on[2], mine[2];
                float4 dh[2], dq[2], dc[2], cp[2], qp[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    c[k] = c0 + 256 * k + lane * 4;
                    on[k] = c[k] < Hp;
                    cs[k] = on[k] ? c[k] : 0;
                    mine[k] = c[k] >= g_lo && c[k] < g_hi;
                    cf[k] = mine[k] ? c[k] : g_lo;
                    dh[k] = ggpm_zero4(); dq[k] = ggpm_zero4(); dc[k] = ggpm_zero4();
                    cp[k] = ggpm_ld4(a.Ccur + po + cs[k]);
                    qp[k] = ggpm_ldx<ST16>(a.Qcur, po + cs[k]);
                }
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.scol, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 2) {
                        float4 xf[2][2], dfc[2][2], ds[2][2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const size_t e = (size_t)ggpm_list_at(chunk, j + u, m) * Hp;
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                xf[u][k] = ggpm_ld4(a.Xf + e + cs[k]);
                                dfc[u][k] = ggpm_ld4(a.dFCin + e + cs[k]);
                                ds[u][k] = ggpm_ldx<ST16>(a.dSin, e + cf[k]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float4 f = ggpm_fsigmoid4<BF16>(xf[u][k] + qp[k]);
                                const float4 dff = dfc[u][k] * f;
                                dh[k] = dh[k] + ds[u][k];
                                dc[k] = dc[k] + dff;
                                dq[k] = dq[k] + dff * cp[k] * one_minus(f);
                            }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (!on[k]) continue;
                    if constexpr (SPLIT) ggpm_split_store(I1, PLANE, LDH, lr, c[k], dq[k]);
                    else ggpm_st4(T1 + lr * LD + c[k], dq[k]);
                    if (mine[k]) {
                        ggpm_st4(T0 + lr * LD + c[k], dh[k]);
                        ggpm_st4(T2 + lr * LD + c[k], dc[k]);
                        if (p < a.E1) ggpm_stx<ST16>(a.DQ, (size_t)p * Hp + c[k], dq[k]);
                    }
                }
            }
        }
    }

    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps2[1] = {a.WfT};
    std::conditional_t<GM == 0, GgpmRing<1>, GgpmNoRing> ring2;
    std::conditional_t<SPLIT, GgpmSplitRing<1>, GgpmNoRing> sring2;
    if constexpr (GM == 0)
        if (!a.first && t < t_end) ggpm_ring_prefetch<1>(wps2, KC, t, lane, ring2);      // under the wait for the gatherers
    if constexpr (SPLIT)
        if (!a.first && t < t_end) ggpm_split_ring_prefetch<1>(wps2, KC32, t, lane, sring2);
    if (!a.first) ggpm_lds_barrier();      // LDS tiles only: the dqf stash stores finish under the GEMM

    const int lr = lane & 15;
    float* Ta = lds + 3 * ROWS * LD;      // fused P3 only: complete di_pre / do_pre / du_pre rows
    float* Tb = lds + 4 * ROWS * LD;
    float* Tc = lds + 5 * ROWS * LD;
    for (int tt = t; tt < t_end; tt += GGPM_NWA) {
        // split operands: everything the epilogue derives from the lane id (row, column, LDS and global addresses) is formed
        // BEHIND the GEMM from an opaque copy of it -- hoisted in front of the tile loop those ~20 values lived across the
        // GEMM beside its ring and spilled
        int lane_e = lane;
        int c = 16 * tt + 4 * (lane >> 4), lr = lane & 15;
        float4 gi[RTT], go[RTT], gu[RTT], cc[RTT], fco[RTT], oxi[RTT], oxo[RTT], oxu[RTT], oxf[RTT], dhd[RTT], dcd[RTT];
        auto load_inputs = [&](int r) {
            const int row = r0 + 16 * r + lr;
            const unsigned o = ((unsigned)(row < a.E1 ? row : 0) * (unsigned)Hp + (unsigned)c) * 4u;      // BYTES (E1 * Hp < 2^29, checked by the host)
            gi[r] = go[r] = gu[r] = cc[r] = fco[r] = oxi[r] = oxo[r] = oxu[r] = oxf[r] = ggpm_zero4();
            if (!a.final_pass) {
                gi[r] = ggpm_ldxo<ST16>(a.I, o); go[r] = ggpm_ldxo<ST16>(a.O, o); gu[r] = ggpm_ldxo<ST16>(a.U, o); cc[r] = ggpm_ld4o(a.Ccur, o);
                fco[r] = ggpm_ld4o(a.F, o);
                if (!a.first) {        // depth D starts the dX sums
                    if (!a.skip_xsum) { oxi[r] = ggpm_ld4o(a.dXi, o); oxo[r] = ggpm_ld4o(a.dXo, o); oxu[r] = ggpm_ld4o(a.dXu, o); }
                    oxf[r] = ggpm_ld4o(a.dXf, o);
                }
            }
            dhd[r] = a.first ? ggpm_ld4o(a.dHD, o) : ggpm_zero4();
            dcd[r] = (a.first && a.dCD) ? ggpm_ld4o(a.dCD, o) : ggpm_zero4();
        };
        if constexpr (RTT == 1 && !SPLIT) load_inputs(0);     // in flight under the GEMM (two row tiles / split operands: the
                                                              // registers go to the GEMM, each tile's operands are fetched
                                                              // where its epilogue starts)
        f32x4 acc[1][RTT];
        ggpm_zero_acc<1, RTT>(acc);
        if (!a.first) {
            const float* const tiles[1] = {T1};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                // (the ring is NOT carried into the epilogue -- its eleven operand loads need the registers: a wave's second
                // tile fetches its first fragments here)
                if (tt != t) ggpm_split_ring_prefetch<1>(wps2, KC32, tt, lane, sring2);
                const __bf16* const imgs[1] = {I1};
                ggpm_wave_gemm_split<1>(imgs, PLANE, LDH, wps2, KC32, tt, -1, lane, acc, sring2);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<1, RTT>(tiles, LD, wps2, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<1, RTT>(tiles, LD, wps2, KC, tt, tn, lane, acc, ring2);
        }
        if constexpr (SPLIT) {
            asm volatile("" : "+v"(lane_e));
            c = 16 * tt + 4 * (lane_e >> 4);
            lr = lane_e & 15;
        }
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            if constexpr (RTT != 1 || SPLIT) load_inputs(r);
            const int lrow = 16 * r + lr, row = r0 + lrow;
            const unsigned o = ((unsigned)(row < a.E1 ? row : 0) * (unsigned)Hp + (unsigned)c) * 4u;      // BYTES (E1 * Hp < 2^29, checked by the host)
            auto keep_iou = [&](float4 vi, float4 vo, float4 vu) {      // the complete gate-gradient rows for the fused dS phase
                if constexpr (SPLIT) {
                    ggpm_split_store(Ia, PLANE, LDH, lrow, c, vi);
                    ggpm_split_store(Ib, PLANE, LDH, lrow, c, vo);
                    ggpm_split_store(Ic, PLANE, LDH, lrow, c, vu);
                } else {
                    ggpm_st4(Ta + lrow * LD + c, vi);
                    ggpm_st4(Tb + lrow * LD + c, vo);
                    ggpm_st4(Tc + lrow * LD + c, vu);
                }
            };
            if (row >= a.E1) {
                if (a.fuse_b) keep_iou(ggpm_zero4(), ggpm_zero4(), ggpm_zero4());
                continue;
            }
            const bool frz = a.frozen && a.frozen[row];
            if (a.final_pass) {        // gradient of the incoming (h, c): frozen rows only
                float4 dh0 = ggpm_zero4(), dc0 = ggpm_zero4();
                if (frz) {
                    dh0 = ggpm_f4(acc[0][r]) + ggpm_ld4(T0 + lrow * LD + c) + ggpm_ld4o(a.carry_h, o);
                    dc0 = ggpm_ld4(T2 + lrow * LD + c) + ggpm_ld4o(a.carry_c, o);
                }
                if (a.scat_idx) {
                    const int id = frz ? a.scat_idx[row] : -1;
                    if (id >= 0) {
                        float* dh_to = a.scat_h + (size_t)id * Hp + c;
                        float* dc_to = a.scat_c + (size_t)id * Hp + c;
                        ggpm_st4(dh_to, ggpm_ld4(dh_to) + dh0);
                        ggpm_st4(dc_to, ggpm_ld4(dc_to) + dc0);
                    }
                } else {
                    ggpm_st4o(a.dHin, o, dh0);
                    ggpm_st4o(a.dCin, o, dc0);
                }
                continue;
            }
            float4 dip = ggpm_zero4(), dop = ggpm_zero4(), dup = ggpm_zero4(), dfc = ggpm_zero4();
            if (row != 0 || a.frozen) {
                float4 dh, dc;
                if (a.first) { dh = dhd[r]; dc = dcd[r]; }
                else { dh = ggpm_f4(acc[0][r]) + ggpm_ld4(T0 + lrow * LD + c); dc = ggpm_ld4(T2 + lrow * LD + c); }
                if (frz) {             // (h, c)_t = (h, c)_{t-1}: carry both gradients to the previous depth
                    if (!a.first) {        // (the first backward depth starts the carries: no memset)
                        dh = dh + ggpm_ld4o(a.carry_h, o);
                        dc = dc + ggpm_ld4o(a.carry_c, o);
                    }
                    ggpm_st4o(a.carry_h, o, dh);
                    ggpm_st4o(a.carry_c, o, dc);
                    dh = ggpm_zero4();
                    dc = ggpm_zero4();
                }
                const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, dcv[4] = {dc.x, dc.y, dc.z, dc.w};
                const float iv[4] = {gi[r].x, gi[r].y, gi[r].z, gi[r].w}, ov[4] = {go[r].x, go[r].y, go[r].z, go[r].w};
                const float uv[4] = {gu[r].x, gu[r].y, gu[r].z, gu[r].w}, cv[4] = {cc[r].x, cc[r].y, cc[r].z, cc[r].w};
                float r_i[4], r_o[4], r_u[4], r_c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tc = tanhf(cv[k]);
                    const float dct = dcv[k] + dhv[k] * ov[k] * (1.f - tc * tc);
                    r_c[k] = dct;
                    r_o[k] = dhv[k] * tc * ov[k] * (1.f - ov[k]);
                    r_i[k] = dct * uv[k] * iv[k] * (1.f - iv[k]);
                    r_u[k] = dct * iv[k] * (1.f - uv[k] * uv[k]);
                }
                dip = make_float4(r_i[0], r_i[1], r_i[2], r_i[3]);
                dop = make_float4(r_o[0], r_o[1], r_o[2], r_o[3]);
                dup = make_float4(r_u[0], r_u[1], r_u[2], r_u[3]);
                dfc = make_float4(r_c[0], r_c[1], r_c[2], r_c[3]);
                if constexpr (ST16) { dip = ggpm_rne4(dip); dop = ggpm_rne4(dop); dup = ggpm_rne4(dup); }      // (as stored)
            }
            ggpm_stxo<ST16>(a.DI, o, dip);
            ggpm_stxo<ST16>(a.DO, o, dop);
            ggpm_stxo<ST16>(a.DU, o, dup);
            ggpm_st4o(a.dFCout, o, dfc);
            if (!a.skip_xsum) {
                ggpm_st4o(a.dXi, o, oxi[r] + dip);
                ggpm_st4o(a.dXo, o, oxo[r] + dop);
                ggpm_st4o(a.dXu, o, oxu[r] + dup);
            }
            ggpm_st4o(a.dXf, o, oxf[r] + dfc * fco[r]);      // dXf_e += dFC_e * sum_p c_p f(1-f)
            if (a.fuse_b) keep_iou(dip, dop, dup);
        }
    }
    if (!a.fuse_b) return;

    // ---- P3 (single column group, RTT = 1 only): dS = di_pre.Wi_h + do_pre.Wo_h + du_pre.Wu_h from the rows held here
    if constexpr (RTT == 1) {
        const int row = r0 + lr;
        const float* const wps3[3] = {a.WiT, a.WoT, a.WuT};
        std::conditional_t<GM == 0, GgpmRing<3>, GgpmNoRing> ring3;
        std::conditional_t<SPLIT, GgpmSplitRing<3>, GgpmNoRing> sring3;
        if constexpr (GM == 0)
            if (wave < NT) ggpm_ring_prefetch<3>(wps3, KC, wave, lane, ring3);
        if constexpr (SPLIT)
            if (wave < NT) ggpm_split_ring_prefetch<3>(wps3, KC32, wave, lane, sring3);
        ggpm_lds_barrier();
        for (int tt = wave; tt < NT; tt += GGPM_NWA) {
            f32x4 acc[3][1];
            ggpm_zero_acc<3, 1>(acc);
            {
                const float* const tiles[3] = {Ta, Tb, Tc};
                const int tn = tt + GGPM_NWA < NT ? tt + GGPM_NWA : -1;
                if constexpr (SPLIT) {      // (the three products are summed: one accumulator pair, acc[0]; acc[1], acc[2] stay 0)
                    const __bf16* const imgs[3] = {Ia, Ib, Ic};
                    ggpm_wave_gemm_split<3, false, true>(imgs, PLANE, LDH, wps3, KC32, tt, tn, lane, acc, sring3);
                } else if constexpr (BF16) ggpm_wave_gemm_bf16<3, 1>(tiles, LD, wps3, Hp, tt, lane, acc);
                else ggpm_wave_gemm_ring<3, 1>(tiles, LD, wps3, KC, tt, tn, lane, acc, ring3);
            }
            const int c = 16 * tt + 4 * (lane >> 4);
            if (row < a.E1)
                ggpm_stx<ST16>(a.dSout, (size_t)row * Hp + c, ggpm_f4(acc[0][0]) + ggpm_f4(acc[1][0]) + ggpm_f4(acc[2][0]));
        }
    }
}

// Kernel B (same geometry as A): dS = di_pre.Wi_h + do_pre.Wo_h + du_pre.Wu_h (for depth t-1).
template <int GM, int RTT, bool ST16 = false>
__global__ void GGPM_A_BOUNDS lstm_bwd_b(LstmBwdArgs a) {
    constexpr int ROWS = RTT * 16;
    constexpr bool BF16 = GM == 1, SPLIT = GM == 2;
    static_assert(!SPLIT || RTT == 1, "split operands: one row tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* Ta = lds;
    float* Tb = lds + ROWS * LD;
    float* Tc = lds + 2 * ROWS * LD;
    const int LDH = ggpm_split_ldh(Hp), PLANE = ROWS * LDH, KC32 = ggpm_kc32_dev(Hp), IMG = ggpm_split_image_halves(ROWS, Hp);
    __bf16* Ia = reinterpret_cast<__bf16*>(lds);
    __bf16* Ib = Ia + IMG;
    __bf16* Ic = Ib + IMG;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;
    const int grp = blockIdx.y;
    const int t_end = min(NT, (grp + 1) * a.tg);
    const float* const wps[3] = {a.WiT, a.WoT, a.WuT};
    std::conditional_t<GM == 0, GgpmRing<3>, GgpmNoRing> ring;
    std::conditional_t<SPLIT, GgpmSplitRing<3>, GgpmNoRing> sring;
    if constexpr (GM == 0)
        if (grp * a.tg + wave < t_end) ggpm_ring_prefetch<3>(wps, KC, grp * a.tg + wave, lane, ring);     // under the row copies
    if constexpr (SPLIT) {
        if (grp * a.tg + wave < t_end) ggpm_split_ring_prefetch<3>(wps, KC32, grp * a.tg + wave, lane, sring);
        ggpm_split_init(Ia, 3, ROWS, Hp);
        ggpm_load_rows_to_lds_split<ROWS>(a.DI, r0, a.E1, Hp, Ia, PLANE, LDH);
        ggpm_load_rows_to_lds_split<ROWS>(a.DO, r0, a.E1, Hp, Ib, PLANE, LDH);
        ggpm_load_rows_to_lds_split<ROWS>(a.DU, r0, a.E1, Hp, Ic, PLANE, LDH);
    } else {
        ggpm_load_rows_to_lds<ROWS, ST16>(a.DI, r0, a.E1, Hp, LD, Ta);
        ggpm_load_rows_to_lds<ROWS, ST16>(a.DO, r0, a.E1, Hp, LD, Tb);
        ggpm_load_rows_to_lds<ROWS, ST16>(a.DU, r0, a.E1, Hp, LD, Tc);
    }
    __syncthreads();
    for (int tt = grp * a.tg + wave; tt < t_end; tt += GGPM_NWA) {
        f32x4 acc[3][RTT];
        ggpm_zero_acc<3, RTT>(acc);
        {
            const float* const tiles[3] = {Ta, Tb, Tc};
            const int tn = tt + GGPM_NWA < t_end ? tt + GGPM_NWA : -1;
            if constexpr (SPLIT) {
                const __bf16* const imgs[3] = {Ia, Ib, Ic};
                ggpm_wave_gemm_split<3, false, true>(imgs, PLANE, LDH, wps, KC32, tt, tn, lane, acc, sring);
            } else if constexpr (BF16) ggpm_wave_gemm_bf16<3, RTT>(tiles, LD, wps, Hp, tt, lane, acc);
            else ggpm_wave_gemm_ring<3, RTT>(tiles, LD, wps, KC, tt, tn, lane, acc, ring);
        }
        const int c = 16 * tt + 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < RTT; ++r) {
            const int e = r0 + 16 * r + (lane & 15);
            if (e < a.E1)
                ggpm_stx<ST16>(a.dSout, (size_t)e * Hp + c, ggpm_f4(acc[0][r]) + ggpm_f4(acc[1][r]) + ggpm_f4(acc[2][r]));
        }
    }
}

template <typename K>
inline void set_lds(K kernel, size_t bytes) { ggpm_set_lds(kernel, bytes); }      // (common.h)

inline size_t lds_tiles(int n, int Hp, int rows = ROWS) { return (size_t)n * rows * (Hp + 4) * sizeof(float); }
// ggpm_level_prefer_narrow (mpn_gru.hip, common.h): two row tiles per workgroup for the dense fp32 level calls of this thread
inline bool use_rt2(int Hp, bool sparse, int bf16) {
    return ggpm_prefer_narrow() && !sparse && !bf16 && lds_tiles(3, Hp, 32) <= 160 * 1024;
}

// Environment switches are read once: getenv walks the whole environment (~0.5 us) and the launch helpers below run
// ~120 times per training step.
inline bool env_no_fuse_b() { static const bool v = ggpm_dev_env("GGPM_NO_FUSE_B") != nullptr; return v; }
inline bool env_adebug() { static const bool v = ggpm_dev_env("GGPM_ADEBUG") != nullptr; return v; }

inline int pick_tg(int E1, int NT) {
    static const char* const tg_env = ggpm_dev_env("GGPM_TG");
    if (const char* e = tg_env) { int v = atoi(e); if (v >= 1 && v <= 64) return v; }   // tuning override
    static const char* const tg_large_env = ggpm_dev_env("GGPM_TG_LARGE");
    if (const char* e = tg_large_env) {      // tuning override for the large (atom) levels only
        int v = atoi(e);
        if (v >= 1 && v <= 64 && (E1 + 15) / 16 > 64) return v < NT ? v : NT;
    }
    return ggpm_tiles_per_group(E1, NT);
}

// Gate mode of a level call (mpn_gru.hip: gate_mode): fp32 calls run their gate products on split operands (mode 2) where
// one 16-row tile per workgroup applies and the backward's two fp32 tiles + the dqf image fit the LDS.
inline int gate_mode(int dtype, int Hp, bool rt2, bool single_group, bool sparse) {
    if (dtype == 1) return 1;
    if (dtype == 2) return 0;
    static const bool on = [] { const char* e = ggpm_dev_env("GGPM_GATE_SPLIT"); return !e || atoi(e) != 0; }();
    if (!on || rt2) return 0;
    if (dtype != 3 && (!single_group || sparse)) return 0;
    // kernel A of the backward: two fp32 tiles + the dqf image; kernel B of the backward: three gate-gradient images
    const size_t img = ggpm_split_image_bytes(16, Hp);
    return (lds_tiles(2, Hp) + img <= 160 * 1024 && 3 * img <= 160 * 1024) ? 2 : 0;
}

void launch_fwd(LstmFwdArgs a, bool stash, bool with_b, double flops1, hipStream_t s) {
    const int Hp = a.Hp, NT = Hp / 16;
    const bool rt2 = use_rt2(Hp, a.frozen != nullptr, a.bf16);
    const int rows = rt2 ? 32 : 16;
    dim3 grid_a(ggpm_ceil_div(a.E1, rows), ggpm_ceil_div(NT, a.tg));
    const bool split = a.bf16 == 2;
    const size_t img_b = ggpm_split_image_bytes(rows, Hp);
    const size_t l_fused = split ? lds_tiles(1, Hp) + 2 * img_b : lds_tiles(3, Hp);
    a.fuse_b = (!rt2 && with_b && grid_a.y == 1 && l_fused <= 160 * 1024 && !env_no_fuse_b()) ? 1 : 0;
    if (a.fuse_b) with_b = false;
    const size_t la = a.fuse_b ? l_fused : split ? lds_tiles(1, Hp) + img_b : lds_tiles(2, Hp, rows);
    const size_t lb = split ? img_b : lds_tiles(1, Hp, rows);
    ggpm_timing_begin(2, s, ((a.fuse_b ? 1 : 0) + (a.h0_zero ? 0 : 3)) * flops1);     // the first depth has no gate products
    auto go = [&](auto kernel) {
        set_lds(kernel, la);
        kernel<<<grid_a, GGPM_NWA * 64, la, s>>>(a);
    };
    if (a.st16) go(lstm_fwd_a<true, 1, 1, true>);      // (training levels only: always with stashes; bf16 mode has no two-row-tile form)
    else if (rt2) { if (stash) go(lstm_fwd_a<true, 0, 2>); else go(lstm_fwd_a<false, 0, 2>); }
    else if (a.bf16 == 2) { if (stash) go(lstm_fwd_a<true, 2, 1>); else go(lstm_fwd_a<false, 2, 1>); }
    else if (a.bf16 == 1) { if (stash) go(lstm_fwd_a<true, 1, 1>); else go(lstm_fwd_a<false, 1, 1>); }
    else { if (stash) go(lstm_fwd_a<true, 0, 1>); else go(lstm_fwd_a<false, 0, 1>); }
    ggpm_timing_end(2, s);
    if (with_b) {
        ggpm_timing_begin(6, s, 1 * flops1);
        if (a.st16) { set_lds(lstm_fwd_b<1, 1, true>, lb); lstm_fwd_b<1, 1, true><<<grid_a, GGPM_NWA * 64, lb, s>>>(a); }
        else if (rt2) { set_lds(lstm_fwd_b<0, 2>, lb); lstm_fwd_b<0, 2><<<grid_a, GGPM_NWA * 64, lb, s>>>(a); }
        else if (a.bf16 == 2) { set_lds(lstm_fwd_b<2, 1>, lb); lstm_fwd_b<2, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a); }
        else if (a.bf16 == 1) { set_lds(lstm_fwd_b<1, 1>, lb); lstm_fwd_b<1, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a); }
        else { set_lds(lstm_fwd_b<0, 1>, lb); lstm_fwd_b<0, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a); }
        ggpm_timing_end(6, s);
    }
}

void launch_bwd(LstmBwdArgs a, bool with_b, double flops1, hipStream_t s) {
    const int Hp = a.Hp, NT = Hp / 16;
    const bool rt2 = use_rt2(Hp, a.frozen != nullptr, a.bf16);
    const int rows = rt2 ? 32 : 16;
    dim3 grid_a(ggpm_ceil_div(a.E1, rows), ggpm_ceil_div(NT, a.tg));
    const bool split = a.bf16 == 2;
    const size_t img_b = ggpm_split_image_bytes(rows, Hp);
    const size_t l3 = split ? 3 * img_b : lds_tiles(3, Hp, rows);                  // kernel B: the three gate-gradient row sets
    const size_t l_fused = split ? lds_tiles(2, Hp) + 4 * img_b : lds_tiles(6, Hp);
    a.fuse_b = (!rt2 && with_b && !a.final_pass && grid_a.y == 1 && l_fused <= 160 * 1024 && !env_no_fuse_b()) ? 1 : 0;
    if (a.fuse_b) with_b = false;
    const size_t la = a.fuse_b ? l_fused : split ? lds_tiles(2, Hp) + img_b : lds_tiles(3, Hp, rows);
    ggpm_timing_begin(3, s, (a.fuse_b ? 4 : 1) * flops1);
    if (a.st16) { set_lds(lstm_bwd_a<1, 1, true>, la); lstm_bwd_a<1, 1, true><<<grid_a, GGPM_NWA * 64, la, s>>>(a); }
    else if (rt2) { set_lds(lstm_bwd_a<0, 2>, la); lstm_bwd_a<0, 2><<<grid_a, GGPM_NWA * 64, la, s>>>(a); }
    else if (a.bf16 == 2) { set_lds(lstm_bwd_a<2, 1>, la); lstm_bwd_a<2, 1><<<grid_a, GGPM_NWA * 64, la, s>>>(a); }
    else if (a.bf16 == 1) { set_lds(lstm_bwd_a<1, 1>, la); lstm_bwd_a<1, 1><<<grid_a, GGPM_NWA * 64, la, s>>>(a); }
    else { set_lds(lstm_bwd_a<0, 1>, la); lstm_bwd_a<0, 1><<<grid_a, GGPM_NWA * 64, la, s>>>(a); }
    ggpm_timing_end(3, s);
    if (with_b) {
        ggpm_timing_begin(7, s, 3 * flops1);
        if (a.st16) { set_lds(lstm_bwd_b<1, 1, true>, l3); lstm_bwd_b<1, 1, true><<<grid_a, GGPM_NWA * 64, l3, s>>>(a); }
        else if (rt2) { set_lds(lstm_bwd_b<0, 2>, l3); lstm_bwd_b<0, 2><<<grid_a, GGPM_NWA * 64, l3, s>>>(a); }
        else if (a.bf16 == 2) { set_lds(lstm_bwd_b<2, 1>, l3); lstm_bwd_b<2, 1><<<grid_a, GGPM_NWA * 64, l3, s>>>(a); }
        else if (a.bf16 == 1) { set_lds(lstm_bwd_b<1, 1>, l3); lstm_bwd_b<1, 1><<<grid_a, GGPM_NWA * 64, l3, s>>>(a); }
        else { set_lds(lstm_bwd_b<0, 1>, l3); lstm_bwd_b<0, 1><<<grid_a, GGPM_NWA * 64, l3, s>>>(a); }
        ggpm_timing_end(7, s);
    }
}

}  // namespace

extern "C" size_t ggpm_lstm_pack_floats(int H) {
    return 4 * ggpm_packed_matrix_slot(ggpm_padded_hidden(H));
}

static int lstm_shape_ok(int Hp) { return lds_tiles(3, Hp) <= 160 * 1024; }

namespace {
__global__ void lstm_sparse_init_state(const float* __restrict__ h_in, const float* __restrict__ c_in,
                                       const unsigned char* __restrict__ frozen, float* __restrict__ H0,
                                       float* __restrict__ C0, int Hp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= Hp) return;
    const size_t o = (size_t)r * Hp + c;
    H0[o] = frozen[r] ? h_in[o] : 0.f;
    C0[o] = frozen[r] ? c_in[o] : 0.f;
}
}  // namespace

static int lstm_forward_impl(int E1, int H, int depth, const float* Xi, const float* Xo, const float* Xu,
                             const float* Xf, const float* Wi_h, int ld_wi, const float* Wo_h, int ld_wo,
                             const float* Wu_h, int ld_wu, const float* Wf_h, int ld_wf,
                             const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs, float* Cs, float* Qs,
                             float* Ss, float* Is, float* Os, float* Us, float* Fs, float* wpack,
                             int save_for_backward, const float* h_in, const float* c_in,
                             const unsigned char* frozen, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    const bool weights_packed = ggpm_take_weights_packed();      // (consumed on every path)
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xi || !Xo || !Xu || !Xf || !Wi_h || !Wo_h || !Wu_h || !Wf_h ||
        !pred_rowptr || !pred_col || !Hs || !Cs || !Qs || !wpack)
        return GGPM_ERR_ARG;
    if (save_for_backward && (!Ss || !Is || !Os || !Us || !Fs)) return GGPM_ERR_ARG;
    const int Hp = ggpm_padded_hidden(H);
    if (!lstm_shape_ok(Hp) || (size_t)E1 * Hp >= ((size_t)1 << 30)) return GGPM_ERR_UNSUPPORTED;      // (32-bit byte offsets inside a slot)
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    const int bf16 = gate_mode(ggpm_gate_dtype(), Hp, use_rt2(Hp, frozen != nullptr, ggpm_gate_dtype() == 1),
                               pick_tg(E1, Hp / 16) >= Hp / 16, frozen != nullptr);      // gate mode 0 / 1 / 2
    const size_t mstep = ggpm_packed_matrix_floats(Hp, bf16);
    float* pWi = wpack; float* pWo = wpack + mstep; float* pWu = wpack + 2 * mstep; float* pWf = wpack + 3 * mstep;
    {
        GgpmPackArgs pk = {};
        pk.W[0] = Wi_h; pk.ldw[0] = ld_wi; pk.W[1] = Wo_h; pk.ldw[1] = ld_wo; pk.W[2] = Wu_h; pk.ldw[2] = ld_wu;
        pk.W[3] = Wf_h; pk.ldw[3] = ld_wf;
        pk.H = H; pk.Hp = Hp; pk.transpose = 0; pk.dst = wpack; pk.bias = nullptr; pk.bias_out = nullptr; pk.bf16 = bf16;
        if (!weights_packed) ggpm_launch_pack(pk, 4, s);
    }
    const int tg = pick_tg(E1, Hp / 16);
    const float *gs_h = nullptr, *gs_c = nullptr;
    const int32_t* gs_idx = nullptr;
    const bool gathered = ggpm_take_gather_state(&gs_h, &gs_c, &gs_idx) && frozen && gs_c;      // (consumed on every path)
    if (frozen) {      // sparse_forward: start from the caller's (h, c); qf^0 = Wf_h h^0 by one B launch
        dim3 ig(ggpm_ceil_div(Hp, 256), E1);
        // (h_in == Hs and c_in == Cs: the caller put the masked start state into slot 0 itself; ggpm_forward_gather_state:
        // the qf^0 launch fetches it through the index and writes slot 0 of both on the way)
        if ((h_in != Hs || c_in != Cs) && !gathered) lstm_sparse_init_state<<<ig, 256, 0, s>>>(h_in, c_in, frozen, Hs, Cs, Hp);
        LstmFwdArgs a0 = {};
        a0.E1 = E1; a0.Hp = Hp; a0.tg = tg; a0.Hnew = Hs; a0.Qnew = Qs; a0.Wf = pWf; a0.bf16 = bf16;
        if (gathered) { a0.src_h = gs_h; a0.src_c = gs_c; a0.src_idx = gs_idx; a0.Cnew = Cs; }
        const size_t lb = bf16 == 2 ? ggpm_split_image_bytes(ROWS, Hp) : lds_tiles(1, Hp);
        dim3 grid_a(ggpm_ceil_div(E1, ROWS), ggpm_ceil_div(Hp / 16, tg));
        if (bf16 == 2) { set_lds(lstm_fwd_b<2, 1>, lb); lstm_fwd_b<2, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a0); }
        else if (bf16 == 1) { set_lds(lstm_fwd_b<1, 1>, lb); lstm_fwd_b<1, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a0); }
        else { set_lds(lstm_fwd_b<0, 1>, lb); lstm_fwd_b<0, 1><<<grid_a, GGPM_NWA * 64, lb, s>>>(a0); }
    } else {
        (void)hipMemsetAsync(Hs, 0, slot * sizeof(float), s);
        (void)hipMemsetAsync(Cs, 0, slot * sizeof(float), s);
        (void)hipMemsetAsync(Qs, 0, slot * sizeof(float), s);
    }

    const double flops1 = 2.0 * (double)(E1 - 1) * H * H;   // algorithmic flops of ONE gate product
    int run_depth = ggpm_take_run_depth();
    if (run_depth <= 0 || run_depth > depth || frozen || !save_for_backward) run_depth = depth;
    // bf16 storage (tile_mma.h): bf16 gate products, dense, training, every stash contraction on the bf16 tall kernel
    const bool st16 = bf16 == 1 && !frozen && save_for_backward && ggpm_bf16_storage_applies(E1, H);
    for (int t = 1; t <= run_depth; ++t) {
        LstmFwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.Xi = Xi; a.Xo = Xo; a.Xu = Xu; a.Xf = Xf;
        a.Wi = pWi; a.Wo = pWo; a.Wu = pWu; a.Wf = pWf; a.rowptr = pred_rowptr; a.col = pred_col;
        a.frozen = frozen;
        a.bf16 = bf16;
        a.st16 = st16 ? 1 : 0;
        a.h0_zero = (t == 1 && !frozen) ? 1 : 0;
        if (save_for_backward) {
            a.Hprev = ggpm_slot_ptr(Hs, t - 1, slot, st16); a.Hnew = ggpm_slot_ptr(Hs, t, slot, st16);
            a.Hout = (st16 && t == depth) ? Hs + (size_t)depth * slot : nullptr;      // the level's result stays fp32, at its usual place
            a.Cprev = Cs + (size_t)(t - 1) * slot; a.Cnew = Cs + (size_t)t * slot;
            a.Qprev = ggpm_slot_ptr(Qs, t - 1, slot, st16); a.Qnew = (t < depth) ? ggpm_slot_ptr(Qs, t, slot, st16) : nullptr;
            a.S = ggpm_slot_ptr(Ss, t - 1, slot, st16); a.I = ggpm_slot_ptr(Is, t - 1, slot, st16);
            a.O = ggpm_slot_ptr(Os, t - 1, slot, st16); a.U = ggpm_slot_ptr(Us, t - 1, slot, st16);
            a.F = Fs + (size_t)(t - 1) * slot;
        } else {
            a.Hprev = Hs + (size_t)((t - 1) & 1) * slot; a.Hnew = Hs + (size_t)(t & 1) * slot;
            a.Cprev = Cs + (size_t)((t - 1) & 1) * slot; a.Cnew = Cs + (size_t)(t & 1) * slot;
            a.Qprev = Qs + (size_t)((t - 1) & 1) * slot; a.Qnew = Qs + (size_t)(t & 1) * slot;
            a.S = a.I = a.O = a.U = a.F = nullptr;
        }
        launch_fwd(a, save_for_backward != 0, t < depth, flops1, s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_lstm_forward(int E1, int H, int depth, const float* Xi, const float* Xo, const float* Xu,
                                 const float* Xf, const float* Wi_h, int ld_wi, const float* Wo_h, int ld_wo,
                                 const float* Wu_h, int ld_wu, const float* Wf_h, int ld_wf,
                                 const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs, float* Cs,
                                 float* Qs, float* Ss, float* Is, float* Os, float* Us, float* Fs,
                                 float* wpack, int save_for_backward, ggpm_stream_t stream) {
    return lstm_forward_impl(E1, H, depth, Xi, Xo, Xu, Xf, Wi_h, ld_wi, Wo_h, ld_wo, Wu_h, ld_wu, Wf_h, ld_wf, pred_rowptr,
                             pred_col, Hs, Cs, Qs, Ss, Is, Os, Us, Fs, wpack, save_for_backward, nullptr, nullptr,
                             nullptr, stream);
}

extern "C" int ggpm_lstm_sparse_forward(int E1, int H, int depth, const float* h_in, const float* c_in,
                                        const unsigned char* frozen, const float* Xi, const float* Xo,
                                        const float* Xu, const float* Xf, const float* Wi_h, int ld_wi,
                                        const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu,
                                        const float* Wf_h, int ld_wf, const int32_t* pred_rowptr,
                                        const int32_t* pred_col, float* Hs, float* Cs, float* Qs, float* Ss,
                                        float* Is, float* Os, float* Us, float* Fs, float* wpack,
                                        int save_for_backward, ggpm_stream_t stream) {
    if (!h_in || !c_in || !frozen) return GGPM_ERR_ARG;
    return lstm_forward_impl(E1, H, depth, Xi, Xo, Xu, Xf, Wi_h, ld_wi, Wo_h, ld_wo, Wu_h, ld_wu, Wf_h, ld_wf, pred_rowptr,
                             pred_col, Hs, Cs, Qs, Ss, Is, Os, Us, Fs, wpack, save_for_backward, h_in, c_in, frozen,
                             stream);
}

extern "C" size_t ggpm_lstm_backward_workspace_bytes(int E1, int H, int depth) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    const size_t slot = (size_t)E1 * Hp;
    size_t f = 0;
    f += 3 * (size_t)depth * slot;                     // DI, DO, DU
    f += (size_t)depth * slot;                         // DQ (slot t = dqf^t; slot 0 only used by sparse_forward)
    f += 6 * slot;                                     // dS / dFC double buffers + dh / dc carries
    f += 4 * ggpm_packed_matrix_slot((int)Hp);         // packed transposes (the largest gate mode's)
    size_t bytes = f * sizeof(float);
    bytes += ggpm_gemm_workspace_bytes(H, H, depth * E1);
    return bytes + 256;
}

static int lstm_weight_grads_impl(int E1, int H, int depth, const float* Hs, const float* Ss, float* work,
                                  size_t work_bytes, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h,
                                  int ld_dwu, float* dWf_h, int ld_dwf, bool with_slot0, int lo, ggpm_stream_t stream);

static int lstm_backward_impl(int E1, int H, int depth, const float* Xf, const float* Wi_h, int ld_wi,
                                  const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                                  int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col,
                                  const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                  const float* Cs, const float* Qs, const float* Ss, const float* Is,
                                  const float* Os, const float* Us, const float* Fs, const float* dHD, float* dXi,
                                  float* dXo,
                                  float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                                  float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                                  size_t work_bytes, int weight_grads, const unsigned char* frozen,
                                  const float* dCD, float* dHin, float* dCin, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (ggpm_take_sparse_skip_wgrads() && frozen) weight_grads = 0;      // (ggpm_lstm_sparse_weight_grads follows)
    const bool weights_packed = ggpm_take_weights_packed();      // (consumed on every path)
    float *ss_h = nullptr, *ss_c = nullptr;
    const int32_t* ss_idx = nullptr;
    const bool scattered = ggpm_take_scatter_state(&ss_h, &ss_c, &ss_idx) && frozen && ss_c;      // (consumed on every path)
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xf || !Wi_h || !Wo_h || !Wu_h || !Wf_h || !pred_rowptr || !pred_col ||
        !succ_rowptr || !succ_col || !Hs || !Cs || !Qs || !Ss || !Is || !Os || !Us || !Fs || !dHD || !dXi || !dXo || !dXu ||
        !dXf || !dWi_h || !dWo_h || !dWu_h || !dWf_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_lstm_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    if (!lstm_shape_ok(Hp) || (size_t)E1 * Hp >= ((size_t)1 << 30)) return GGPM_ERR_UNSUPPORTED;      // (32-bit byte offsets inside a slot)
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;

    const bool skip_xsum = ggpm_take_skip_x_sums() && !frozen;
    // (the packed transposes come first: their place does not depend on E1 -- ggpm_weights_packed)
    float* w = work;
    const int bf16 = gate_mode(ggpm_gate_dtype(), Hp, use_rt2(Hp, frozen != nullptr, ggpm_gate_dtype() == 1),
                               pick_tg(E1, Hp / 16) >= Hp / 16, frozen != nullptr);      // gate mode 0 / 1 / 2
    const size_t mstep = ggpm_packed_matrix_floats(Hp, bf16);
    float* pWiT = w; float* pWoT = w + mstep; float* pWuT = w + 2 * mstep; float* pWfT = w + 3 * mstep;
    w += 4 * ggpm_packed_matrix_slot(Hp);
    float* DI = w; w += (size_t)depth * slot;
    float* DO = w; w += (size_t)depth * slot;
    float* DU = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)depth * slot;
    float* dSb[2]; float* dFb[2];
    dSb[0] = w; w += slot; dSb[1] = w; w += slot; dFb[0] = w; w += slot; dFb[1] = w; w += slot;
    float* carry_h = w; w += slot; float* carry_c = w; w += slot;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);
    {       // deferred weight gradients (ggpm_backward_defer_stash): the stashes go to the caller's stacked buffers
        float* ext[4];
        if (ggpm_take_defer_stash(ext)) {
            if (!frozen || !ext[1] || !ext[2] || !ext[3]) return GGPM_ERR_ARG;
            DI = ext[0]; DO = ext[1]; DU = ext[2]; DQ = ext[3];
            weight_grads = 0;
        }
    }
    (void)skws; (void)skbytes;

    {
        GgpmPackArgs pk = {};
        pk.W[0] = Wi_h; pk.ldw[0] = ld_wi; pk.W[1] = Wo_h; pk.ldw[1] = ld_wo; pk.W[2] = Wu_h; pk.ldw[2] = ld_wu;
        pk.W[3] = Wf_h; pk.ldw[3] = ld_wf;
        pk.H = H; pk.Hp = Hp; pk.transpose = 1; pk.dst = pWiT; pk.bias = nullptr; pk.bias_out = nullptr; pk.bf16 = bf16;
        if (!weights_packed) ggpm_launch_pack(pk, 4, s);
    }
    // dXi / dXo / dXu / dXf are started (not accumulated) by the first backward depth

    const int tg = pick_tg(E1, Hp / 16);
    const double flops1 = 2.0 * (double)(E1 - 1) * H * H;   // algorithmic flops of ONE gate product
    // tree-side levels: d(h^t), d(c^t) vanish below step `lo` (nilpotent Jacobian, common.h)
    int lo = ggpm_take_backward_lo();
    if (lo < 1 || lo > depth || frozen) lo = 1;
    const bool st16 = bf16 == 1 && !frozen && ggpm_bf16_storage_applies(E1, H);      // (as the forward decided)
    for (int t = depth; t >= lo; --t) {
        LstmBwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.first = (t == depth);
        a.st16 = st16 ? 1 : 0;
        a.Xf = Xf;
        a.Ccur = Cs + (size_t)t * slot;
        a.Qcur = (t < depth) ? ggpm_slot_ptr(Qs, t, slot, st16) : nullptr;
        a.F = Fs + (size_t)(t - 1) * slot;
        a.I = ggpm_slot_ptr(Is, t - 1, slot, st16); a.O = ggpm_slot_ptr(Os, t - 1, slot, st16);
        a.U = ggpm_slot_ptr(Us, t - 1, slot, st16);
        a.dHD = dHD;
        a.dSin = dSb[(t + 1) & 1]; a.dFCin = dFb[(t + 1) & 1];
        a.dSout = dSb[t & 1]; a.dFCout = dFb[t & 1];
        a.DQ = (t < depth) ? ggpm_slot_ptr(DQ, t, slot, st16) : nullptr;
        a.frozen = frozen; a.dCD = dCD; a.carry_h = carry_h; a.carry_c = carry_c; a.final_pass = 0;
        a.dHin = nullptr; a.dCin = nullptr;
        a.DI = ggpm_slot_ptr(DI, t - 1, slot, st16); a.DO = ggpm_slot_ptr(DO, t - 1, slot, st16);
        a.DU = ggpm_slot_ptr(DU, t - 1, slot, st16);
        a.dXi = dXi; a.dXo = dXo; a.dXu = dXu; a.dXf = dXf;
        a.WiT = pWiT; a.WoT = pWoT; a.WuT = pWuT; a.WfT = pWfT; a.bf16 = bf16;
        a.srowptr = succ_rowptr; a.scol = succ_col;
        a.skip_xsum = skip_xsum ? 1 : 0;
        launch_bwd(a, t > 1 || frozen != nullptr, flops1, s);
    }
    GGPM_CHECK_LAUNCH();
    if (frozen) {      // gradient of the incoming (h, c): one more gather + dqf.Wf_h launch at t = 0
        LstmBwdArgs a = {};
        a.E1 = E1; a.Hp = Hp; a.tg = tg; a.first = 0; a.final_pass = 1;
        a.Xf = Xf; a.Ccur = Cs; a.Qcur = Qs;
        a.dSin = dSb[1]; a.dFCin = dFb[1];          // written by the launches of depth 1
        a.DQ = DQ; a.WfT = pWfT; a.srowptr = succ_rowptr; a.scol = succ_col; a.bf16 = bf16;
        a.frozen = frozen; a.carry_h = carry_h; a.carry_c = carry_c; a.dHin = dHin; a.dCin = dCin;
        if (scattered) { a.scat_h = ss_h; a.scat_c = ss_c; a.scat_idx = ss_idx; }
        launch_bwd(a, false, flops1, s);
        GGPM_CHECK_LAUNCH();
    }

    if (!weight_grads) return GGPM_OK;
    return lstm_weight_grads_impl(E1, H, depth, Hs, Ss, work, work_bytes, dWi_h, ld_dwi, dWo_h, ld_dwo, dWu_h, ld_dwu,
                                  dWf_h, ld_dwf, frozen != nullptr, lo, stream);
}

// where ggpm_lstm_backward left its di_pre / do_pre / du_pre stashes inside `work` (slot t-1 of each = backward step t)
extern "C" int ggpm_lstm_backward_stashes(float* work, int E1, int H, int depth, float** DI, float** DO, float** DU) {
    if (!work || !DI || !DO || !DU || E1 <= 0 || H <= 0 || depth <= 0) return GGPM_ERR_ARG;
    const size_t Hp = (size_t)ggpm_padded_hidden(H), slot = (size_t)E1 * Hp;
    *DI = work + 4 * ggpm_packed_matrix_slot((int)Hp);
    *DO = *DI + (size_t)depth * slot;
    *DU = *DO + (size_t)depth * slot;
    return GGPM_OK;
}

extern "C" int ggpm_lstm_backward(int E1, int H, int depth, const float* Xf, const float* Wi_h, int ld_wi,
                                  const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                                  int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col,
                                  const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                  const float* Cs, const float* Qs, const float* Ss, const float* Is,
                                  const float* Os, const float* Us, const float* Fs, const float* dHD, float* dXi,
                                  float* dXo, float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h,
                                  int ld_dwo, float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                                  size_t work_bytes, int weight_grads, ggpm_stream_t stream) {
    return lstm_backward_impl(E1, H, depth, Xf, Wi_h, ld_wi, Wo_h, ld_wo, Wu_h, ld_wu, Wf_h, ld_wf, pred_rowptr, pred_col,
                              succ_rowptr, succ_col, Hs, Cs, Qs, Ss, Is, Os, Us, Fs, dHD, dXi, dXo, dXu, dXf, dWi_h, ld_dwi,
                              dWo_h, ld_dwo, dWu_h, ld_dwu, dWf_h, ld_dwf, work, work_bytes, weight_grads, nullptr, nullptr,
                              nullptr, nullptr, stream);
}

// sparse_forward backward: takes dL/dh_D and dL/dc_D, additionally returns dHin / dCin (zero on the recomputed rows)
extern "C" int ggpm_lstm_sparse_backward(int E1, int H, int depth, const unsigned char* frozen, const float* Xf,
                                         const float* Wi_h, int ld_wi, const float* Wo_h, int ld_wo,
                                         const float* Wu_h, int ld_wu, const float* Wf_h, int ld_wf,
                                         const int32_t* pred_rowptr, const int32_t* pred_col,
                                         const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                         const float* Cs, const float* Qs, const float* Ss, const float* Is,
                                         const float* Os, const float* Us, const float* Fs, const float* dHD,
                                         const float* dCD, float* dHin, float* dCin, float* dXi, float* dXo,
                                         float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                                         float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                                         size_t work_bytes, ggpm_stream_t stream) {
    if (!frozen || !dHin || !dCin) return GGPM_ERR_ARG;
    return lstm_backward_impl(E1, H, depth, Xf, Wi_h, ld_wi, Wo_h, ld_wo, Wu_h, ld_wu, Wf_h, ld_wf, pred_rowptr, pred_col,
                              succ_rowptr, succ_col, Hs, Cs, Qs, Ss, Is, Os, Us, Fs, dHD, dXi, dXo, dXu, dXf, dWi_h, ld_dwi,
                              dWo_h, ld_dwo, dWu_h, ld_dwu, dWf_h, ld_dwf, work, work_bytes, 1, frozen, dCD, dHin, dCin,
                              stream);
}

// Weight gradients of the LSTM message function from the stashes ggpm_lstm_backward left in `work`.
static int lstm_weight_grads_impl(int E1, int H, int depth, const float* Hs, const float* Ss, float* work,
                                  size_t work_bytes, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h,
                                  int ld_dwu, float* dWf_h, int ld_dwf, bool with_slot0, int lo, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Hs || !Ss || !work || !dWi_h || !dWo_h || !dWu_h || !dWf_h)
        return GGPM_ERR_ARG;
    if (lo < 1 || lo > depth || with_slot0) lo = 1;       // backward steps depth .. lo ran (stash slots lo-1 .. depth-1)
    if (work_bytes < ggpm_lstm_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    float* w = work + 4 * ggpm_packed_matrix_slot(Hp);             // (layout of lstm_backward_impl)
    float* DI = w; w += (size_t)depth * slot;
    float* DO = w; w += (size_t)depth * slot;
    float* DU = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)depth * slot;
    w += 6 * slot;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);
    const int KD = (depth - lo + 1) * E1;
    // bf16 storage (as the forward / backward decided): bf16 stashes in the first half of their buffers, read as they are
    const bool st16 = ggpm_gate_dtype() == 1 && !with_slot0 && ggpm_bf16_storage_applies(E1, H);
    const int tall_mode = st16 ? 2 : (ggpm_gate_dtype() == 1 ? 1 : 0);
    const float* Sl = ggpm_slot_ptr(Ss, lo - 1, slot, st16);
    int rc;
    // the three or four contractions in ONE launch and one reduce (they share the split-K workspace)
    ggpm_gemm_problem gp[4] = {{ggpm_slot_ptr(DI, lo - 1, slot, st16), Hp, Sl, Hp, dWi_h, ld_dwi, H, nullptr, 0, GGPM_ACT_NONE, 0},
                               {ggpm_slot_ptr(DO, lo - 1, slot, st16), Hp, Sl, Hp, dWo_h, ld_dwo, H, nullptr, 0, GGPM_ACT_NONE, 0},
                               {ggpm_slot_ptr(DU, lo - 1, slot, st16), Hp, Sl, Hp, dWu_h, ld_dwu, H, nullptr, 0, GGPM_ACT_NONE, 0},
                               {nullptr, Hp, nullptr, Hp, dWf_h, ld_dwf, H, nullptr, 0, GGPM_ACT_NONE, 0}};
    int Ks[4] = {KD, KD, KD, 0};
    if (depth > lo || with_slot0) {
        const int first_slot = with_slot0 ? 0 : lo;    // dqf^t pairs with h^t; slot 0 exists for sparse_forward only
        gp[3].A = ggpm_slot_ptr(DQ, first_slot, slot, st16);
        gp[3].B = ggpm_slot_ptr(Hs, first_slot, slot, st16);
        Ks[3] = (depth - first_slot) * E1;
        rc = ggpm_gemm_tall_grouped(H, H, 4, gp, Ks, skws, skbytes, stream, tall_mode);
        if (rc) return rc;
    } else {
        rc = ggpm_gemm_tall_grouped(H, H, 3, gp, Ks, skws, skbytes, stream, tall_mode);
        if (rc) return rc;
        for (int r = 0; r < H; ++r) (void)hipMemsetAsync(dWf_h + (size_t)r * ld_dwf, 0, H * sizeof(float), s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// The hidden-half weight gradients of MANY sparse backward calls at once (see ggpm_gru_weight_grads_stacked).
extern "C" int ggpm_lstm_weight_grads_stacked(int rows, int rows_q, int H, const float* DI, const float* DO,
                                              const float* DU, const float* Ss, const float* DQ, const float* Hs,
                                              float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h, int ld_dwu,
                                              float* dWf_h, int ld_dwf, float* work, size_t work_bytes,
                                              ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (rows <= 0 || rows_q <= 0 || H <= 0 || !DI || !DO || !DU || !Ss || !DQ || !Hs || !dWi_h || !dWo_h || !dWu_h ||
        !dWf_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_weight_grads_stacked_workspace_bytes(H, rows > rows_q ? rows : rows_q)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    float* skws = work + (size_t)256 * Hp;
    const size_t skbytes = work_bytes - (size_t)256 * Hp * sizeof(float);
    const ggpm_gemm_problem gp[4] = {{DI, Hp, Ss, Hp, dWi_h, ld_dwi, H, nullptr, 0, GGPM_ACT_NONE, 0},
                                     {DO, Hp, Ss, Hp, dWo_h, ld_dwo, H, nullptr, 0, GGPM_ACT_NONE, 0},
                                     {DU, Hp, Ss, Hp, dWu_h, ld_dwu, H, nullptr, 0, GGPM_ACT_NONE, 0},
                                     {DQ, Hp, Hs, Hp, dWf_h, ld_dwf, H, nullptr, 0, GGPM_ACT_NONE, 0}};
    const int Ks[4] = {rows, rows, rows, rows_q};
    const int rc = ggpm_gemm_tall_grouped(H, H, 4, gp, Ks, skws, skbytes, stream);
    if (rc) return rc;
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

int ggpm_lstm_sparse_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, float* work, size_t work_bytes,
                                  float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h, int ld_dwu, float* dWf_h,
                                  int ld_dwf, ggpm_stream_t stream) {
    return lstm_weight_grads_impl(E1, H, depth, Hs, Ss, work, work_bytes, dWi_h, ld_dwi, dWo_h, ld_dwo, dWu_h, ld_dwu, dWf_h,
                                  ld_dwf, true, 1, stream);
}

extern "C" int ggpm_lstm_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, float* work,
                                      size_t work_bytes, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                                      float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, ggpm_stream_t stream) {
    return lstm_weight_grads_impl(E1, H, depth, Hs, Ss, work, work_bytes, dWi_h, ld_dwi, dWo_h, ld_dwo, dWu_h, ld_dwu,
                                  dWf_h, ld_dwf, false, ggpm_take_wgrad_lo(), stream);
}
