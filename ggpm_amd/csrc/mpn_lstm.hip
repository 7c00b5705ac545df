// Fused LSTM message step for gfx950 (all 32 shipped reference configs use rnn_type = LSTM).
//
// Reference arithmetic: LSTM.forward / LSTM.LSTM (ggpm/rnn.py:96-108, 85-94).  Restated over CSR
// predecessors; the x-halves of W_i/W_o/W/W_f (+ their biases) are hoisted (Xi, Xo, Xu, Xf) and the hidden
// half of the forget gate is applied once per message (qf = Wf_h h) instead of once per padded slot.
//
// Same workgroup geometry as mpn_gru.hip (16 message rows x all Hp columns, 4 waves, MFMA f32 16x16x4):
//   fwd  P1 gather (h_p, c_p, qf_p) -> s, fc tiles | P2 [Wi_h;Wo_h;Wu_h].s + gate math -> h', c'
//        P3 qf' = Wf_h h'
//   bwd  P1 successors -> dh partial, dqf, dc tiles | P2 dh += dqf.Wf_h ; gate derivatives
//        P3 dS = di_pre.Wi_h + do_pre.Wo_h + du_pre.Wu_h | P4 dXf accumulation over predecessors
#include "tile_mma.h"

namespace {

constexpr int ROWS = 16;
constexpr int NWAVES = 16;

struct LstmFwdArgs {
    int E1, Hp;
    const float *Xi, *Xo, *Xu, *Xf;
    const float *Hprev, *Cprev, *Qprev;
    float *Hnew, *Cnew, *Qnew;
    float *S, *I, *O, *U;            // stash slot (nullptr when not saving)
    const float *Wi, *Wo, *Wu, *Wf;  // packed
    const int32_t *rowptr, *col;
    int write_q;
};

template <int TPW, int NW, bool STASH>
__global__ void __launch_bounds__(NW * 64) lstm_step_fwd(LstmFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* Ts = lds;
    float* Tf = lds + ROWS * LD;
    float* Th = lds + 2 * ROWS * LD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;

    // ---- P1: gather predecessors (null slots read row 0: h = c = 0)
    for (int lr = wave; lr < ROWS; lr += NW) {
        const int row = r0 + lr;
        const GgpmRowList rl = ggpm_row_list(a.rowptr, row, a.E1);
        for (int c0 = 0; c0 < Hp; c0 += 256) {
            const int c = c0 + lane * 4;
            const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing
            float4 s = ggpm_zero4(), fc = ggpm_zero4();
            if (rl.n > 0) {
                const float4 xf = ggpm_ld4(a.Xf + (size_t)row * Hp + cs);
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 4) {
                        float4 h[4], cc[4], q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                            h[u] = ggpm_ld4(a.Hprev + p);
                            cc[u] = ggpm_ld4(a.Cprev + p);
                            q[u] = ggpm_ld4(a.Qprev + p);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            s = s + h[u];
                            fc = fc + ggpm_sigmoid4(xf + q[u]) * cc[u];
                        }
                    }
                }
            }
            if (on) {
                ggpm_st4(Ts + lr * LD + c, s);
                ggpm_st4(Tf + lr * LD + c, fc);
                if (STASH && row < a.E1) ggpm_st4(a.S + (size_t)row * Hp + c, s);
            }
        }
    }

    const int lr = lane & 15, row = r0 + lr;
    const bool live = row < a.E1;
    const bool act = live && row != 0;
    float4 xi[TPW], xo[TPW], xu[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + NW * i;
        const int c = 16 * t + 4 * (lane >> 4);
        const bool ok = act && t < NT;
        const size_t o = (size_t)row * Hp + c;
        xi[i] = ok ? ggpm_ld4(a.Xi + o) : ggpm_zero4();
        xo[i] = ok ? ggpm_ld4(a.Xo + o) : ggpm_zero4();
        xu[i] = ok ? ggpm_ld4(a.Xu + o) : ggpm_zero4();
    }
    __syncthreads();

    // ---- P2: [Wi_h; Wo_h; Wu_h] . s + gate math
    {
        f32x4 acci[TPW], acco[TPW], accu[TPW];
        ggpm_zero_acc<TPW>(acci);
        ggpm_zero_acc<TPW>(acco);
        ggpm_zero_acc<TPW>(accu);
        ggpm_tile_gemm<TPW, NW>(Ts, LD, a.Wi, KC, NT, wave, lane, acci);
        ggpm_tile_gemm<TPW, NW>(Ts, LD, a.Wo, KC, NT, wave, lane, acco);
        ggpm_tile_gemm<TPW, NW>(Ts, LD, a.Wu, KC, NT, wave, lane, accu);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            float4 h = ggpm_zero4(), cn = ggpm_zero4(), gi = ggpm_zero4(), go = ggpm_zero4(), gu = ggpm_zero4();
            if (act) {
                const float4 pi = ggpm_f4(acci[i]) + xi[i];
                const float4 po = ggpm_f4(acco[i]) + xo[i];
                const float4 pu = ggpm_f4(accu[i]) + xu[i];
                const float4 fc = ggpm_ld4(Tf + lr * LD + c);
                gi = ggpm_sigmoid4(pi);
                go = ggpm_sigmoid4(po);
                gu = make_float4(tanhf(pu.x), tanhf(pu.y), tanhf(pu.z), tanhf(pu.w));
                cn = gi * gu + fc;
                h = go * make_float4(tanhf(cn.x), tanhf(cn.y), tanhf(cn.z), tanhf(cn.w));
            }
            ggpm_st4(Th + lr * LD + c, h);
            if (live) {
                const size_t o = (size_t)row * Hp + c;
                ggpm_st4(a.Hnew + o, h);
                ggpm_st4(a.Cnew + o, cn);
                if (STASH) {
                    ggpm_st4(a.I + o, gi);
                    ggpm_st4(a.O + o, go);
                    ggpm_st4(a.U + o, gu);
                }
            }
        }
    }

    // ---- P3: qf' = Wf_h h'
    if (a.write_q) {
        __syncthreads();
        f32x4 accq[TPW];
        ggpm_zero_acc<TPW>(accq);
        ggpm_tile_gemm<TPW, NW>(Th, LD, a.Wf, KC, NT, wave, lane, accq);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT || !live) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            ggpm_st4(a.Qnew + (size_t)row * Hp + c, ggpm_f4(accq[i]));
        }
    }
}

struct LstmBwdArgs {
    int E1, Hp;
    int first, last;
    const float* Xf;
    const float *Ccur, *Qcur;        // Cs[t], Qs[t]        (P1, P2)
    const float *Cprv, *Qprv;        // Cs[t-1], Qs[t-1]    (P4)
    const float *I, *O, *U;          // stash slot t-1
    const float* dHD;
    const float *dSin, *dFCin;
    float *dSout, *dFCout;
    float* DQ;                       // dqf^t stash slot (nullptr when first)
    float *DI, *DO, *DU;             // stash slot t-1
    float *dXi, *dXo, *dXu, *dXf;    // running sums (zeroed by the driver)
    const float *WiT, *WoT, *WuT, *WfT;
    const int32_t *rowptr, *col, *srowptr, *scol;
};

template <int TPW, int NW>
__global__ void __launch_bounds__(NW * 64) lstm_step_bwd(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* T0 = lds;                  // dh partial -> di_pre
    float* T1 = lds + ROWS * LD;      // dqf -> do_pre
    float* T2 = lds + 2 * ROWS * LD;  // dc -> du_pre
    float* T3 = lds + 3 * ROWS * LD;  // dFC (= total dc of this depth)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;

    // ---- P1: gather over successors (null slots read row 0 where dS = dFC = 0)
    if (!a.first) {
        for (int lr = wave; lr < ROWS; lr += NW) {
            const int p = r0 + lr;
            const GgpmRowList rl = ggpm_row_list(a.srowptr, p, a.E1);
            for (int c0 = 0; c0 < Hp; c0 += 256) {
                const int c = c0 + lane * 4;
                const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing
                float4 dh = ggpm_zero4(), dq = ggpm_zero4(), dc = ggpm_zero4();
                if (rl.n > 0) {
                    const float4 cp = ggpm_ld4(a.Ccur + (size_t)p * Hp + cs);
                    const float4 qp = ggpm_ld4(a.Qcur + (size_t)p * Hp + cs);
                    for (int base = 0; base < rl.n; base += 64) {
                        const int chunk = ggpm_list_chunk(a.scol, rl, base, lane);
                        const int m = min(64, rl.n - base);
                        for (int j = 0; j < m; j += 4) {
                            float4 xf[4], dfc[4], ds[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const size_t e = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                                xf[u] = ggpm_ld4(a.Xf + e);
                                dfc[u] = ggpm_ld4(a.dFCin + e);
                                ds[u] = ggpm_ld4(a.dSin + e);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float4 f = ggpm_sigmoid4(xf[u] + qp);
                                const float4 one_f = make_float4(1.f - f.x, 1.f - f.y, 1.f - f.z, 1.f - f.w);
                                dh = dh + ds[u];
                                dc = dc + dfc[u] * f;
                                dq = dq + dfc[u] * cp * f * one_f;
                            }
                        }
                    }
                }
                if (on) {
                    ggpm_st4(T0 + lr * LD + c, dh);
                    ggpm_st4(T1 + lr * LD + c, dq);
                    ggpm_st4(T2 + lr * LD + c, dc);
                    if (p < a.E1) ggpm_st4(a.DQ + (size_t)p * Hp + c, dq);
                }
            }
        }
    }

    const int lr = lane & 15, row = r0 + lr;
    const bool live = row < a.E1;
    const bool act = live && row != 0;
    float4 st_i[TPW], st_o[TPW], st_u[TPW], st_c[TPW], st_dh[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + NW * i;
        const int c = 16 * t + 4 * (lane >> 4);
        const bool ok = act && t < NT;
        const size_t o = (size_t)row * Hp + c;
        st_i[i] = ok ? ggpm_ld4(a.I + o) : ggpm_zero4();
        st_o[i] = ok ? ggpm_ld4(a.O + o) : ggpm_zero4();
        st_u[i] = ok ? ggpm_ld4(a.U + o) : ggpm_zero4();
        st_c[i] = ok ? ggpm_ld4(a.Ccur + o) : ggpm_zero4();
        st_dh[i] = (ok && a.first) ? ggpm_ld4(a.dHD + o) : ggpm_zero4();
    }
    if (!a.first) __syncthreads();

    // ---- P2: dh = partial + dqf . Wf_h ; gate derivatives
    {
        f32x4 acc[TPW];
        ggpm_zero_acc<TPW>(acc);
        if (!a.first) {
            ggpm_tile_gemm<TPW, NW>(T1, LD, a.WfT, KC, NT, wave, lane, acc);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            float4 dip = ggpm_zero4(), dop = ggpm_zero4(), dup = ggpm_zero4(), dfc = ggpm_zero4();
            if (act) {
                float4 dh, dc;
                if (a.first) { dh = st_dh[i]; dc = ggpm_zero4(); }
                else { dh = ggpm_f4(acc[i]) + ggpm_ld4(T0 + lr * LD + c); dc = ggpm_ld4(T2 + lr * LD + c); }
                const float4 gi = st_i[i], go = st_o[i], gu = st_u[i], cc = st_c[i];
                const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, dcv[4] = {dc.x, dc.y, dc.z, dc.w};
                const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, ov[4] = {go.x, go.y, go.z, go.w};
                const float uv[4] = {gu.x, gu.y, gu.z, gu.w}, cv[4] = {cc.x, cc.y, cc.z, cc.w};
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
            }
            ggpm_st4(T0 + lr * LD + c, dip);
            ggpm_st4(T1 + lr * LD + c, dop);
            ggpm_st4(T2 + lr * LD + c, dup);
            ggpm_st4(T3 + lr * LD + c, dfc);
            if (live) {
                const size_t o = (size_t)row * Hp + c;
                ggpm_st4(a.DI + o, dip);
                ggpm_st4(a.DO + o, dop);
                ggpm_st4(a.DU + o, dup);
                ggpm_st4(a.dFCout + o, dfc);
                ggpm_st4(a.dXi + o, ggpm_ld4(a.dXi + o) + dip);
                ggpm_st4(a.dXo + o, ggpm_ld4(a.dXo + o) + dop);
                ggpm_st4(a.dXu + o, ggpm_ld4(a.dXu + o) + dup);
            }
        }
    }

    if (!a.last) {
        __syncthreads();
        // ---- P3: dS = di_pre . Wi_h + do_pre . Wo_h + du_pre . Wu_h
        f32x4 accs[TPW];
        ggpm_zero_acc<TPW>(accs);
        ggpm_tile_gemm<TPW, NW>(T0, LD, a.WiT, KC, NT, wave, lane, accs);
        ggpm_tile_gemm<TPW, NW>(T1, LD, a.WoT, KC, NT, wave, lane, accs);
        ggpm_tile_gemm<TPW, NW>(T2, LD, a.WuT, KC, NT, wave, lane, accs);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT || !live) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            ggpm_st4(a.dSout + (size_t)row * Hp + c, ggpm_f4(accs[i]));
        }

        // ---- P4: dXf_e += sum_p dFC_e * c_p * f(1-f),  f = sigmoid(Xf_e + qf_p)   (null slots: c[0] == 0)
        for (int l2 = wave; l2 < ROWS; l2 += NW) {
            const int e = r0 + l2;
            const GgpmRowList rl = ggpm_row_list(a.rowptr, e, a.E1);
            if (rl.n <= 0) continue;
            for (int c0 = 0; c0 < Hp; c0 += 256) {
                const int c = c0 + lane * 4;
                const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing
                const size_t o = (size_t)e * Hp + cs;
                const float4 xf = ggpm_ld4(a.Xf + o);
                const float4 old = ggpm_ld4(a.dXf + o);
                const float4 dfc = ggpm_ld4(T3 + l2 * LD + cs);
                float4 accx = ggpm_zero4();
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 4) {
                        float4 cc[4], q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                            cc[u] = ggpm_ld4(a.Cprv + p);
                            q[u] = ggpm_ld4(a.Qprv + p);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 f = ggpm_sigmoid4(xf + q[u]);
                            const float4 one_f = make_float4(1.f - f.x, 1.f - f.y, 1.f - f.z, 1.f - f.w);
                            accx = accx + dfc * cc[u] * f * one_f;
                        }
                    }
                }
                if (on) ggpm_st4(a.dXf + o, old + accx);
            }
        }
    }
}

template <typename K>
inline void set_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
}

template <int TPW>
int launch_fwd(const LstmFwdArgs& a, bool stash, size_t lds_bytes, int grid, hipStream_t s) {
    if (stash) {
        set_lds(lstm_step_fwd<TPW, NWAVES, true>, lds_bytes);
        lstm_step_fwd<TPW, NWAVES, true><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    } else {
        set_lds(lstm_step_fwd<TPW, NWAVES, false>, lds_bytes);
        lstm_step_fwd<TPW, NWAVES, false><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    }
    return 0;
}

template <int TPW>
int launch_bwd(const LstmBwdArgs& a, size_t lds_bytes, int grid, hipStream_t s) {
    set_lds(lstm_step_bwd<TPW, NWAVES>, lds_bytes);
    lstm_step_bwd<TPW, NWAVES><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    return 0;
}

}  // namespace

#define GGPM_DISPATCH_TPW(tpw, CALL)            \
    switch (tpw) {                              \
        case 1: CALL(1); break;                 \
        case 2: CALL(2); break;                 \
        case 3: CALL(3); break;                 \
        default: return GGPM_ERR_UNSUPPORTED;   \
    }

extern "C" size_t ggpm_lstm_pack_floats(int H) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    return 4 * Hp * Hp;
}

extern "C" int ggpm_lstm_forward(int E1, int H, int depth, const float* Xi, const float* Xo, const float* Xu,
                                 const float* Xf, const float* Wi_h, int ld_wi, const float* Wo_h, int ld_wo,
                                 const float* Wu_h, int ld_wu, const float* Wf_h, int ld_wf,
                                 const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs, float* Cs,
                                 float* Qs, float* Ss, float* Is, float* Os, float* Us, float* wpack,
                                 int save_for_backward, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xi || !Xo || !Xu || !Xf || !Wi_h || !Wo_h || !Wu_h || !Wf_h ||
        !pred_rowptr || !pred_col || !Hs || !Cs || !Qs || !wpack)
        return GGPM_ERR_ARG;
    if (save_for_backward && (!Ss || !Is || !Os || !Us)) return GGPM_ERR_ARG;
    const int Hp = ggpm_padded_hidden(H);
    const int NT = Hp / 16, tpw = ggpm_ceil_div(NT, NWAVES);
    const size_t lds_bytes = (size_t)3 * ROWS * (Hp + 4) * sizeof(float);
    if (tpw > 3 || lds_bytes > 160 * 1024) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    float* pWi = wpack; float* pWo = wpack + HH; float* pWu = wpack + 2 * HH; float* pWf = wpack + 3 * HH;
    ggpm_launch_pack(Wi_h, ld_wi, H, Hp, 0, pWi, s);
    ggpm_launch_pack(Wo_h, ld_wo, H, Hp, 0, pWo, s);
    ggpm_launch_pack(Wu_h, ld_wu, H, Hp, 0, pWu, s);
    ggpm_launch_pack(Wf_h, ld_wf, H, Hp, 0, pWf, s);
    (void)hipMemsetAsync(Hs, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(Cs, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(Qs, 0, slot * sizeof(float), s);

    const int grid = ggpm_ceil_div(E1, ROWS);
    const double flops = 2.0 * 4.0 * (double)(E1 - 1) * H * H;
    for (int t = 1; t <= depth; ++t) {
        LstmFwdArgs a;
        a.E1 = E1; a.Hp = Hp; a.Xi = Xi; a.Xo = Xo; a.Xu = Xu; a.Xf = Xf;
        a.Wi = pWi; a.Wo = pWo; a.Wu = pWu; a.Wf = pWf; a.rowptr = pred_rowptr; a.col = pred_col;
        a.write_q = (t < depth);
        if (save_for_backward) {
            a.Hprev = Hs + (size_t)(t - 1) * slot; a.Hnew = Hs + (size_t)t * slot;
            a.Cprev = Cs + (size_t)(t - 1) * slot; a.Cnew = Cs + (size_t)t * slot;
            a.Qprev = Qs + (size_t)(t - 1) * slot; a.Qnew = (t < depth) ? Qs + (size_t)t * slot : nullptr;
            a.S = Ss + (size_t)(t - 1) * slot; a.I = Is + (size_t)(t - 1) * slot;
            a.O = Os + (size_t)(t - 1) * slot; a.U = Us + (size_t)(t - 1) * slot;
        } else {
            a.Hprev = Hs + (size_t)((t - 1) & 1) * slot; a.Hnew = Hs + (size_t)(t & 1) * slot;
            a.Cprev = Cs + (size_t)((t - 1) & 1) * slot; a.Cnew = Cs + (size_t)(t & 1) * slot;
            a.Qprev = Qs + (size_t)((t - 1) & 1) * slot; a.Qnew = Qs + (size_t)(t & 1) * slot;
            a.S = a.I = a.O = a.U = nullptr;
        }
        ggpm_timing_begin(2, s, flops);
#define CALL(T) launch_fwd<T>(a, save_for_backward != 0, lds_bytes, grid, s)
        GGPM_DISPATCH_TPW(tpw, CALL)
#undef CALL
        ggpm_timing_end(2, s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" size_t ggpm_lstm_backward_workspace_bytes(int E1, int H, int depth) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    const size_t slot = (size_t)E1 * Hp;
    size_t f = 0;
    f += 3 * (size_t)depth * slot;                     // DI, DO, DU
    f += (size_t)(depth > 1 ? depth - 1 : 1) * slot;   // DQ
    f += 4 * slot;                                     // dS / dFC double buffers
    f += 4 * Hp * Hp;                                  // packed transposes
    size_t bytes = f * sizeof(float);
    bytes += ggpm_gemm_workspace_bytes(H, H, depth * E1);
    return bytes + 256;
}

extern "C" int ggpm_lstm_backward(int E1, int H, int depth, const float* Xf, const float* Wi_h, int ld_wi,
                                  const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                                  int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col,
                                  const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                  const float* Cs, const float* Qs, const float* Ss, const float* Is,
                                  const float* Os, const float* Us, const float* dHD, float* dXi, float* dXo,
                                  float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                                  float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                                  size_t work_bytes, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xf || !Wi_h || !Wo_h || !Wu_h || !Wf_h || !pred_rowptr || !pred_col ||
        !succ_rowptr || !succ_col || !Hs || !Cs || !Qs || !Ss || !Is || !Os || !Us || !dHD || !dXi || !dXo || !dXu ||
        !dXf || !dWi_h || !dWo_h || !dWu_h || !dWf_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_lstm_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    const int NT = Hp / 16, tpw = ggpm_ceil_div(NT, NWAVES);
    const size_t lds_bytes = (size_t)4 * ROWS * (Hp + 4) * sizeof(float);
    if (tpw > 3 || lds_bytes > 160 * 1024) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;

    float* w = work;
    float* DI = w; w += (size_t)depth * slot;
    float* DO = w; w += (size_t)depth * slot;
    float* DU = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)(depth > 1 ? depth - 1 : 1) * slot;
    float* dSb[2]; float* dFb[2];
    dSb[0] = w; w += slot; dSb[1] = w; w += slot; dFb[0] = w; w += slot; dFb[1] = w; w += slot;
    float* pWiT = w; w += HH; float* pWoT = w; w += HH; float* pWuT = w; w += HH; float* pWfT = w; w += HH;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);

    ggpm_launch_pack(Wi_h, ld_wi, H, Hp, 1, pWiT, s);
    ggpm_launch_pack(Wo_h, ld_wo, H, Hp, 1, pWoT, s);
    ggpm_launch_pack(Wu_h, ld_wu, H, Hp, 1, pWuT, s);
    ggpm_launch_pack(Wf_h, ld_wf, H, Hp, 1, pWfT, s);
    (void)hipMemsetAsync(dXi, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(dXo, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(dXu, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(dXf, 0, slot * sizeof(float), s);

    const int grid = ggpm_ceil_div(E1, ROWS);
    const double flops = 2.0 * 4.0 * (double)(E1 - 1) * H * H;
    for (int t = depth; t >= 1; --t) {
        LstmBwdArgs a;
        a.E1 = E1; a.Hp = Hp; a.first = (t == depth); a.last = (t == 1);
        a.Xf = Xf;
        a.Ccur = Cs + (size_t)t * slot;
        a.Qcur = (t < depth) ? Qs + (size_t)t * slot : nullptr;
        a.Cprv = Cs + (size_t)(t - 1) * slot; a.Qprv = Qs + (size_t)(t - 1) * slot;
        a.I = Is + (size_t)(t - 1) * slot; a.O = Os + (size_t)(t - 1) * slot; a.U = Us + (size_t)(t - 1) * slot;
        a.dHD = dHD;
        a.dSin = dSb[(t + 1) & 1]; a.dFCin = dFb[(t + 1) & 1];
        a.dSout = dSb[t & 1]; a.dFCout = dFb[t & 1];
        a.DQ = (t < depth) ? DQ + (size_t)(t - 1) * slot : nullptr;
        a.DI = DI + (size_t)(t - 1) * slot; a.DO = DO + (size_t)(t - 1) * slot; a.DU = DU + (size_t)(t - 1) * slot;
        a.dXi = dXi; a.dXo = dXo; a.dXu = dXu; a.dXf = dXf;
        a.WiT = pWiT; a.WoT = pWoT; a.WuT = pWuT; a.WfT = pWfT;
        a.rowptr = pred_rowptr; a.col = pred_col; a.srowptr = succ_rowptr; a.scol = succ_col;
        ggpm_timing_begin(3, s, flops);
#define CALL(T) launch_bwd<T>(a, lds_bytes, grid, s)
        GGPM_DISPATCH_TPW(tpw, CALL)
#undef CALL
        ggpm_timing_end(3, s);
    }
    GGPM_CHECK_LAUNCH();

    const int KD = depth * E1;
    int rc;
    rc = ggpm_gemm(1, 0, H, H, KD, DI, Hp, Ss, Hp, dWi_h, ld_dwi, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
    if (rc) return rc;
    rc = ggpm_gemm(1, 0, H, H, KD, DO, Hp, Ss, Hp, dWo_h, ld_dwo, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
    if (rc) return rc;
    rc = ggpm_gemm(1, 0, H, H, KD, DU, Hp, Ss, Hp, dWu_h, ld_dwu, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
    if (rc) return rc;
    if (depth > 1) {
        const int KQ = (depth - 1) * E1;
        rc = ggpm_gemm(1, 0, H, H, KQ, DQ, Hp, Hs + slot, Hp, dWf_h, ld_dwf, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
        if (rc) return rc;
    } else {
        for (int r = 0; r < H; ++r) (void)hipMemsetAsync(dWf_h + (size_t)r * ld_dwf, 0, H * sizeof(float), s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
