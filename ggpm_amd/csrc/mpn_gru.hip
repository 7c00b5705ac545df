// Fused GRU message step for gfx950 -- the hot loop of the hierarchical encoder.
//
// Reference arithmetic: GRU.forward / GRU.GRU (ggpm/rnn.py:41-50, 25-39) + index_select_ND
// (ggpm/nnutils.py:65-70).  Restated over CSR predecessor lists with the x-halves of W_z/W_r/W_h hoisted
// out of the depth loop and U_r applied once per message (q = U_r h + b_u) instead of once per padded slot.
//
// One launch per depth; a workgroup (4 waves) owns 16 message rows for all Hp feature columns:
//   P1  CSR gather of predecessor rows (h_p, q_p) from L2 -> s, g tiles in LDS   (coalesced 16B/lane)
//   P2  gate GEMMs on MFMA f32 16x16x4:  Wz_h . s  and  Wh_h . g  (weights streamed packed from L2)
//       + fused gate math -> h' (global + LDS tile)
//   P3  q' = U_r h' + b_u on MFMA from the LDS tile
// Backward mirrors it (gather over SUCCESSORS through the transposed CSR, so no atomics):
//   P1  dq, dh-partial tiles from successors   P2  dh = partial + dq.U_r ; gate derivatives
//   P3  dG = dm_pre.Wh_h, dS = ds_dir + dz_pre.Wz_h   P4  dXr accumulation over predecessors
// Weight gradients are three tall split-K GEMMs over the [depth*E1, Hp] stashes (gemm.hip).
#include "tile_mma.h"
#include <cstdlib>

__global__ void ggpm_pack_weight_kernel(const float* __restrict__ W, int ldw, int H, int Hp, int transpose,
                                        float* __restrict__ dst) {
    const int KC = Hp / 16;
    const int lane = threadIdx.x;            // 64
    const int kc = blockIdx.x, t = blockIdx.y;
    const int out = 16 * t + (lane & 15);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 16 * kc + 4 * (lane >> 4) + i;
        float x = 0.f;
        if (out < H && k < H) x = transpose ? W[(size_t)k * ldw + out] : W[(size_t)out * ldw + k];
        v[i] = x;
    }
    ggpm_st4(dst + ggpm_pack_index(t, kc, KC, lane), make_float4(v[0], v[1], v[2], v[3]));
}

void ggpm_launch_pack(const float* W, int ldw, int H, int Hp, int transpose, float* dst, hipStream_t s) {
    dim3 grid(Hp / 16, Hp / 16);
    ggpm_pack_weight_kernel<<<grid, 64, 0, s>>>(W, ldw, H, Hp, transpose, dst);
}

namespace {

constexpr int ROWS = 16;
constexpr int NWAVES = 16;   // waves per workgroup: 4 per SIMD hide the L2 latency of the weight stream

struct GruFwdArgs {
    int E1, Hp;
    const float *Xz, *Xr, *Xh;
    const float *Hprev, *Qprev;
    float *Hnew, *Qnew;
    float *S, *G, *Z, *M;          // stash slot of this depth (nullptr when not saving)
    const float *Wz, *Wh, *Ur;     // packed
    const float* bu;               // [Hp] zero padded
    const int32_t *rowptr, *col;
    int write_q;                   // 0 on the last depth: q^depth is never consumed, skip P3
    int ablate;                    // timing experiments only (GGPM_ABLATE): 1 no gather, 2 no GEMM, 4 no P3
};

__global__ void gru_init_state(float* __restrict__ H0, float* __restrict__ Q0, const float* __restrict__ bu,
                               int E1, int Hp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= Hp) return;
    H0[(size_t)r * Hp + c] = 0.f;
    Q0[(size_t)r * Hp + c] = bu[c];
}

__global__ void pad_bias(const float* __restrict__ b, int H, int Hp, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Hp) out[c] = (c < H) ? b[c] : 0.f;
}

template <int TPW, int NW, bool STASH>
__global__ void __launch_bounds__(NW * 64) gru_step_fwd(GruFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* Ts = lds;                 // s tile
    float* Tg = lds + ROWS * LD;     // g tile
    float* Th = lds + 2 * ROWS * LD; // h' tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;

    // ---- P1: gather predecessors (one wave per row; 4 independent predecessor rows in flight)
    for (int lr = wave; lr < ROWS; lr += NW) {
        const int row = r0 + lr;
        const GgpmRowList rl = ggpm_row_list(a.rowptr, row, a.E1);
        for (int c0 = 0; c0 < Hp; c0 += 256) {
            const int c = c0 + lane * 4;
            const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing
            float4 s = ggpm_zero4(), g = ggpm_zero4();
            if (rl.n > 0 && !(a.ablate & 1)) {
                const float4 xr = ggpm_ld4(a.Xr + (size_t)row * Hp + cs);
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 4) {
                        float4 h[4], q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                            h[u] = ggpm_ld4(a.Hprev + p);
                            q[u] = ggpm_ld4(a.Qprev + p);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            s = s + h[u];
                            g = g + ggpm_sigmoid4(xr + q[u]) * h[u];   // null slots: h[0] == 0
                        }
                    }
                }
            }
            if (on) {
                ggpm_st4(Ts + lr * LD + c, s);
                ggpm_st4(Tg + lr * LD + c, g);
                if (STASH && row < a.E1) {
                    ggpm_st4(a.S + (size_t)row * Hp + c, s);
                    ggpm_st4(a.G + (size_t)row * Hp + c, g);
                }
            }
        }
    }

    // prefetch the hoisted input terms of this wave's tiles while the other waves finish their gathers
    const int lr = lane & 15, row = r0 + lr;
    const bool live = row < a.E1;
    const bool act = live && row != 0;
    float4 xz[TPW], xh[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + NW * i;
        const int c = 16 * t + 4 * (lane >> 4);
        const bool ok = act && t < NT;
        xz[i] = ok ? ggpm_ld4(a.Xz + (size_t)row * Hp + c) : ggpm_zero4();
        xh[i] = ok ? ggpm_ld4(a.Xh + (size_t)row * Hp + c) : ggpm_zero4();
    }
    __syncthreads();

    // ---- P2: gate GEMMs + gate math
    {
        f32x4 accz[TPW], accm[TPW];
        ggpm_zero_acc<TPW>(accz);
        ggpm_zero_acc<TPW>(accm);
        if (!(a.ablate & 2)) {
            ggpm_tile_gemm<TPW, NW>(Ts, LD, a.Wz, KC, NT, wave, lane, accz);
            ggpm_tile_gemm<TPW, NW>(Tg, LD, a.Wh, KC, NT, wave, lane, accm);
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            float4 h = ggpm_zero4(), z = ggpm_zero4(), m = ggpm_zero4();
            if (act) {
                const float4 s = ggpm_ld4(Ts + lr * LD + c);
                const float4 pz = ggpm_f4(accz[i]) + xz[i], pm = ggpm_f4(accm[i]) + xh[i];
                z = ggpm_sigmoid4(pz);
                m = make_float4(tanhf(pm.x), tanhf(pm.y), tanhf(pm.z), tanhf(pm.w));
                h = make_float4((1.f - z.x) * s.x + z.x * m.x, (1.f - z.y) * s.y + z.y * m.y,
                                (1.f - z.z) * s.z + z.z * m.z, (1.f - z.w) * s.w + z.w * m.w);
            }
            ggpm_st4(Th + lr * LD + c, h);
            if (live) {
                ggpm_st4(a.Hnew + (size_t)row * Hp + c, h);
                if (STASH) {
                    ggpm_st4(a.Z + (size_t)row * Hp + c, z);
                    ggpm_st4(a.M + (size_t)row * Hp + c, m);
                }
            }
        }
    }

    // ---- P3: q' = U_r h' + b_u
    if (a.write_q && !(a.ablate & 4)) {
        __syncthreads();
        f32x4 accq[TPW];
        ggpm_zero_acc<TPW>(accq);
        ggpm_tile_gemm<TPW, NW>(Th, LD, a.Ur, KC, NT, wave, lane, accq);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT || !live) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            ggpm_st4(a.Qnew + (size_t)row * Hp + c, ggpm_f4(accq[i]) + ggpm_ld4(a.bu + c));
        }
    }
}

struct GruBwdArgs {
    int E1, Hp;
    int first, last;               // first: t == depth (dH comes from dHD); last: t == 1
    const float* Xr;
    const float *Hcur, *Qcur;      // Hs[t],   Qs[t]      (successor side of P1; unused when first)
    const float *Hprv, *Qprv;      // Hs[t-1], Qs[t-1]    (predecessor side of P4; unused when last)
    const float *S, *G, *Z, *M;    // stash slot t-1
    const float* dHD;              // [E1,Hp], used when first
    const float *dSin, *dGin;      // from launch t+1
    float *dSout, *dGout;          // for launch t-1
    float* DQ;                     // stash slot for dq^t (nullptr when first)
    float *DMP, *DZP;              // stash slot t-1
    float *dXz, *dXr, *dXh;        // running sums (zeroed by the driver)
    const float *WzT, *WhT, *UrT;  // packed transposes
    const int32_t *rowptr, *col;   // predecessors
    const int32_t *srowptr, *scol; // successors
};

template <int TPW, int NW>
__global__ void __launch_bounds__(NW * 64) gru_step_bwd(GruBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Hp = a.Hp, LD = Hp + 4, KC = Hp / 16, NT = Hp / 16;
    float* T0 = lds;                  // dh partial -> ds_dir
    float* T1 = lds + ROWS * LD;      // dq -> dz_pre
    float* T2 = lds + 2 * ROWS * LD;  // dm_pre -> dG
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * ROWS;

    // ---- P1: gather over successors of p:  dh_p += dS_e + dG_e*r ; dq_p += dG_e * h_p * r(1-r)
    // (null slots read row 0, where dS = dG = 0)
    if (!a.first) {
        for (int lr = wave; lr < ROWS; lr += NW) {
            const int p = r0 + lr;
            const GgpmRowList rl = ggpm_row_list(a.srowptr, p, a.E1);
            for (int c0 = 0; c0 < Hp; c0 += 256) {
                const int c = c0 + lane * 4;
                const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing
                float4 dh = ggpm_zero4(), dq = ggpm_zero4();
                if (rl.n > 0) {   // wave-uniform: every lane takes part in the list broadcast below
                    const float4 hp = ggpm_ld4(a.Hcur + (size_t)p * Hp + cs);
                    const float4 qp = ggpm_ld4(a.Qcur + (size_t)p * Hp + cs);
                    for (int base = 0; base < rl.n; base += 64) {
                        const int chunk = ggpm_list_chunk(a.scol, rl, base, lane);
                        const int m = min(64, rl.n - base);
                        for (int j = 0; j < m; j += 4) {
                            float4 xr[4], dg[4], ds[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const size_t e = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                                xr[u] = ggpm_ld4(a.Xr + e);
                                dg[u] = ggpm_ld4(a.dGin + e);
                                ds[u] = ggpm_ld4(a.dSin + e);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float4 r = ggpm_sigmoid4(xr[u] + qp);
                                const float4 one_r = make_float4(1.f - r.x, 1.f - r.y, 1.f - r.z, 1.f - r.w);
                                dh = dh + ds[u] + dg[u] * r;
                                dq = dq + dg[u] * hp * r * one_r;
                            }
                        }
                    }
                }
                if (on) {
                    ggpm_st4(T0 + lr * LD + c, dh);
                    ggpm_st4(T1 + lr * LD + c, dq);
                    if (p < a.E1) ggpm_st4(a.DQ + (size_t)p * Hp + c, dq);
                }
            }
        }
    }

    // prefetch this wave's stash operands for P2
    const int lr = lane & 15, row = r0 + lr;
    const bool live = row < a.E1;
    const bool act = live && row != 0;
    float4 st_s[TPW], st_z[TPW], st_m[TPW], st_dh[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + NW * i;
        const int c = 16 * t + 4 * (lane >> 4);
        const bool ok = act && t < NT;
        const size_t o = (size_t)row * Hp + c;
        st_s[i] = ok ? ggpm_ld4(a.S + o) : ggpm_zero4();
        st_z[i] = ok ? ggpm_ld4(a.Z + o) : ggpm_zero4();
        st_m[i] = ok ? ggpm_ld4(a.M + o) : ggpm_zero4();
        st_dh[i] = (ok && a.first) ? ggpm_ld4(a.dHD + o) : ggpm_zero4();
    }
    if (!a.first) __syncthreads();

    // ---- P2: dh = partial + dq . U_r ; gate derivatives
    {
        f32x4 acc[TPW];
        ggpm_zero_acc<TPW>(acc);
        if (!a.first) {
            ggpm_tile_gemm<TPW, NW>(T1, LD, a.UrT, KC, NT, wave, lane, acc);
            __syncthreads();   // every wave is done reading T1 before it is overwritten below
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            float4 dsdir = ggpm_zero4(), dzp = ggpm_zero4(), dmp = ggpm_zero4();
            if (act) {
                const float4 dh = a.first ? st_dh[i] : (ggpm_f4(acc[i]) + ggpm_ld4(T0 + lr * LD + c));
                const float4 s = st_s[i], z = st_z[i], m = st_m[i];
                const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, sv[4] = {s.x, s.y, s.z, s.w};
                const float zv[4] = {z.x, z.y, z.z, z.w}, mv[4] = {m.x, m.y, m.z, m.w};
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
            }
            ggpm_st4(T0 + lr * LD + c, dsdir);
            ggpm_st4(T1 + lr * LD + c, dzp);
            ggpm_st4(T2 + lr * LD + c, dmp);
            if (live) {
                const size_t o = (size_t)row * Hp + c;
                ggpm_st4(a.DZP + o, dzp);
                ggpm_st4(a.DMP + o, dmp);
                ggpm_st4(a.dXz + o, ggpm_ld4(a.dXz + o) + dzp);
                ggpm_st4(a.dXh + o, ggpm_ld4(a.dXh + o) + dmp);
            }
        }
    }

    // ---- P3: dG = dm_pre . Wh_h ; dS = ds_dir + dz_pre . Wz_h   (consumed by launch t-1)
    if (!a.last) {
        __syncthreads();
        f32x4 accg[TPW], accs[TPW];
        ggpm_zero_acc<TPW>(accg);
        ggpm_zero_acc<TPW>(accs);
        ggpm_tile_gemm<TPW, NW>(T2, LD, a.WhT, KC, NT, wave, lane, accg);
        ggpm_tile_gemm<TPW, NW>(T1, LD, a.WzT, KC, NT, wave, lane, accs);
        __syncthreads();       // T2 is about to be overwritten with dG
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            if (t >= NT) continue;
            const int c = 16 * t + 4 * (lane >> 4);
            const float4 dg = ggpm_f4(accg[i]);
            const float4 ds = ggpm_f4(accs[i]) + ggpm_ld4(T0 + lr * LD + c);
            ggpm_st4(T2 + lr * LD + c, dg);
            if (live) {
                const size_t o = (size_t)row * Hp + c;
                ggpm_st4(a.dGout + o, dg);
                ggpm_st4(a.dSout + o, ds);
            }
        }
        __syncthreads();

        // ---- P4: dXr_e += sum_p dG_e * h_p * r(1-r),  r = sigmoid(Xr_e + q_p)   (null slots: h[0] == 0)
        for (int l2 = wave; l2 < ROWS; l2 += NW) {
            const int e = r0 + l2;
            const GgpmRowList rl = ggpm_row_list(a.rowptr, e, a.E1);
            if (rl.n <= 0) continue;
            for (int c0 = 0; c0 < Hp; c0 += 256) {
                const int c = c0 + lane * 4;
                const bool on = c < Hp;
            const int cs = on ? c : 0;   // lanes past the row end load column 0 (no branch) and store nothing     // no early exit: every lane takes part in the list broadcast
                const size_t o = (size_t)e * Hp + cs;
                const float4 xr = ggpm_ld4(a.Xr + o);
                const float4 dxr_old = ggpm_ld4(a.dXr + o);
                const float4 dg = ggpm_ld4(T2 + l2 * LD + cs);
                float4 accx = ggpm_zero4();
                for (int base = 0; base < rl.n; base += 64) {
                    const int chunk = ggpm_list_chunk(a.col, rl, base, lane);
                    const int m = min(64, rl.n - base);
                    for (int j = 0; j < m; j += 4) {
                        float4 h[4], q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t p = (size_t)ggpm_list_at(chunk, j + u, m) * Hp + cs;
                            h[u] = ggpm_ld4(a.Hprv + p);
                            q[u] = ggpm_ld4(a.Qprv + p);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 r = ggpm_sigmoid4(xr + q[u]);
                            const float4 one_r = make_float4(1.f - r.x, 1.f - r.y, 1.f - r.z, 1.f - r.w);
                            accx = accx + dg * h[u] * r * one_r;
                        }
                    }
                }
                if (on) ggpm_st4(a.dXr + o, dxr_old + accx);
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
int launch_fwd(const GruFwdArgs& a, bool stash, size_t lds_bytes, int grid, hipStream_t s) {
    if (stash) {
        set_lds(gru_step_fwd<TPW, NWAVES, true>, lds_bytes);
        gru_step_fwd<TPW, NWAVES, true><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    } else {
        set_lds(gru_step_fwd<TPW, NWAVES, false>, lds_bytes);
        gru_step_fwd<TPW, NWAVES, false><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    }
    return 0;
}

template <int TPW>
int launch_bwd(const GruBwdArgs& a, size_t lds_bytes, int grid, hipStream_t s) {
    set_lds(gru_step_bwd<TPW, NWAVES>, lds_bytes);
    gru_step_bwd<TPW, NWAVES><<<grid, NWAVES * 64, lds_bytes, s>>>(a);
    return 0;
}

}  // namespace

extern "C" size_t ggpm_gru_pack_floats(int H) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    return 3 * Hp * Hp + Hp;
}

#define GGPM_DISPATCH_TPW(tpw, CALL)            \
    switch (tpw) {                              \
        case 1: CALL(1); break;                 \
        case 2: CALL(2); break;                 \
        case 3: CALL(3); break;                 \
        default: return GGPM_ERR_UNSUPPORTED;   \
    }

extern "C" int ggpm_gru_forward(int E1, int H, int depth, const float* Xz, const float* Xr, const float* Xh,
                                const float* Wz_h, int ld_wz, const float* Ur, int ld_ur, const float* bu,
                                const float* Wh_h, int ld_wh, const int32_t* pred_rowptr,
                                const int32_t* pred_col, float* Hs, float* Qs, float* Ss, float* Gs, float* Zs,
                                float* Ms, float* wpack, int save_for_backward, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xz || !Xr || !Xh || !Wz_h || !Ur || !bu || !Wh_h || !pred_rowptr ||
        !pred_col || !Hs || !Qs || !wpack)
        return GGPM_ERR_ARG;
    if (save_for_backward && (!Ss || !Gs || !Zs || !Ms)) return GGPM_ERR_ARG;
    const int Hp = ggpm_padded_hidden(H);
    const int NT = Hp / 16, tpw = ggpm_ceil_div(NT, NWAVES);
    const size_t lds_bytes = (size_t)3 * ROWS * (Hp + 4) * sizeof(float);
    if (tpw > 3 || lds_bytes > 160 * 1024) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;
    float* pWz = wpack; float* pWh = wpack + HH; float* pUr = wpack + 2 * HH; float* pbu = wpack + 3 * HH;
    ggpm_launch_pack(Wz_h, ld_wz, H, Hp, 0, pWz, s);
    ggpm_launch_pack(Wh_h, ld_wh, H, Hp, 0, pWh, s);
    ggpm_launch_pack(Ur, ld_ur, H, Hp, 0, pUr, s);
    pad_bias<<<ggpm_ceil_div(Hp, 256), 256, 0, s>>>(bu, H, Hp, pbu);
    dim3 ig(ggpm_ceil_div(Hp, 256), E1);
    gru_init_state<<<ig, 256, 0, s>>>(Hs, Qs, pbu, E1, Hp);

    const int grid = ggpm_ceil_div(E1, ROWS);
    const double flops = 2.0 * 3.0 * (double)(E1 - 1) * H * H;   // algorithmic: 3 gate products per message
    for (int t = 1; t <= depth; ++t) {
        GruFwdArgs a;
        a.E1 = E1; a.Hp = Hp; a.Xz = Xz; a.Xr = Xr; a.Xh = Xh;
        a.Wz = pWz; a.Wh = pWh; a.Ur = pUr; a.bu = pbu; a.rowptr = pred_rowptr; a.col = pred_col;
        if (save_for_backward) {
            a.Hprev = Hs + (size_t)(t - 1) * slot; a.Hnew = Hs + (size_t)t * slot;
            a.Qprev = Qs + (size_t)(t - 1) * slot;
            a.Qnew = (t < depth) ? Qs + (size_t)t * slot : nullptr;
            a.S = Ss + (size_t)(t - 1) * slot; a.G = Gs + (size_t)(t - 1) * slot;
            a.Z = Zs + (size_t)(t - 1) * slot; a.M = Ms + (size_t)(t - 1) * slot;
        } else {
            a.Hprev = Hs + (size_t)((t - 1) & 1) * slot; a.Hnew = Hs + (size_t)(t & 1) * slot;
            a.Qprev = Qs + (size_t)((t - 1) & 1) * slot; a.Qnew = Qs + (size_t)(t & 1) * slot;
            a.S = a.G = a.Z = a.M = nullptr;
        }
        a.write_q = (t < depth);
        { const char* e = getenv("GGPM_ABLATE"); a.ablate = e ? atoi(e) : 0; }
        ggpm_timing_begin(0, s, flops);
#define CALL(T) launch_fwd<T>(a, save_for_backward != 0, lds_bytes, grid, s)
        GGPM_DISPATCH_TPW(tpw, CALL)
#undef CALL
        ggpm_timing_end(0, s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" size_t ggpm_gru_backward_workspace_bytes(int E1, int H, int depth) {
    const size_t Hp = (size_t)ggpm_padded_hidden(H);
    const size_t slot = (size_t)E1 * Hp;
    size_t f = 0;
    f += 2 * (size_t)depth * slot;                 // DMP, DZP
    f += (size_t)(depth > 1 ? depth - 1 : 1) * slot;   // DQ
    f += 4 * slot;                                 // dS/dG double buffers
    f += 3 * Hp * Hp;                              // packed transposes
    f += 64 * Hp;                                  // colsum scratch
    size_t bytes = f * sizeof(float);
    bytes += ggpm_gemm_workspace_bytes(H, H, depth * E1);   // split-K slabs (largest contraction)
    return bytes + 256;
}

extern "C" int ggpm_gru_backward(int E1, int H, int depth, const float* Xr, const float* Wz_h, int ld_wz,
                                 const float* Ur, int ld_ur, const float* Wh_h, int ld_wh,
                                 const int32_t* pred_rowptr, const int32_t* pred_col,
                                 const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                                 const float* Qs, const float* Ss, const float* Gs, const float* Zs,
                                 const float* Ms, const float* dHD, float* dXz, float* dXr, float* dXh,
                                 float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu, float* dWh_h,
                                 int ld_dwh, float* work, size_t work_bytes, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xr || !Wz_h || !Ur || !Wh_h || !pred_rowptr || !pred_col ||
        !succ_rowptr || !succ_col || !Hs || !Qs || !Ss || !Gs || !Zs || !Ms || !dHD || !dXz || !dXr || !dXh ||
        !dWz_h || !dUr || !dbu || !dWh_h || !work)
        return GGPM_ERR_ARG;
    if (work_bytes < ggpm_gru_backward_workspace_bytes(E1, H, depth)) return GGPM_ERR_WORKSPACE;
    const int Hp = ggpm_padded_hidden(H);
    const int NT = Hp / 16, tpw = ggpm_ceil_div(NT, NWAVES);
    const size_t lds_bytes = (size_t)3 * ROWS * (Hp + 4) * sizeof(float);
    if (tpw > 3 || lds_bytes > 160 * 1024) return GGPM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp, slot = (size_t)E1 * Hp;

    float* w = work;
    float* DMP = w; w += (size_t)depth * slot;
    float* DZP = w; w += (size_t)depth * slot;
    float* DQ = w; w += (size_t)(depth > 1 ? depth - 1 : 1) * slot;
    float* dSb[2]; float* dGb[2];
    dSb[0] = w; w += slot; dSb[1] = w; w += slot; dGb[0] = w; w += slot; dGb[1] = w; w += slot;
    float* pWzT = w; w += HH; float* pWhT = w; w += HH; float* pUrT = w; w += HH;
    float* csws = w; w += (size_t)64 * Hp;
    float* skws = w;
    const size_t skbytes = work_bytes - (size_t)((char*)skws - (char*)work);

    ggpm_launch_pack(Wz_h, ld_wz, H, Hp, 1, pWzT, s);
    ggpm_launch_pack(Wh_h, ld_wh, H, Hp, 1, pWhT, s);
    ggpm_launch_pack(Ur, ld_ur, H, Hp, 1, pUrT, s);
    (void)hipMemsetAsync(dXz, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(dXr, 0, slot * sizeof(float), s);
    (void)hipMemsetAsync(dXh, 0, slot * sizeof(float), s);

    const int grid = ggpm_ceil_div(E1, ROWS);
    const double flops = 2.0 * 3.0 * (double)(E1 - 1) * H * H;   // algorithmic: 3 gate products per message
    for (int t = depth; t >= 1; --t) {
        GruBwdArgs a;
        a.E1 = E1; a.Hp = Hp; a.first = (t == depth); a.last = (t == 1);
        a.Xr = Xr;
        a.Hcur = Hs + (size_t)t * slot;
        a.Qcur = (t < depth) ? Qs + (size_t)t * slot : nullptr;
        a.Hprv = Hs + (size_t)(t - 1) * slot; a.Qprv = Qs + (size_t)(t - 1) * slot;
        a.S = Ss + (size_t)(t - 1) * slot; a.G = Gs + (size_t)(t - 1) * slot;
        a.Z = Zs + (size_t)(t - 1) * slot; a.M = Ms + (size_t)(t - 1) * slot;
        a.dHD = dHD;
        a.dSin = dSb[(t + 1) & 1]; a.dGin = dGb[(t + 1) & 1];
        a.dSout = dSb[t & 1]; a.dGout = dGb[t & 1];
        a.DQ = (t < depth) ? DQ + (size_t)(t - 1) * slot : nullptr;
        a.DMP = DMP + (size_t)(t - 1) * slot; a.DZP = DZP + (size_t)(t - 1) * slot;
        a.dXz = dXz; a.dXr = dXr; a.dXh = dXh;
        a.WzT = pWzT; a.WhT = pWhT; a.UrT = pUrT;
        a.rowptr = pred_rowptr; a.col = pred_col; a.srowptr = succ_rowptr; a.scol = succ_col;
        ggpm_timing_begin(1, s, flops);
#define CALL(T) launch_bwd<T>(a, lds_bytes, grid, s)
        GGPM_DISPATCH_TPW(tpw, CALL)
#undef CALL
        ggpm_timing_end(1, s);
    }
    GGPM_CHECK_LAUNCH();

    // weight gradients: tall contractions over every (depth, message) row of the stashes
    const int KD = depth * E1;
    int rc;
    rc = ggpm_gemm(1, 0, H, H, KD, DMP, Hp, Gs, Hp, dWh_h, ld_dwh, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
    if (rc) return rc;
    rc = ggpm_gemm(1, 0, H, H, KD, DZP, Hp, Ss, Hp, dWz_h, ld_dwz, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
    if (rc) return rc;
    if (depth > 1) {
        const int KQ = (depth - 1) * E1;
        rc = ggpm_gemm(1, 0, H, H, KQ, DQ, Hp, Hs + slot, Hp, dUr, ld_dur, H, nullptr, 0, GGPM_ACT_NONE, 0, skws, skbytes, stream);
        if (rc) return rc;
        rc = ggpm_colsum(DQ, Hp, KQ, H, dbu, csws, stream);
        if (rc) return rc;
    } else {
        for (int r = 0; r < H; ++r) (void)hipMemsetAsync(dUr + (size_t)r * ld_dur, 0, H * sizeof(float), s);
        (void)hipMemsetAsync(dbu, 0, H * sizeof(float), s);
    }
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
