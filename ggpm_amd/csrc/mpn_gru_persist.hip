// Persistent GRU depth loop for gfx950: ALL `depth` message steps of a level in ONE launch.
//
// Same arithmetic and the same stashes as mpn_gru.hip (reference ggpm/rnn.py:25-50), different execution:
// the level's message rows are cut into molecule clusters (cluster.h); a cluster is owned by `ncg` workgroups,
// one per 64-feature column group, which keep their slice of the three gate matrices IN REGISTERS for the
// whole launch (12 waves: 4 output tiles x {W_z, W_h, U_r}, one 16 x Hp weight fragment = KC float4 per lane)
// and hand activations to each other through L2 with write-through stores and a per-cluster arrival counter.
// Per depth and workgroup:
//   P1  gather its 64 columns of s = sum h_p, g = sum r_p h_p over the cluster's rows (h, q of its own columns
//       come back from its own previous step)                                  -> exchange + stash
//   --  cluster sync
//   P2  per 16-row tile: full s, g rows -> LDS, gate GEMMs on MFMA from the register-resident fragments,
//       gate math -> h' (its 64 columns)                                       -> exchange + stash
//   --  cluster sync
//   P3  per 16-row tile: full h' rows -> LDS, q' = U_r h' + b_u (its 64 columns), kept for the next P1.
// No kernel boundary, no weight re-streaming and no full-width redundant gathers inside the depth loop.
#include "tile_mma.h"
#include "cluster.h"
#include <cstdlib>
#include <cstdio>

namespace {

constexpr int PW = 12;              // waves per workgroup
constexpr int TPW = 4;              // output tiles (of 16 features) per workgroup
constexpr int CGW = TPW * 16;       // features per column group

struct GruPFwdArgs {
    int E1, Hp, depth, ncg, P;      // P: exchange row pitch in floats (= Hp + 4, the LDS tile pitch)
    const float *Xz, *Xr, *Xh;      // [E1][Hp]
    float *Hs, *Qs;                 // [depth+1][E1][Hp]
    float *Ss, *Gs, *Zs, *Ms, *Rs;  // [depth][E1][Hp]
    const float *Wz, *Wh, *Ur, *bu; // packed fragments / padded bias
    const int32_t *rowptr, *col;    // predecessors
    const int32_t* clusters;
    float *XS, *XG, *XH;            // exchange buffers [E1 + 16][P]
    unsigned* sync;
    unsigned long long* dbg;        // optional timeline (GGPM_PDEBUG): wall_clock64 stamps of workgroup 0, depth 5
    int ablate;                     // timing experiments only (GGPM_PABLATE): 1 no P1, 2 no cluster syncs, 4 no P2, 8 no P3
};

__device__ __forceinline__ f32x4 v4(float4 a) { return f32x4{a.x, a.y, a.z, a.w}; }
// sigmoid on the hardware exp2 / rcp units (about 2 ulp; the gather phase is VALU bound with the libm forms)
__device__ __forceinline__ float fsig(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float4 fsig4(float4 a) { return make_float4(fsig(a.x), fsig(a.y), fsig(a.z), fsig(a.w)); }

constexpr int RCAP = 256;          // cluster rows whose predecessor lists are cached in LDS (<= 4 entries each)

// LDS-DMA copy of the 16 full rows [r0, r0+16) of one exchange array into an LDS tile.  The exchange arrays use
// the SAME row pitch as the LDS tiles (LD = 16 KC + 4 floats: the bank-conflict padding), so a tile is one
// contiguous byte range and is moved by ceil(64 LD / 1024) full-width wave instructions of 1 KiB each (the LDS
// destination of a global_load_lds is wave-uniform base + lane*16), L1 bypassed (aux 16 = sc1: the rows were
// written by other workgroups in this launch).  No VGPRs are spent on the data; the issuing wave drains vmcnt
// before the workgroup barrier that publishes the tile.  Rows past the cluster end are copied as they are (the
// arrays carry 16 rows of slack): an MFMA output row only depends on its own operand row and those outputs are
// never stored.
template <int KC>
__device__ __forceinline__ void glds_tile(const float* src, float* tile, int r0, int wv, int nwv, int lane) {
    constexpr int LD = KC * 16 + 4, BYTES = 16 * LD * 4, NCH = (BYTES + 1023) / 1024;
    const char* g = reinterpret_cast<const char*>(src + (size_t)r0 * LD) + lane * 16;
    char* l = reinterpret_cast<char*>(tile);
#pragma unroll 1
    for (int i = wv; i < NCH; i += nwv) {
        if (i * 1024 + lane * 16 < BYTES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024),
                                             (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 16);
    }
}

// acc += W(fragment in registers) x tile^T over all KC chunks, LDS reads two chunks ahead of the MFMAs
template <int KC>
__device__ __forceinline__ f32x4 mfma_tile(const f32x4 (&wreg)[KC], const float* tb) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b0 = *reinterpret_cast<const f32x4*>(tb);
    f32x4 b1 = *reinterpret_cast<const f32x4*>(tb + (KC > 1 ? 16 : 0));
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const f32x4 b = b0;
        b0 = b1;
        if (kc + 2 < KC) b1 = *reinterpret_cast<const f32x4*>(tb + (kc + 2) * 16);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[kc][s4], b[s4], acc, 0, 0, 0);
    }
    return acc;
}

#define STAMP(id)                                                                         \
    do {                                                                                  \
        if (a.dbg && w == 0 && d == 5 && lane == 0 && (wave == 0 || wave == 8 || wave == 10) && nst < 40) \
            a.dbg[(wave == 0 ? 0 : (wave == 8 ? 1 : 2)) * 64 + nst++] = ((unsigned long long)(id) << 48) | (wall_clock64() & 0xffffffffffffull); \
    } while (0)

template <int KC>
__global__ void __launch_bounds__(PW * 64) gru_fwd_persist(GruPFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LD = KC * 16 + 4, TILE = 16 * LD;
    float* Tbuf = lds;                           // 6 tiles: 3 buffers of [s, g]
    f32x4* Ex = reinterpret_cast<f32x4*>(lds + 6 * TILE);                       // [2 buffers][z, s, m][TPW][64]
    int* PT = reinterpret_cast<int*>(lds + 6 * TILE + 2 * 3 * TPW * 64 * 4);     // [RCAP][4] predecessor ids
    int* flags = PT + RCAP * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = a.Hp, NT = Hp / 16;
    constexpr int P = KC * 16 + 4;           // exchange row pitch = LDS tile pitch

    // ---- logical workgroup id by ticket (start order), cluster-major
    if (tid == 0) {
        flags[1] = (int)__hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flags[2] = 1;                            // all predecessor lists fit the LDS table
    }
    __syncthreads();
    int w = flags[1];
    const int nclu = a.clusters[0];
    int cl = w / a.ncg, cg = w - cl * a.ncg;
    if (a.ablate & 512) {          // experiment: blockIdx-based placement, cluster members share blockIdx % 8
        w = blockIdx.x;
        const int xcd = w & 7, idx = w >> 3;
        cl = xcd + 8 * (idx / a.ncg);
        cg = idx % a.ncg;
    }
    if (cl >= nclu) return;
    const int lo = a.clusters[1 + cl], hi = a.clusters[2 + cl];
    unsigned* cnt = a.sync + GGPM_SYNC_HDR + cl;
    unsigned* tmo = a.sync + 1;

    // ---- this wave's register-resident weight fragment
    const int role = wave >> 2, j = wave & 3;         // role 0: W_z, 1: W_h, 2: U_r
    const int t = cg * TPW + j;
    const bool tile_on = t < NT;
    f32x4 wreg[KC];
    {
        const float* wsrc = role == 0 ? a.Wz : (role == 1 ? a.Wh : a.Ur);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            wreg[kc] = tile_on ? *reinterpret_cast<const f32x4*>(wsrc + ggpm_pack_index(t, kc, KC, lane))
                               : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned xbytes = (unsigned)((size_t)(a.E1 + 16) * P * sizeof(float));
    const __amdgpu_buffer_rsrc_t rS = ggpm_rsrc(a.XS, xbytes), rG = ggpm_rsrc(a.XG, xbytes),
                                 rH = ggpm_rsrc(a.XH, xbytes);
    const size_t slot = (size_t)a.E1 * Hp;

    // gather geometry: 16 lanes cover the 64 columns of one row, 4 rows per wave instruction
    const int gcol = cg * CGW + (lane & 15) * 4;
    const bool gcol_on = gcol < Hp;
    // MFMA geometry: lane holds features 16t + 4(lane>>4) + 0..3 of row (lane & 15)
    const int mrow = lane & 15, mcol = 16 * t + 4 * (lane >> 4);
    const int boff = mrow * LD + 4 * (lane >> 4);

    // ---- predecessor table (the graph does not change over the depth loop) + state of depth 0 (h = 0, q = b_u)
    if (hi - lo > RCAP && tid == 0) flags[2] = 0;
    for (int r = tid; r < hi - lo && r < RCAP; r += PW * 64) {
        const int e0 = a.rowptr[lo + r], n = a.rowptr[lo + r + 1] - e0;
        if (n > 4) flags[2] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) PT[r * 4 + k] = k < n ? a.col[e0 + k] : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool fast_lists = flags[2] != 0;

    const int n_rt = (hi - lo + 15) >> 4;
    for (int d = 0; d < a.depth; ++d) {
        int nst = 0;
        STAMP(1);
        float* S = a.Ss + (size_t)d * slot;
        float* G = a.Gs + (size_t)d * slot;
        float* Z = a.Zs + (size_t)d * slot;
        float* M = a.Ms + (size_t)d * slot;
        float* R = a.Rs + (size_t)d * slot;
        float* Hn = a.Hs + (size_t)(d + 1) * slot;
        float* Qn = a.Qs + (size_t)(d + 1) * slot;
        // own-column h, q of depth d: written by THIS workgroup (slot 0: by the init launch) with plain stores,
        // so they sit in this XCD's L2; read back with L1-bypassing loads
        const __amdgpu_buffer_rsrc_t rHd = ggpm_rsrc(a.Hs + (size_t)d * slot, (unsigned)(slot * sizeof(float)));
        const __amdgpu_buffer_rsrc_t rQd = ggpm_rsrc(a.Qs + (size_t)d * slot, (unsigned)(slot * sizeof(float)));

        // ---------------- P1: column-sliced gather
        for (int base = lo; base < hi && !(a.ablate & 1); base += PW * 4) {
            const int row = base + wave * 4 + (lane >> 4);
            const bool rv = row < hi;
            float4 xr = ggpm_zero4();
            if (rv && gcol_on) xr = ggpm_ld4(a.Xr + (size_t)row * Hp + gcol);
            float4 s = ggpm_zero4(), g = ggpm_zero4(), rc = ggpm_zero4();
            if (fast_lists) {
                int pr[4] = {0, 0, 0, 0};
                if (rv) {
                    const int4 pv = *reinterpret_cast<const int4*>(PT + (row - lo) * 4);
                    pr[0] = pv.x; pr[1] = pv.y; pr[2] = pv.z; pr[3] = pv.w;
                }
                f32x4 h[4], q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {                 // empty slots (id 0) must not touch row 0: it belongs to
                    const unsigned o = (gcol_on && pr[k])     // cluster 0; they read past the buffer end instead
                                           ? (unsigned)(((size_t)pr[k] * Hp + gcol) * sizeof(float)) : 0xffffff00u;
                    h[k] = ggpm_xld(rHd, o);                  // (past the end of the buffer: reads 0)
                    q[k] = ggpm_xld(rQd, o);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (pr[k] == 0) continue;
                    const float4 r = fsig4(xr + ggpm_f4(q[k]));
                    const float4 rh = r * ggpm_f4(h[k]);
                    s = s + ggpm_f4(h[k]);
                    g = g + rh;
                    rc = rc + rh * make_float4(1.f - r.x, 1.f - r.y, 1.f - r.z, 1.f - r.w);
                }
            } else {
                int e0 = 0, e1 = 0;
                if (rv) { e0 = a.rowptr[row]; e1 = a.rowptr[row + 1]; }
                for (int e = e0; e < e1; e += 2) {
                    const int p0 = a.col[e];
                    const int p1 = (e + 1 < e1) ? a.col[e + 1] : 0;
                    const unsigned o0 = gcol_on ? (unsigned)(((size_t)p0 * Hp + gcol) * sizeof(float)) : 0xffffff00u;
                    const unsigned o1 = (gcol_on && p1) ? (unsigned)(((size_t)p1 * Hp + gcol) * sizeof(float)) : 0xffffff00u;
                    const f32x4 h0 = ggpm_xld(rHd, o0), q0 = ggpm_xld(rQd, o0);
                    const f32x4 h1 = ggpm_xld(rHd, o1), q1 = ggpm_xld(rQd, o1);
                    const float4 r0 = fsig4(xr + ggpm_f4(q0)), r1 = fsig4(xr + ggpm_f4(q1));
                    const float4 rh0 = r0 * ggpm_f4(h0), rh1 = r1 * ggpm_f4(h1);
                    s = s + ggpm_f4(h0) + ggpm_f4(h1);
                    g = g + rh0 + rh1;
                    rc = rc + rh0 * make_float4(1.f - r0.x, 1.f - r0.y, 1.f - r0.z, 1.f - r0.w) +
                         rh1 * make_float4(1.f - r1.x, 1.f - r1.y, 1.f - r1.z, 1.f - r1.w);
                }
            }
            if (rv && gcol_on) {
                const unsigned off = (unsigned)(((size_t)row * P + gcol) * sizeof(float));
                if (a.ablate & 256) { ggpm_xst_l2(rS, off, v4(s)); ggpm_xst_l2(rG, off, v4(g)); }
                else { ggpm_xst(rS, off, v4(s)); ggpm_xst(rG, off, v4(g)); }
                {
                    const size_t o = (size_t)row * Hp + gcol;
                    ggpm_st4(S + o, s);
                    ggpm_st4(G + o, g);
                    ggpm_st4(R + o, rc);
                }
            }
        }
        STAMP(2);
        if (!(a.ablate & 2) && !ggpm_cluster_sync(cnt, (unsigned)((2 * d + 1) * a.ncg), tmo, flags)) return;
        STAMP(3);

        // ---------------- P2: gate GEMMs + gate math per 16-row tile, software pipelined over the tiles
        // (three LDS tile buffers).  Iteration rt:
        //   gate waves   MFMA on tile rt -> z, s / m into the LDS exchange buffer rt & 1
        //   U_r waves    drain what they issued one iteration ago (DMA of tile rt+1, stores of tile rt-2), gate
        //                math + stores of tile rt-1, LDS-DMA of tile rt+2  -- so nobody waits for a store or a
        //                DMA that was issued in the same iteration
        const int n_rt2 = (a.ablate & 4) ? 0 : n_rt;
        const bool dma_wave = wave == 8 || wave == 9;     // the other two U_r waves (10, 11) store the results: a
        float4 x_cur = ggpm_zero4();                      // wave that waits for its DMA never waits for a store
        if (n_rt2) {
            // the first two tiles are fetched by ALL waves (a single wave pulls fresh hand-off data at ~13 GB/s)
            glds_tile<KC>(a.XS, Tbuf, lo, wave, PW, lane);
            glds_tile<KC>(a.XG, Tbuf + TILE, lo, wave, PW, lane);
            if (role < 2 && tile_on && lo + mrow < hi)
                x_cur = ggpm_ld4((role == 0 ? a.Xz : a.Xh) + (size_t)(lo + mrow) * Hp + mcol);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (n_rt2 > 1 && dma_wave) {
                glds_tile<KC>(a.XS, Tbuf + 2 * TILE, lo + 16, wave - 8, 2, lane);
                glds_tile<KC>(a.XG, Tbuf + 3 * TILE, lo + 16, wave - 8, 2, lane);
            }
        }
        for (int rt = 0; rt <= n_rt2 && n_rt2; ++rt) {
            const int r0 = lo + rt * 16;
            STAMP(10 + rt);
            ggpm_lds_barrier();       // tile rt has landed, exchange buffer (rt-1) & 1 is complete
            STAMP(20 + rt);
            if (dma_wave) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // DMA of tile rt+1 (issued one iteration ago)
                if (rt + 2 < n_rt2 && !(a.ablate & 32)) {
                    float* Tn = Tbuf + ((rt + 2) % 3) * 2 * TILE;
                    glds_tile<KC>(a.XS, Tn, r0 + 32, wave - 8, 2, lane);
                    glds_tile<KC>(a.XG, Tn + TILE, r0 + 32, wave - 8, 2, lane);
                }
            } else if (role == 2) {
                if (rt >= 1 && !(a.ablate & 64)) {
                    const int row = r0 - 16 + mrow;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int jj = (wave - 10) * 2 + u, tt = cg * TPW + jj;
                        const int col = 16 * tt + 4 * (lane >> 4);
                        const bool on = tt < NT && row < hi;
                        const f32x4* E = Ex + ((rt - 1) & 1) * 3 * TPW * 64 + jj * 64 + lane;
                        f32x4 zv = E[0], sv = E[TPW * 64], m = E[2 * TPW * 64];
                        f32x4 h = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (on && row != 0) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) h[k] = (1.f - zv[k]) * sv[k] + zv[k] * m[k];
                        } else {
                            zv = f32x4{0.f, 0.f, 0.f, 0.f};
                            m = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        if (on) {
                            if (a.ablate & 256) ggpm_xst_l2(rH, (unsigned)(((size_t)row * P + col) * sizeof(float)), h);
                            else ggpm_xst(rH, (unsigned)(((size_t)row * P + col) * sizeof(float)), h);
                            {
                                const size_t o = (size_t)row * Hp + col;
                                *reinterpret_cast<f32x4*>(Hn + o) = h;
                                *reinterpret_cast<f32x4*>(Z + o) = zv;
                                *reinterpret_cast<f32x4*>(M + o) = m;
                            }
                        }
                    }
                }
            } else if (rt < n_rt2) {
                const float* Ts = Tbuf + (rt % 3) * 2 * TILE;
                const float* tile = role == 0 ? Ts : Ts + TILE;
                float4 x_next = ggpm_zero4();
                if (rt + 1 < n_rt2 && tile_on && r0 + 16 + mrow < hi)
                    x_next = ggpm_ld4((role == 0 ? a.Xz : a.Xh) + (size_t)(r0 + 16 + mrow) * Hp + mcol);
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(a.ablate & 16)) acc = mfma_tile<KC>(wreg, tile + boff);
                const float4 pre = ggpm_f4(acc) + x_cur;
                f32x4* E = Ex + (rt & 1) * 3 * TPW * 64 + j * 64 + lane;
                if (a.ablate & 128) {
                    E[role == 0 ? 0 : 2 * TPW * 64] = v4(pre);
                } else if (role == 0) {
                    E[0] = v4(fsig4(pre));
                    E[TPW * 64] = *reinterpret_cast<const f32x4*>(Ts + mrow * LD + mcol);
                } else {
                    E[2 * TPW * 64] = f32x4{tanhf(pre.x), tanhf(pre.y), tanhf(pre.z), tanhf(pre.w)};
                }
                x_cur = x_next;
            }
        }
        STAMP(4);
        if (!(a.ablate & 2) && !ggpm_cluster_sync(cnt, (unsigned)((2 * d + 2) * a.ncg), tmo, flags)) return;
        STAMP(5);

        // ---------------- P3: q' = U_r h' + b_u for this column group (nobody reads q of the last depth); the
        // gate waves (idle here) stream the h' rows two tiles ahead
        if (d + 1 < a.depth && !(a.ablate & 8)) {
            glds_tile<KC>(a.XH, Tbuf, lo, wave, PW, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (role != 2 && n_rt > 1) glds_tile<KC>(a.XH, Tbuf + 2 * TILE, lo + 16, wave, 8, lane);
            const float4 bq = (role == 2 && tile_on) ? ggpm_ld4(a.bu + mcol) : ggpm_zero4();
            for (int rt = 0; rt < n_rt; ++rt) {
                const int r0 = lo + rt * 16;
                const float* Th = Tbuf + (rt % 3) * 2 * TILE;
                STAMP(30 + rt);
                ggpm_lds_barrier();
                if (role != 2) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile rt+1, issued one iteration ago
                    if (rt + 2 < n_rt) glds_tile<KC>(a.XH, Tbuf + ((rt + 2) % 3) * 2 * TILE, r0 + 32, wave, 8, lane);
                } else {
                    const int row = r0 + mrow;
                    const f32x4 acc = mfma_tile<KC>(wreg, Th + boff);
                    if (tile_on && row < hi) *reinterpret_cast<f32x4*>(Qn + (size_t)row * Hp + mcol) = acc + v4(bq);
                }
            }
        }
        STAMP(6);
        // own h', q' columns must have landed before this workgroup's next gather reads them back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <typename K>
inline void set_lds_p(K kernel, size_t bytes) {
    static size_t have = 0;
    if (bytes > have) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
        have = bytes;
    }
}

template <int KC>
int launch_fwd_persist(const GruPFwdArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)6 * 16 * (KC * 16 + 4) + 2 * 3 * TPW * 64 * 4 + RCAP * 4 + 16) * sizeof(float);
    set_lds_p(gru_fwd_persist<KC>, lds);
    gru_fwd_persist<KC><<<grid, PW * 64, lds, s>>>(a);
    return 0;
}

__global__ void gru_p_init_slot0(float* __restrict__ H0, float* __restrict__ Q0, const float* __restrict__ bu, int Hp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= Hp) return;
    H0[(size_t)r * Hp + c] = 0.f;
    Q0[(size_t)r * Hp + c] = bu[c];
}

}  // namespace

extern "C" int ggpm_gru_persistent_supported(int H) {
    const int KC = ggpm_padded_hidden(H) / 16;
    return KC == 1 || KC == 2 || KC == 7 || KC == 16 || KC == 19;
}

extern "C" int ggpm_gru_persistent_ncg(int H) { return ggpm_ceil_div(ggpm_padded_hidden(H) / 16, TPW); }

// Rows per cluster the builder should aim for so that clusters x column groups fills (and does not exceed) the chip.
extern "C" int ggpm_gru_persistent_target_rows(int E1, int H) {
    const int ncg = ggpm_gru_persistent_ncg(H);
    int nclu = 248 / ncg;
    if (nclu < 1) nclu = 1;
    int target = ggpm_ceil_div(E1, nclu);
    return target < 8 ? 8 : target;
}

// floats of exchange workspace / uint32 of sync words a launch needs
extern "C" size_t ggpm_gru_persistent_workspace_floats(int E1, int H) {
    return (size_t)3 * (E1 + 16) * (ggpm_padded_hidden(H) + 4);
}

extern "C" int ggpm_gru_forward_persistent(int E1, int H, int depth, const float* Xz, const float* Xr, const float* Xh,
                                           const float* Wz_h, int ld_wz, const float* Ur, int ld_ur, const float* bu,
                                           const float* Wh_h, int ld_wh, const int32_t* pred_rowptr,
                                           const int32_t* pred_col, const int32_t* clusters, int target_rows,
                                           float* Hs, float* Qs, float* Ss, float* Gs, float* Zs, float* Ms, float* Rs,
                                           float* wpack, float* xwork, uint32_t* sync, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (E1 <= 0 || H <= 0 || depth <= 0 || !Xz || !Xr || !Xh || !Wz_h || !Ur || !bu || !Wh_h || !pred_rowptr ||
        !pred_col || !clusters || target_rows <= 0 || !Hs || !Qs || !Ss || !Gs || !Zs || !Ms || !Rs || !wpack ||
        !xwork || !sync)
        return GGPM_ERR_ARG;
    if (!ggpm_gru_persistent_supported(H)) return GGPM_ERR_UNSUPPORTED;
    const int Hp = ggpm_padded_hidden(H), KC = Hp / 16;
    hipStream_t s = (hipStream_t)stream;
    const size_t HH = (size_t)Hp * Hp;
    float* pWz = wpack; float* pWh = wpack + HH; float* pUr = wpack + 2 * HH; float* pbu = wpack + 3 * HH;
    {
        GgpmPackArgs pk = {};
        pk.W[0] = Wz_h; pk.ldw[0] = ld_wz; pk.W[1] = Wh_h; pk.ldw[1] = ld_wh; pk.W[2] = Ur; pk.ldw[2] = ld_ur;
        pk.H = H; pk.Hp = Hp; pk.transpose = 0; pk.dst = wpack; pk.bias = bu; pk.bias_out = pbu;
        ggpm_launch_pack(pk, 3, s);
    }
    dim3 ig(ggpm_ceil_div(Hp, 256), E1);
    gru_p_init_slot0<<<ig, 256, 0, s>>>(Hs, Qs, pbu, Hp);

    GruPFwdArgs a = {};
    a.E1 = E1; a.Hp = Hp; a.depth = depth; a.ncg = ggpm_gru_persistent_ncg(H); a.P = Hp + 4;
    a.Xz = Xz; a.Xr = Xr; a.Xh = Xh; a.Hs = Hs; a.Qs = Qs; a.Ss = Ss; a.Gs = Gs; a.Zs = Zs; a.Ms = Ms; a.Rs = Rs;
    a.Wz = pWz; a.Wh = pWh; a.Ur = pUr; a.bu = pbu; a.rowptr = pred_rowptr; a.col = pred_col; a.clusters = clusters;
    const size_t xs = (size_t)(E1 + 16) * a.P;
    a.XS = xwork; a.XG = xwork + xs; a.XH = xwork + 2 * xs;
    a.sync = sync;
    if (const char* e = getenv("GGPM_PABLATE")) a.ablate = atoi(e);
    static unsigned long long* dbg_buf = nullptr;
    if (getenv("GGPM_PDEBUG")) {
        if (!dbg_buf) (void)hipMalloc(&dbg_buf, 3 * 64 * 8);
        (void)hipMemsetAsync(dbg_buf, 0, 3 * 64 * 8, s);
        a.dbg = dbg_buf;
    }
    const int max_clusters = ggpm_ceil_div(E1, target_rows) + 1;
    const size_t sync_bytes = (size_t)ggpm_round_up(GGPM_SYNC_HDR + max_clusters, 4) * sizeof(uint32_t);
    if (hipMemsetAsync(sync, 0, sync_bytes, s) != hipSuccess) return GGPM_ERR_LAUNCH;
    int grid = max_clusters * a.ncg;
    if (a.ablate & 512) grid = 8 * ggpm_ceil_div(max_clusters, 8) * a.ncg;
    const double flops1 = 2.0 * (double)(E1 - 1) * H * H;
    ggpm_timing_begin(0, s, 3 * flops1 * depth);
    switch (KC) {
        case 1: launch_fwd_persist<1>(a, grid, s); break;
        case 2: launch_fwd_persist<2>(a, grid, s); break;
        case 7: launch_fwd_persist<7>(a, grid, s); break;
        case 16: launch_fwd_persist<16>(a, grid, s); break;
        case 19: launch_fwd_persist<19>(a, grid, s); break;
        default: return GGPM_ERR_UNSUPPORTED;
    }
    ggpm_timing_end(0, s);
    GGPM_CHECK_LAUNCH();
    if (a.dbg) {
        unsigned long long host[3 * 64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host, a.dbg, sizeof(host), hipMemcpyDeviceToHost);
        for (int wv = 0; wv < 3; ++wv) {
            fprintf(stderr, "[pdebug E1=%d wave %d]", E1, wv == 0 ? 0 : (wv == 1 ? 8 : 10));
            unsigned long long t0 = host[0] & 0xffffffffffffull;
            for (int i = 0; i < 40 && host[wv * 64 + i]; ++i)
                fprintf(stderr, " %d:%.2f", (int)(host[wv * 64 + i] >> 48), ((host[wv * 64 + i] & 0xffffffffffffull) - t0) * 0.01);
            fprintf(stderr, "\n");
        }
    }
    return GGPM_OK;
}

// 0 = no persistent launch since the last call hit its spin limit; reads (and clears) the timeout word: host sync.
extern "C" int ggpm_persistent_timeout(uint32_t* sync, ggpm_stream_t stream) {
    uint32_t v = 0;
    if (hipMemcpyAsync(&v, sync + 1, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}
