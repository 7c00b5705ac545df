// Cluster builder: cut a level's message rows into closed ranges (see cluster.h) -- one single-workgroup launch.
#include "cluster.h"

namespace {

constexpr int BT = 1024;

// scratch: A = [0,E1) , B = [E1,2E1) , V = [2E1,3E1)
__global__ void __launch_bounds__(BT) build_clusters_k(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int E1, int target,
                                                       int32_t* __restrict__ table, int32_t* __restrict__ scratch) {
    __shared__ int part[BT];
    int32_t* A = scratch;            // hi -> inclusive prefix max
    int32_t* B = scratch + E1;       // lo -> inclusive suffix min
    int32_t* V = scratch + 2 * E1;   // valid cut flags
    const int tid = threadIdx.x;
    for (int e = tid; e < E1; e += BT) {
        int lo = e, hi = e;
        for (int j = rowptr[e]; j < rowptr[e + 1]; ++j) {
            const int p = col[j];
            lo = min(lo, p);
            hi = max(hi, p);
        }
        A[e] = hi;
        B[e] = lo;
    }
    __syncthreads();
    const int chunk = (E1 + BT - 1) / BT;
    const int c0 = tid * chunk, c1 = min(E1, c0 + chunk);
    // ---- prefix max of A
    int run = -1;
    for (int e = c0; e < c1; ++e) run = max(run, A[e]);
    part[tid] = run;
    __syncthreads();
    for (int d = 1; d < BT; d <<= 1) {
        const int v = tid >= d ? part[tid - d] : -1;
        __syncthreads();
        part[tid] = max(part[tid], v);
        __syncthreads();
    }
    run = tid > 0 ? part[tid - 1] : -1;
    __syncthreads();
    for (int e = c0; e < c1; ++e) {
        run = max(run, A[e]);
        A[e] = run;
    }
    // ---- suffix min of B
    int rmin = 0x7fffffff;
    for (int e = c0; e < c1; ++e) rmin = min(rmin, B[e]);
    part[tid] = rmin;
    __syncthreads();
    for (int d = 1; d < BT; d <<= 1) {
        const int v = tid + d < BT ? part[tid + d] : 0x7fffffff;
        __syncthreads();
        part[tid] = min(part[tid], v);
        __syncthreads();
    }
    rmin = tid + 1 < BT ? part[tid + 1] : 0x7fffffff;
    for (int e = c1 - 1; e >= c0; --e) {
        rmin = min(rmin, B[e]);
        B[e] = rmin;
    }
    __syncthreads();
    // ---- a cut in front of row b is valid when nothing below b reaches b or above and nothing from b on reaches below
    for (int b = tid; b < E1; b += BT) V[b] = (b >= 1 && A[b - 1] < b && B[b] >= b) ? 1 : 0;
    __syncthreads();
    // ---- greedy: first valid cut at least `target` rows after the previous one (one wave, ballot search)
    if (tid < 64) {
        int n = 0, pos = target;
        if (tid == 0) table[1] = 0;
        while (pos < E1) {
            const int b = pos + tid;
            const unsigned long long m = __ballot(b < E1 && V[b] != 0);
            if (m) {
                const int cut = pos + __ffsll((long long)m) - 1;
                ++n;
                if (tid == 0) table[1 + n] = cut;
                pos = cut + target;
            } else {
                pos += 64;
            }
        }
        ++n;
        if (tid == 0) {
            table[1 + n] = E1;
            table[0] = n;
        }
    }
}

}  // namespace

int ggpm_build_clusters_impl(const int32_t* rowptr, const int32_t* col, int E1, int target, int32_t* table,
                             int32_t* scratch, hipStream_t s) {
    if (E1 <= 0 || target <= 0 || !rowptr || !col || !table || !scratch) return GGPM_ERR_ARG;
    build_clusters_k<<<1, BT, 0, s>>>(rowptr, col, E1, target, table, scratch);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// table: int32[E1 + 4]; scratch: int32[3 * E1]
extern "C" int ggpm_build_clusters(const int32_t* pred_rowptr, const int32_t* pred_col, int E1, int target_rows,
                                   int32_t* table, int32_t* scratch, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    return ggpm_build_clusters_impl(pred_rowptr, pred_col, E1, target_rows, table, scratch, (hipStream_t)stream);
}
