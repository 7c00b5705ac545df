// Graph layout kernels: MolGraph.tensorize()'s zero-padded int64 neighbour tables -> int32 CSR,
// and the CSR transposes that make every backward scatter a deterministic gather.
// Integer/byte work, HBM/latency bound; sizes are a few thousand rows per batch.
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 1024;

__global__ void count_nonzero_rows(const int64_t* __restrict__ padded, int rows, int width,
                                   int32_t* __restrict__ counts) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t* p = padded + (size_t)r * width;
    int c = 0;
    for (int k = 0; k < width; ++k) c += (p[k] != 0);
    counts[r] = c;
}

// Exclusive scan of counts[0..n) into out[0..n], out[n] = total. One block; in-place allowed.
__global__ void __launch_bounds__(SCAN_THREADS) exclusive_scan_1block(const int32_t* counts, int n,
                                                                      int32_t* out) {
    __shared__ int32_t partial[SCAN_THREADS];
    const int t = threadIdx.x;
    const int ipt = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(t * ipt, n), hi = min(lo + ipt, n);
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    partial[t] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        int32_t v = (t >= off) ? partial[t - off] : 0;
        __syncthreads();
        partial[t] += v;
        __syncthreads();
    }
    int32_t run = partial[t] - sum;   // exclusive prefix of this thread's segment
    for (int i = lo; i < hi; ++i) {
        int32_t c = counts[i];
        out[i] = run;
        run += c;
    }
    if (t == SCAN_THREADS - 1) out[n] = partial[t];
}

__global__ void fill_csr_from_padded(const int64_t* __restrict__ padded, int rows, int width,
                                     const int32_t* __restrict__ rowptr, int32_t* __restrict__ col) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t* p = padded + (size_t)r * width;
    int o = rowptr[r];
    for (int k = 0; k < width; ++k) {
        int64_t v = p[k];
        if (v != 0) col[o++] = (int32_t)v;
    }
}

__global__ void zero_i32(int32_t* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ void count_columns(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int rows,
                              int ncols, int32_t* __restrict__ counts) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
        int c = col[j];
        if (c >= 0 && c < ncols) atomicAdd(&counts[c], 1);   // integer atomics: the COUNT is order independent
    }
}

__global__ void fill_transpose(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int rows,
                               int ncols, const int32_t* __restrict__ rowptrT, int32_t* __restrict__ cursor,
                               int32_t* __restrict__ colT) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
        int c = col[j];
        if (c >= 0 && c < ncols) {
            int pos = atomicAdd(&cursor[c], 1);
            colT[rowptrT[c] + pos] = r;
        }
    }
}

// Slot order above depends on atomic arrival; sorting each (short) list ascending makes the
// transpose -- and therefore every floating-point sum that walks it -- bitwise reproducible.
__global__ void sort_lists(const int32_t* __restrict__ rowptrT, int32_t* __restrict__ colT, int ncols) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    int lo = rowptrT[c], hi = rowptrT[c + 1];
    for (int i = lo + 1; i < hi; ++i) {
        int32_t v = colT[i];
        int j = i - 1;
        while (j >= lo && colT[j] > v) { colT[j + 1] = colT[j]; --j; }
        colT[j + 1] = v;
    }
}

__global__ void extract_column_k(const int64_t* __restrict__ mat, int rows, int width, int column,
                                 int32_t* __restrict__ out) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = (int32_t)mat[(size_t)r * width + column];
}

// ---- single-workgroup fast paths (small graphs: one launch instead of 3 / 6) -----------------------------
constexpr int SMALL_ROWS = 16384;     // rows (and transposed columns) handled by one 1024-thread workgroup

__device__ __forceinline__ void block_exclusive_scan_inplace(int32_t* data, int n, int32_t* partial) {
    // data[0..n) counts -> exclusive prefix; data[n] = total.  partial: SCAN_THREADS ints of LDS.
    const int t = threadIdx.x;
    const int ipt = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int lo = min(t * ipt, n), hi = min(lo + ipt, n);
    int32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += data[i];
    partial[t] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        int32_t v = (t >= off) ? partial[t - off] : 0;
        __syncthreads();
        partial[t] += v;
        __syncthreads();
    }
    int32_t run = partial[t] - sum;
    for (int i = lo; i < hi; ++i) {
        int32_t c = data[i];
        data[i] = run;
        run += c;
    }
    if (t == SCAN_THREADS - 1) data[n] = partial[t];
    __syncthreads();
}

__global__ void __launch_bounds__(SCAN_THREADS) padded_to_csr_small(const int64_t* __restrict__ padded, int rows,
                                                                     int width, int32_t* __restrict__ rowptr,
                                                                     int32_t* __restrict__ col) {
    __shared__ int32_t partial[SCAN_THREADS];
    for (int r = threadIdx.x; r < rows; r += SCAN_THREADS) {
        const int64_t* p = padded + (size_t)r * width;
        int c = 0;
        for (int k = 0; k < width; ++k) c += (p[k] != 0);
        rowptr[r] = c;
    }
    __syncthreads();
    block_exclusive_scan_inplace(rowptr, rows, partial);
    for (int r = threadIdx.x; r < rows; r += SCAN_THREADS) {
        const int64_t* p = padded + (size_t)r * width;
        int o = rowptr[r];
        for (int k = 0; k < width; ++k) {
            const int64_t v = p[k];
            if (v != 0) col[o++] = (int32_t)v;
        }
    }
}

// Ascending order inside every transposed row makes the backward sums deterministic.  Rows of up to SORT_SERIAL entries are
// sorted by the thread that owns the column; longer ones (an index with few distinct ids: the molecule of every prediction
// row, 90 rows per molecule -- 320 us of serial insertion sort through global memory before) by RANK through LDS: one wave
// per row of up to SORT_WAVE_CAP entries, the whole workgroup per row beyond that (up to 16 384 entries).
constexpr int SORT_SERIAL = 24, SORT_WAVE_CAP = 512, SORT_WAVES = SCAN_THREADS / 64, SORT_LIST = 512;
constexpr int SORT_BLOCK_CAP = SORT_WAVES * SORT_WAVE_CAP;      // entries of one LDS chunk of the workgroup path
constexpr int SORT_PER_THREAD = 16;                             // workgroup path: rows of up to 16 x 1024 entries

__device__ __forceinline__ void insertion_sort_global(int32_t* colT, int lo, int hi) {
    for (int i = lo + 1; i < hi; ++i) {
        const int32_t v = colT[i];
        int j = i - 1;
        while (j >= lo && colT[j] > v) { colT[j + 1] = colT[j]; --j; }
        colT[j + 1] = v;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) csr_transpose_small(const int32_t* __restrict__ rowptr,
                                                                     const int32_t* __restrict__ col, int rows,
                                                                     int ncols, int32_t* __restrict__ rowptrT,
                                                                     int32_t* __restrict__ colT,
                                                                     int32_t* __restrict__ cursor) {
    __shared__ int32_t partial[SCAN_THREADS];
    __shared__ int32_t stage[SORT_BLOCK_CAP];
    __shared__ int32_t wave_list[SORT_LIST], block_list[SORT_LIST];
    __shared__ int32_t n_wave, n_block;
    if (threadIdx.x == 0) { n_wave = 0; n_block = 0; }
    for (int c = threadIdx.x; c <= ncols; c += SCAN_THREADS) rowptrT[c] = 0;
    for (int c = threadIdx.x; c < ncols; c += SCAN_THREADS) cursor[c] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += SCAN_THREADS)
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            const int c = col[j];
            if (c >= 0 && c < ncols) atomicAdd(&rowptrT[c], 1);
        }
    __syncthreads();
    block_exclusive_scan_inplace(rowptrT, ncols, partial);
    for (int r = threadIdx.x; r < rows; r += SCAN_THREADS)
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            const int c = col[j];
            if (c >= 0 && c < ncols) colT[rowptrT[c] + atomicAdd(&cursor[c], 1)] = r;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < ncols; c += SCAN_THREADS) {
        const int lo = rowptrT[c], hi = rowptrT[c + 1], n = hi - lo;
        if (n <= SORT_SERIAL) { insertion_sort_global(colT, lo, hi); continue; }
        int slot = -1;
        if (n <= SORT_WAVE_CAP) { slot = atomicAdd(&n_wave, 1); if (slot < SORT_LIST) wave_list[slot] = c; }
        else if (n <= SORT_PER_THREAD * SCAN_THREADS) { slot = atomicAdd(&n_block, 1); if (slot < SORT_LIST) block_list[slot] = c; }
        if (slot < 0 || slot >= SORT_LIST) insertion_sort_global(colT, lo, hi);      // (no room in the lists / longer than the LDS)
    }
    __syncthreads();
    const int nw = min((int)n_wave, SORT_LIST), nb = min((int)n_block, SORT_LIST);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = 0; base < nw; base += SORT_WAVES) {         // (uniform trip count: the barriers below are reached by every wave)
        const int k = base + wave;
        int lo = 0, n = 0;
        if (k < nw) { const int c = wave_list[k]; lo = rowptrT[c]; n = rowptrT[c + 1] - lo; }
        int32_t* mine = stage + wave * SORT_WAVE_CAP;
        for (int i = lane; i < n; i += 64) mine[i] = colT[lo + i];
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            const int32_t v = mine[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) { const int32_t x = mine[j]; rank += (x < v) || (x == v && j < i); }
            colT[lo + rank] = v;
        }
        __syncthreads();
    }
    for (int k = 0; k < nb; ++k) {
        // up to SORT_PER_THREAD entries per thread stay in registers while the row passes through the LDS in chunks
        const int c = block_list[k], lo = rowptrT[c], n = rowptrT[c + 1] - lo;
        int32_t v[SORT_PER_THREAD];
        int rank[SORT_PER_THREAD];
#pragma unroll
        for (int q = 0; q < SORT_PER_THREAD; ++q) {
            const int i = threadIdx.x + q * SCAN_THREADS;
            v[q] = i < n ? colT[lo + i] : 0;
            rank[q] = 0;
        }
        for (int c0 = 0; c0 < n; c0 += SORT_BLOCK_CAP) {
            const int cn = min(SORT_BLOCK_CAP, n - c0);
            for (int j = threadIdx.x; j < cn; j += SCAN_THREADS) stage[j] = colT[lo + c0 + j];
            __syncthreads();
            for (int j = 0; j < cn; ++j) {
                const int32_t x = stage[j];
                const int gj = c0 + j;
#pragma unroll
                for (int q = 0; q < SORT_PER_THREAD; ++q)
                    rank[q] += (x < v[q]) || (x == v[q] && gj < threadIdx.x + q * SCAN_THREADS);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < SORT_PER_THREAD; ++q)
            if (threadIdx.x + q * SCAN_THREADS < n) colT[lo + rank[q]] = v[q];
        __syncthreads();
    }
}

}  // namespace

extern "C" int ggpm_padded_to_csr(const int64_t* padded, int rows, int width, int32_t* rowptr, int32_t* col,
                                  ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!padded || !rowptr || !col || rows <= 0 || width <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (rows <= SMALL_ROWS) {
        padded_to_csr_small<<<1, SCAN_THREADS, 0, s>>>(padded, rows, width, rowptr, col);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    const int T = 256, B = ggpm_ceil_div(rows, T);
    count_nonzero_rows<<<B, T, 0, s>>>(padded, rows, width, rowptr);
    exclusive_scan_1block<<<1, SCAN_THREADS, 0, s>>>(rowptr, rows, rowptr);
    fill_csr_from_padded<<<B, T, 0, s>>>(padded, rows, width, rowptr, col);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_csr_transpose(const int32_t* rowptr, const int32_t* col, int rows, int ncols,
                                  int32_t* rowptrT, int32_t* colT, int32_t* cursor, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!rowptr || !col || !rowptrT || !colT || !cursor || rows <= 0 || ncols <= 0) return GGPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (rows <= SMALL_ROWS && ncols <= SMALL_ROWS) {
        csr_transpose_small<<<1, SCAN_THREADS, 0, s>>>(rowptr, col, rows, ncols, rowptrT, colT, cursor);
        GGPM_CHECK_LAUNCH();
        return GGPM_OK;
    }
    const int T = 256;
    zero_i32<<<ggpm_ceil_div(ncols + 1, T), T, 0, s>>>(rowptrT, ncols + 1);
    zero_i32<<<ggpm_ceil_div(ncols, T), T, 0, s>>>(cursor, ncols);
    count_columns<<<ggpm_ceil_div(rows, T), T, 0, s>>>(rowptr, col, rows, ncols, rowptrT);
    exclusive_scan_1block<<<1, SCAN_THREADS, 0, s>>>(rowptrT, ncols, rowptrT);
    fill_transpose<<<ggpm_ceil_div(rows, T), T, 0, s>>>(rowptr, col, rows, ncols, rowptrT, cursor, colT);
    sort_lists<<<ggpm_ceil_div(ncols, T), T, 0, s>>>(rowptrT, colT, ncols);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

extern "C" int ggpm_extract_column(const int64_t* mat, int rows, int width, int column, int32_t* out,
                                   ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    if (!mat || !out || rows <= 0 || column < 0 || column >= width) return GGPM_ERR_ARG;
    extract_column_k<<<ggpm_ceil_div(rows, 256), 256, 0, (hipStream_t)stream>>>(mat, rows, width, column, out);
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
