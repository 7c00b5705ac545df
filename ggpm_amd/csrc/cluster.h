// Molecule clusters and in-launch synchronisation for the persistent depth-loop kernels (gfx950).
//
// The message-passing recurrence only couples messages of the same molecule, so a level's message rows fall
// into CLOSED row ranges (no predecessor edge crosses a range boundary).  A "cluster" is such a range of at
// least `target` rows; the workgroups that own the column groups of one cluster run all `depth` steps inside
// ONE launch and only ever wait for each other (never for the rest of the grid).
//
// Forward progress does not depend on dispatch order or placement: logical workgroup ids are handed out by a
// ticket counter in start order, the members of a cluster hold consecutive tickets, so every workgroup a
// spinning member waits for has either started or is the next to start, and all other resident clusters are
// complete and finish on their own.  Spins are bounded anyway (timeout word), so every wave reaches the exit.
//
// Hand-off protocol (MI355X guide, "inter-workgroup visibility", counter form): payload stored write-through
// (buffer stores with sc1), every storing wave drains vmcnt, workgroup barrier, ONE lane adds to the cluster's
// arrival counter (agent scope); the consumer polls that counter relaxed from one lane, a workgroup barrier
// follows, and every load of handed-off bytes bypasses the CU's L1 (buffer loads / LDS-DMA loads with sc1).
#pragma once
#include "common.h"

// table layout (int32): [0] = number of clusters n, [1 .. n+1] = row bounds (bounds[0] = 0, bounds[n] = E1)
static inline size_t ggpm_cluster_table_len(int E1) { return (size_t)E1 + 4; }

// sync block layout (uint32): [0] ticket, [1] timeout flag, [2..3] pad, [4 + c] arrival counter of cluster c
constexpr int GGPM_SYNC_HDR = 4;
constexpr unsigned GGPM_SPIN_LIMIT = 1u << 22;

#if defined(__HIPCC__)
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned ggpm_u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ggpm_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// write-through / L1-bypassing 16-byte accesses for handed-off data (aux 16 = sc1)
__device__ __forceinline__ f32x4 ggpm_xld(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void ggpm_xst(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ggpm_u32x4, v), r, byte_off, 0, 16);
}
// plain (L2-resident) form: only valid when every reader shares the writer's XCD
__device__ __forceinline__ void ggpm_xst_l2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ggpm_u32x4, v), r, byte_off, 0, 0);
}

// All waves of the workgroup call this after their exchange stores; returns false on timeout (uniform).
__device__ __forceinline__ bool ggpm_cluster_sync(unsigned* cnt, unsigned target, unsigned* tmo, int* lds_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        for (unsigned spins = 0;; ++spins) {
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            if (spins > GGPM_SPIN_LIMIT || __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *lds_ok = ok;
    }
    __syncthreads();
    return *lds_ok != 0;
}
#endif

// implemented in cluster.hip
int ggpm_build_clusters_impl(const int32_t* rowptr, const int32_t* col, int E1, int target, int32_t* table,
                             int32_t* scratch, hipStream_t s);
