// Sustained fp32 MFMA rate and shader clock under load (dev tool, standalone):
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_clock.hip -o tools/probe/mfma_clock && tools/probe/mfma_clock
// Every wave runs `chains` independent v_mfma_f32_32x32x2_f32 accumulation chains; lane 0 of workgroup 0 stamps
// s_memtime (shader clock) and the 100 MHz wall clock around the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* st, int iters, float seed) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = seed * (threadIdx.x % 7 + 1), b = seed * (threadIdx.x % 5 + 1);
    unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st[0] = t1 - t0; st[1] = w1 - w0; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same with v_mfma_f32_16x16x4_f32 and CHAINS independent accumulator blocks (distinct a / b operands per block row / column)
template <int R>
__global__ void __launch_bounds__(256) probe16(float* out, unsigned long long* st, int iters, float seed) {
    f32x4 acc[R][R];
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[R], b[R];
    for (int i = 0; i < R; ++i) { a[i] = seed * (threadIdx.x % 7 + i); b[i] = seed * (threadIdx.x % 5 + i); }
    unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st[0] = t1 - t0; st[1] = w1 - w0; }
}

template <int R>
static void run16(int wgs, int iters, float seed) {
    float* out; unsigned long long* st;
    hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&st, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe16<R><<<wgs, 256>>>(out, st, iters, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe16<R><<<wgs, 256>>>(out, st, iters, seed);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * R * R * 2048.0;
    printf("16x16x4: wgs %5d blocks %dx%d iters %6d seed %g: %8.1f us  %6.1f TF/s  shader clock %.2f GHz  %.1f cycles per MFMA per wave\n",
           wgs, R, R, iters, seed, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)h[0] / (h[1] * 10.0),
           (double)h[0] / ((double)iters * R * R));
    hipFree(out); hipFree(st);
}

template <int CHAINS>
static void run(int wgs, int iters, float seed) {
    float* out; unsigned long long* st;
    hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&st, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CHAINS><<<wgs, 256>>>(out, st, iters, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<CHAINS><<<wgs, 256>>>(out, st, iters, seed);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * CHAINS * 4096.0;
    printf("wgs %5d chains %d iters %6d seed %g: %8.1f us  %6.1f TF/s  shader clock %.2f GHz  %.1f cycles per MFMA per wave\n",
           wgs, CHAINS, iters, seed, ms * 1e3, flops / (ms * 1e-3) / 1e12, (double)h[0] / (h[1] * 10.0) / 1e0 / 1e0 * 1e-0 / 1e0,
           (double)h[0] / ((double)iters * CHAINS));
    hipFree(out); hipFree(st);
}

int main() {
    for (float seed : {0.f, 1e-3f}) {
        run16<1>(256, 16000, seed);
        run16<2>(256, 4000, seed);
        run16<5>(256, 800, seed);
        run16<5>(512, 800, seed);
        run<1>(256, 8000, seed);
        run<2>(256, 4000, seed);
        run<1>(1024, 2000, seed);
        run<2>(1024, 4000, seed);
        run<2>(1280, 4000, seed);
        run<2>(1024, 16000, seed);
    }
    return 0;
}
