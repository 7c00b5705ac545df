// ulp error of the three sigmoid forms of csrc/common.h on the device, against the double-precision value on the host
// (dev tool, standalone):
//   hipcc --offload-arch=gfx950 -O3 tools/probe/sigmoid_ulp.hip -o tools/probe/sigmoid_ulp && tools/probe/sigmoid_ulp
// Sweeps 2^24 arguments over [-30, 30] (dense near 0: x = 30 * u^3) plus every float in a few binades around the
// sigmoid's steep part, and times 64 evaluations per thread of each form (VALU issue cost, no memory).
#include "../../ggpm_amd/csrc/common.h"
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void eval(const float* __restrict__ x, float* __restrict__ fast, float* __restrict__ acc, float* __restrict__ libm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = ggpm_fsigmoid_fast(x[i]);
    acc[i] = ggpm_fsigmoid_acc(x[i]);
    libm[i] = ggpm_sigmoid(x[i]);
}

template <int F>
__global__ void __launch_bounds__(256) spin(float* out, float seed, int iters) {
    float v[8], s = 0.f;
    for (int k = 0; k < 8; ++k) v[k] = seed * (threadIdx.x % 17 + k) - 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float r = F == 0 ? ggpm_fsigmoid_fast(v[k]) : F == 1 ? ggpm_fsigmoid_acc(v[k]) : ggpm_sigmoid(v[k]);
            s += r;
            v[k] = v[k] + r * 1e-3f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double ulp_err(float got, double want) {
    const float w = (float)want;
    int e;
    frexpf(w, &e);
    const double ulp = ldexp(1.0, e - 24);
    return fabs((double)got - want) / ulp;
}

int main() {
    std::vector<float> xs;
    const int N0 = 1 << 24;
    for (int i = 0; i < N0; ++i) {
        const double u = 2.0 * (i + 0.5) / N0 - 1.0;
        xs.push_back((float)(30.0 * u * u * u));
    }
    for (float x = 0.25f; x < 16.f; x = nextafterf(x, 100.f) + 3e-6f * x) { xs.push_back(x); xs.push_back(-x); }
    for (float x : {0.f, -0.f, 88.f, -88.f, 100.f, -100.f, 1e30f, -1e30f, 1e-30f}) xs.push_back(x);
    const int n = (int)xs.size();
    float *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    eval<<<(n + 255) / 256, 256>>>(dx, d0, d1, d2, n);
    std::vector<float> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), d0, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    const char* names[3] = {"fast (v_exp(-x*log2e), v_rcp)", "accurate (two-term product, Newton rcp)", "libm expf + IEEE division"};
    const std::vector<float>* rs[3] = {&r0, &r1, &r2};
    for (int f = 0; f < 3; ++f) {
        double worst = 0, sum = 0, worst_x = 0;
        long bad = 0, nan = 0;
        for (int i = 0; i < n; ++i) {
            const double want = 1.0 / (1.0 + exp(-(double)xs[i]));
            const float g = (*rs[f])[i];
            if (g != g) { ++nan; continue; }
            if (want < 1e-37) continue;                 // denormal results: flushed
            const double e = ulp_err(g, want);
            sum += e;
            if (e > worst) { worst = e; worst_x = xs[i]; }
            if (e > 1.0) ++bad;
        }
        printf("%-42s max %.3f ulp (at x = %.6g)  mean %.4f ulp  > 1 ulp: %ld of %d  NaN: %ld\n", names[f], worst, worst_x, sum / n, bad, n, nan);
    }
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f = 0; f < 3; ++f) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f == 0) spin<0><<<1024, 256>>>(out, 0.37f, 4096);
            if (f == 1) spin<1><<<1024, 256>>>(out, 0.37f, 4096);
            if (f == 2) spin<2><<<1024, 256>>>(out, 0.37f, 4096);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-42s %.3f ms for 2^33 evaluations\n", names[f], ms);
    }
    return 0;
}
