// Which CUs does a CU-masked stream use? (dev tool)  hipcc --offload-arch=gfx950 -O2 tools/probe/cu_mask.hip -o tools/probe/cu_mask
// Launches 2048 one-wave workgroups that spin ~20 us each on a stream created with hipExtStreamCreateWithCUMask and
// records (XCC id, SE id, CU id) per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <map>
#include <vector>

__global__ void where(unsigned* out) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* label, const uint32_t* mask) {
    hipStream_t s;
    if (mask) {
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", label); return; }
    } else {
        hipStreamCreate(&s);
    }
    const int n = 2048;
    unsigned* d; hipMalloc(&d, n * 8);
    where<<<n, 64, 0, s>>>(d);
    if (hipStreamSynchronize(s) != hipSuccess) { printf("%s: launch failed\n", label); return; }
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;      // xcc -> distinct (se, sh, cu)
    std::map<unsigned, int> per_xcc;
    for (int i = 0; i < n; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[xcc].insert((se << 8) | (sh << 4) | cu);
        per_xcc[xcc]++;
    }
    int total = 0;
    printf("%s:", label);
    for (auto& kv : cus) { printf("  xcc%u: %d wgs on %zu CUs", kv.first, per_xcc[kv.first], kv.second.size()); total += kv.second.size(); }
    printf("  | %d distinct CUs\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    run("no mask          ", nullptr);
    uint32_t m[8];
    for (int i = 0; i < 8; ++i) m[i] = 0x00ffffffu;
    run("0x00ffffff x8    ", m);
    for (int i = 0; i < 8; ++i) m[i] = 0xff000000u;
    run("0xff000000 x8    ", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x77777777u;
    run("0x77777777 x8    ", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x88888888u;
    run("0x88888888 x8    ", m);
    for (int i = 0; i < 8; ++i) m[i] = i < 6 ? 0xffffffffu : 0u;
    run("words 0-5        ", m);
    return 0;
}
