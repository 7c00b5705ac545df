// Library-level entry points: version, error strings, padding rule, optional launch timing.
#include "common.h"
#include <mutex>
#include <vector>

extern "C" int ggpm_version(void) { return 100; }

extern "C" const char* ggpm_error_string(int code) {
    switch (code) {
        case GGPM_OK: return "ok";
        case GGPM_ERR_ARG: return "invalid argument (size, null pointer or leading dimension)";
        case GGPM_ERR_LAUNCH: return "HIP kernel launch failed";
        case GGPM_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case GGPM_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

extern "C" int ggpm_padded_hidden(int H) { return ggpm_round_up(H, 16); }

// ---------------------------------------------------------------- timing sink (debug/bench only)
namespace {
struct Span { hipEvent_t a, b; double flops; };
constexpr int NKIND = 8, NTAG = 4, NCLASS = NKIND * NTAG;
std::mutex g_mu;
bool g_on = false;
std::vector<Span> g_spans[NCLASS];
thread_local int g_tag = 0;
}  // namespace

// The whole-encoder drivers tag the launches of a level (1 atom, 2 attachment, 3 motif level; 0 = untagged) so that the
// bench can report the atom-level launch (one workgroup per CU, MFMA bound) apart from the small latency-bound levels.
void ggpm_timing_tag(int tag) { g_tag = (tag >= 0 && tag < NTAG) ? tag : 0; }

void ggpm_timing_begin(int which, hipStream_t s, double flops) {
    if (!g_on) return;
    which += NKIND * g_tag;
    std::lock_guard<std::mutex> lk(g_mu);
    Span sp;
    sp.flops = flops;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) return;
    (void)hipEventRecord(sp.a, s);
    g_spans[which].push_back(sp);
}

void ggpm_timing_end(int which, hipStream_t s) {
    if (!g_on) return;
    which += NKIND * g_tag;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_spans[which].empty()) (void)hipEventRecord(g_spans[which].back().b, s);
}

extern "C" int ggpm_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return GGPM_OK;
}

extern "C" int ggpm_timing_collect(int which, int* launches, double* total_ms, double* flops) {
    if (which < 0 || which >= NCLASS || !launches || !total_ms || !flops) return GGPM_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    double ms = 0.0, fl = 0.0;
    for (Span& sp : g_spans[which]) {
        float t = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) {
            ms += t; fl += sp.flops; ++n;
        }
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    g_spans[which].clear();
    *launches = n; *total_ms = ms; *flops = fl;
    return GGPM_OK;
}
