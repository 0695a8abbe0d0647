// Optional per-kernel timing with HIP events on the launch stream (used by bench.py's roofline leg).
// Off by default: a disabled CFD_PROF scope costs one predictable branch.  Not usable during stream capture.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cfd_common.h"

namespace {
struct Rec {
    const char* name;
    hipEvent_t a, b;
    double bytes, flops;  // algorithmic HBM bytes / flops this launch stands for (0 = not declared)
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

CfdProfScope::CfdProfScope(const char* name, hipStream_t s, double bytes, double flops) : st(s), idx(-1) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{name, nullptr, nullptr, bytes, flops};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}

CfdProfScope::~CfdProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_recs[idx].b, st);
}

bool cfd_prof_active() { return g_on; }

extern "C" int cfd_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_on = true;
    return CFD_OK;
}

// Synchronises, writes one line per kernel "name count total_ms total_bytes total_flops\n" into buf, disables profiling.
extern "C" int cfd_prof_end(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = false;
    struct Agg { int n = 0; double ms = 0, bytes = 0, flops = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = agg[r.name];
            e.n += 1;
            e.ms += ms;
            e.bytes += r.bytes;
            e.flops += r.flops;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %d %.6f %.0f %.0f\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.bytes,
                 kv.second.flops);
        out += line;
    }
    if (buf && cap) {
        strncpy(buf, out.c_str(), cap - 1);
        buf[cap - 1] = 0;
    }
    return out.size() < cap ? CFD_OK : CFD_ERR_WORKSPACE;
}
