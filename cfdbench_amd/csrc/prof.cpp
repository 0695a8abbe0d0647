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
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

CfdProfScope::CfdProfScope(const char* name, hipStream_t s) : st(s), idx(-1) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{name, nullptr, nullptr};
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

extern "C" int cfd_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_recs.clear();
    g_on = true;
    return CFD_OK;
}

// Synchronises, writes one line per kernel "name count total_ms\n" into buf, disables profiling.
extern "C" int cfd_prof_end(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = false;
    std::map<std::string, std::pair<int, double>> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = agg[r.name];
            e.first += 1;
            e.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && cap) {
        strncpy(buf, out.c_str(), cap - 1);
        buf[cap - 1] = 0;
    }
    return out.size() < cap ? CFD_OK : CFD_ERR_WORKSPACE;
}
