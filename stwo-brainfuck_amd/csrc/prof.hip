// Per-kernel HIP-event timing used by bench.py's roofline object: every kernel launch of the library can be bracketed by a pair of
// events on the launching stream; durations and algorithmic byte counts are aggregated per kernel name.
#include "kernels.h"
#include <map>
#include <vector>
#include <string>
#include <cstdio>
#include <cstdlib>

namespace bf {

struct ProfRec { const char* name; double bytes; hipEvent_t e0, e1; u64 calls; };
struct ProfAgg { u64 calls = 0; double ms = 0, bytes = 0; };

static int g_prof_on = 0;   // 0 off, 1 every instrumented kernel, 2 only the Merkle layer kernel, one event pair per run of back-to-back launches
static bool g_run = false;  // mode 2: inside a run (prof_run_begin .. prof_run_end) launches only add their counts to the run's record
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::map<std::string, ProfAgg> g_agg;

int prof_mode() { return g_prof_on; }
void prof_enable(int mode) { g_prof_on = mode; }

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

void prof_begin(hipStream_t s, const char* name, double bytes) {
    if (g_run) { g_recs.back().bytes += bytes; g_recs.back().calls++; return; }
    ProfRec r{name, bytes, get_event(), get_event(), 1};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s) { if (!g_run) (void)hipEventRecord(g_recs.back().e1, s); }
// A run of consecutive launches of one kernel on one stream with nothing else in between (the layers of one Merkle tree): one event
// pair brackets the whole run, so the instrumentation costs ~60 instead of ~500 event records per proof. The time of a run includes
// the (sub-microsecond) dispatch gaps between its launches.
void prof_run_begin(hipStream_t s, const char* name) {
    if (g_prof_on != 2 || g_run) return;
    ProfRec r{name, 0.0, get_event(), get_event(), 0};
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    g_run = true;
}
void prof_run_end(hipStream_t s) {
    if (!g_run) return;
    g_run = false;
    (void)hipEventRecord(g_recs.back().e1, s);
}

// Must be called after the stream has been synchronised.
void prof_collect() {
    for (auto& r : g_recs) {
        float ms = 0;
        if (r.calls && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { auto& a = g_agg[r.name]; a.calls += r.calls; a.ms += ms; a.bytes += r.bytes; }
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
}
void prof_reset() { g_run = false; prof_collect(); g_agg.clear(); }
std::string prof_report_json() {
    prof_collect();
    std::string s = "{";
    bool first = true;
    for (auto& kv : g_agg) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\":{\"calls\":%llu,\"total_ms\":%.6f,\"bytes\":%.0f}", first ? "" : ",", kv.first.c_str(), (unsigned long long)kv.second.calls, kv.second.ms, kv.second.bytes);
        s += buf; first = false;
    }
    return s + "}";
}

}  // namespace bf
