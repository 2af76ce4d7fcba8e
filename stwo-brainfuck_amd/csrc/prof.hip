// Per-kernel HIP-event timing used by bench.py's roofline object: a kernel launch of the library can be bracketed by a pair of events on
// the launching stream; durations, algorithmic byte counts and work units (Blake2s compressions for the Merkle kernels) are aggregated
// per kernel name. State is kept PER STREAM (a context owns two streams), so contexts driven from different host threads neither race
// nor read each other's pending events; the only shared object is the stream -> state map behind a mutex, touched only while some
// stream has profiling switched on.
#include "kernels.h"
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <cstdio>
#include <cstdlib>

namespace bf {

struct ProfRec { const char* name; double bytes, units; hipEvent_t e0, e1; u64 calls; double aux; };
struct ProfAgg { u64 calls = 0; double ms = 0, bytes = 0, units = 0, aux = 0; };
struct ProfState {
    int mode = 0;          // 0 off, 1 every instrumented kernel, 2 only the Merkle layer kernel, one event pair per run of back-to-back launches
    bool run = false;      // mode 2: inside a run (prof_run_begin .. prof_run_end) launches only add their counts to the run's record
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;   // events belong to the device of the stream's context
    std::map<std::string, ProfAgg> agg;
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};

static std::mutex g_mu;
static std::map<hipStream_t, ProfState> g_states;
static std::atomic<int> g_enabled_streams{0};

// state of a stream, or nullptr when profiling is off for it (the common case costs one relaxed atomic load)
static ProfState* state_of(hipStream_t s) {
    if (g_enabled_streams.load(std::memory_order_relaxed) == 0) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_states.find(s);
    return it == g_states.end() || it->second.mode == 0 ? nullptr : &it->second;   // map nodes are address-stable; a stream is driven by one thread
}

int prof_mode(hipStream_t s) { ProfState* st = state_of(s); return st ? st->mode : 0; }
void prof_enable(hipStream_t s, int mode) {
    std::lock_guard<std::mutex> g(g_mu);
    ProfState& st = g_states[s];
    if ((st.mode != 0) != (mode != 0)) g_enabled_streams += mode ? 1 : -1;
    st.mode = mode;
}

void prof_begin(hipStream_t s, const char* name, double bytes, double units, double aux) {
    ProfState* st = state_of(s);
    if (!st) return;
    if (st->run) { st->recs.back().bytes += bytes; st->recs.back().units += units; st->recs.back().aux += aux; st->recs.back().calls++; return; }
    ProfRec r{name, bytes, units, st->get_event(), st->get_event(), 1, aux};
    (void)hipEventRecord(r.e0, s);
    st->recs.push_back(r);
}
void prof_end(hipStream_t s) { ProfState* st = state_of(s); if (st && !st->run) (void)hipEventRecord(st->recs.back().e1, s); }
// A run of consecutive launches of one kernel on one stream with nothing else in between (the layers of one Merkle tree): one event
// pair brackets the whole run, so the instrumentation costs ~60 instead of ~500 event records per proof. The time of a run includes
// the (sub-microsecond) dispatch gaps between its launches.
void prof_run_begin(hipStream_t s, const char* name) {
    ProfState* st = state_of(s);
    if (!st || st->mode != 2 || st->run) return;
    ProfRec r{name, 0.0, 0.0, st->get_event(), st->get_event(), 0, 0.0};
    (void)hipEventRecord(r.e0, s);
    st->recs.push_back(r);
    st->run = true;
}
void prof_run_end(hipStream_t s) {
    ProfState* st = state_of(s);
    if (!st || !st->run) return;
    st->run = false;
    (void)hipEventRecord(st->recs.back().e1, s);
}

// The stream must have been synchronised by the caller.
static void collect(ProfState& st) {
    for (auto& r : st.recs) {
        float ms = 0;
        if (r.calls && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { auto& a = st.agg[r.name]; a.calls += r.calls; a.ms += ms; a.bytes += r.bytes; a.units += r.units; a.aux += r.aux; }
        st.pool.push_back(r.e0); st.pool.push_back(r.e1);
    }
    st.recs.clear();
}
void prof_reset(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_states.find(s);
    if (it == g_states.end()) return;
    it->second.run = false; collect(it->second); it->second.agg.clear();
}
// Aggregate of the given (synchronised) streams of one context.
std::string prof_report_json(const hipStream_t* streams, int n) {
    std::map<std::string, ProfAgg> sum;
    {
        std::lock_guard<std::mutex> g(g_mu);
        for (int i = 0; i < n; i++) {
            auto it = g_states.find(streams[i]);
            if (it == g_states.end()) continue;
            collect(it->second);
            for (auto& kv : it->second.agg) { auto& a = sum[kv.first]; a.calls += kv.second.calls; a.ms += kv.second.ms; a.bytes += kv.second.bytes; a.units += kv.second.units; a.aux += kv.second.aux; }
        }
    }
    std::string s = "{";
    bool first = true;
    for (auto& kv : sum) {
        char buf[320];
        snprintf(buf, sizeof buf, "%s\"%s\":{\"calls\":%llu,\"total_ms\":%.6f,\"bytes\":%.0f,\"units\":%.0f,\"aux\":%.0f}", first ? "" : ",", kv.first.c_str(), (unsigned long long)kv.second.calls,
                 kv.second.ms, kv.second.bytes, kv.second.units, kv.second.aux);
        s += buf; first = false;
    }
    return s + "}";
}
// Releases the events of a stream that is about to be destroyed.
void prof_forget(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_states.find(s);
    if (it == g_states.end()) return;
    if (it->second.mode != 0) g_enabled_streams -= 1;
    for (auto& r : it->second.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : it->second.pool) (void)hipEventDestroy(e);
    g_states.erase(it);
}

}  // namespace bf
