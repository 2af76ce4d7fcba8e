// Proofs in flight as a capability of the library (include/bfhip.h: bfhip_pool_*): ONE caller thread — the reference's prove_brainfuck has a
// single thread of control (crates/brainfuck_prover/src/brainfuck_air/mod.rs:471-735; SURVEY.md section 8(b) "Who calls it") — hands a batch
// of resident traces (or of programs) to a pool of k sub-contexts on one GPU; k internal worker threads prove them, k at a time, and the call
// returns when every proof of the batch is done. While one proof sits in a single-workgroup chain (tree tops, small FRI layers) or waits for
// its host at a Fiat-Shamir point, the wide kernels of another fill the GPU: +19 % (2 in flight) / +23 % (3) at 2^22 rows, +36..57 % at 2^20
// (profiles/r05_inflight.jsonl, measured with k Python threads over k full contexts; this file makes it one C call).
//
// What the sub-contexts share (all byte-neutral):
//   - the twiddle tree and the circle-point tables of sub-context 0 (read-only after creation; 2 x 2^(max_log_domain - 1) words not held k times);
//   - by default ONE preprocessed commitment per batch (IsFirst(LOG_MAX_ROWS ..= 4): trace independent, mod.rs:495-500 recommits it in every
//     prove_brainfuck call): a builder context enqueues it before the workers start and every proof of the batch reads that tree
//     (prover.hip: SharedPreprocessed). With it a worker needs ONE stream, so k workers + the builder stay within the 4 hardware queues a
//     process gets by default (ctx.h: ensure_aux / ensure_side).
// Everything else is per sub-context as before: arena, staging ring, pinned slots, streams.
#include "../../include/bfhip.h"
#include "ctx.h"
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <string>
#include <chrono>

using namespace bf;

namespace {

struct Job {
    const bfhip_trace* const* traces = nullptr;            // kind 0: resident traces (bfhip_prove_trace per proof)
    const char* const* codes = nullptr;                     // kind 1: programs (bfhip_prove_brainfuck per proof: VM + tables inside the worker)
    const uint8_t* const* inputs = nullptr; const size_t* n_inputs = nullptr;
    uint32_t n = 0, log_max_rows = 0;
    char** json = nullptr; size_t* len = nullptr; int32_t* status = nullptr; double* seconds = nullptr;
    std::atomic<uint32_t> next{0}, failed{0};
    std::mutex err_mu; std::string first_error; uint32_t first_failed = 0xFFFFFFFFu;
};

}  // namespace

struct bfhip_pool {
    int device = 0; uint32_t k = 0, max_log_domain = 0;
    std::vector<bfhip_ctx*> subs;               // subs[0] owns the twiddle tree and the point tables
    bfhip_ctx* builder = nullptr;               // commits the shared preprocessed tree; its arena holds it
    SharedPreprocessed* shared = nullptr;
    int pre_mode = 1;                           // 0: every proof commits its own (the reference's behaviour), 1: once per batch, 2: kept across batches
    std::vector<std::thread> threads;
    std::mutex mu; std::condition_variable cv_work, cv_done;
    uint64_t generation = 0; bool quit = false; uint32_t active = 0;
    Job* job = nullptr;
    std::mutex call_mu;                         // batches of one pool are serial (a second caller thread waits)

    void worker(uint32_t w) {
        uint64_t seen = 0;
        for (;;) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation; j = job;
            }
            for (uint32_t i; (i = j->next.fetch_add(1)) < j->n;) run_one(*j, w, i);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
    void run_one(Job& j, uint32_t w, uint32_t i) {
        double phases[10] = {0};
        char* js = nullptr; size_t n = 0;
        int32_t rc;
        if (j.traces) rc = bfhip_prove_trace(subs[w], j.traces[i], j.log_max_rows, j.json ? &js : nullptr, &n, nullptr, phases);
        else rc = bfhip_prove_brainfuck(subs[w], j.codes[i], j.inputs ? j.inputs[i] : nullptr, j.inputs && j.n_inputs ? j.n_inputs[i] : 0, j.log_max_rows,
                                        j.json ? &js : nullptr, &n, nullptr, phases);
        if (j.json) j.json[i] = rc == 0 ? js : nullptr;
        if (j.len) j.len[i] = rc == 0 ? n : 0;
        if (j.status) j.status[i] = rc;
        if (j.seconds) j.seconds[i] = phases[9];
        if (rc != 0) {
            j.failed++;
            std::lock_guard<std::mutex> g(j.err_mu);
            if (i < j.first_failed) { j.first_failed = i; j.first_error = "proof " + std::to_string(i) + " of the batch: " + bfhip_last_error(); }
        }
    }
    int32_t run(Job& j) {
        std::lock_guard<std::mutex> call(call_mu);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0; i < j.n; i++) { if (j.json) j.json[i] = nullptr; if (j.len) j.len[i] = 0; if (j.status) j.status[i] = -1; if (j.seconds) j.seconds[i] = 0.0; }
        // the batch's preprocessed tree: enqueued on the builder's stream now, awaited by each proof where it first needs it
        const SharedPreprocessed* use = nullptr;
        if (pre_mode != 0 && j.n > 0) {
            builder->c.conv = subs[0]->c.conv;          // a worker whose conventions were changed individually does not match and commits its own
            if (pre_mode == 1 || !shared_preprocessed_matches(shared, builder->c, j.log_max_rows)) shared_preprocessed_build(shared, builder->c, j.log_max_rows);
            use = shared;
        }
        for (auto* s : subs) s->c.shared_pre = use;
        {
            std::unique_lock<std::mutex> lk(mu);
            job = &j; active = k; generation++;
            cv_work.notify_all();
            cv_done.wait(lk, [&] { return active == 0; });
            job = nullptr;
        }
        for (auto* s : subs) s->c.shared_pre = nullptr;
        if (j.seconds) j.seconds[j.n] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (j.failed.load()) { bfhip_set_error(j.first_error); return -1; }
        return 0;
    }
    ~bfhip_pool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_work.notify_all();
        for (auto& t : threads) if (t.joinable()) t.join();
        if (builder) { if (builder->c.stream) { (void)hipSetDevice(device); (void)hipStreamSynchronize(builder->c.stream); } }
        shared_preprocessed_destroy(shared);
        (void)bfhip_ctx_destroy(builder);
        for (size_t i = subs.size(); i-- > 0;) (void)bfhip_ctx_destroy(subs[i]);      // subs[0] (the tables' owner) last
    }
};

#define POOL_TRY try { if (!pool) throw HipError("null pool");
#define POOL_CATCH } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; } catch (...) { bfhip_set_error("unknown error"); return -1; }

extern "C" {

int32_t bfhip_pool_create(int32_t device_id, uint32_t n_in_flight, uint32_t max_log_domain, bfhip_pool** out) {
    bfhip_pool* pool = nullptr;
    try {
        if (!out) throw HipError("null argument");
        if (n_in_flight < 1 || n_in_flight > 16) throw HipError("bfhip_pool_create: n_in_flight must be in [1, 16]");
        pool = new bfhip_pool();
        pool->device = device_id; pool->k = n_in_flight; pool->max_log_domain = max_log_domain;
        for (uint32_t i = 0; i < n_in_flight; i++) {
            pool->subs.push_back(new bfhip_ctx());
            pool->subs.back()->c.init(device_id, max_log_domain, i ? &pool->subs[0]->c : nullptr);
        }
        pool->builder = new bfhip_ctx();
        pool->builder->c.init(device_id, max_log_domain, &pool->subs[0]->c);
        pool->shared = shared_preprocessed_create(pool->builder->c);
        for (uint32_t w = 0; w < n_in_flight; w++) pool->threads.emplace_back([pool, w] { pool->worker(w); });
        *out = pool;
        return 0;
    } catch (const std::exception& e) { delete pool; bfhip_set_error(e.what()); return -1; } catch (...) { delete pool; bfhip_set_error("unknown error"); return -1; }
}

int32_t bfhip_pool_destroy(bfhip_pool* pool) { try { delete pool; return 0; } catch (...) { bfhip_set_error("unknown error"); return -1; } }

int32_t bfhip_pool_size(bfhip_pool* pool, uint32_t* n_in_flight) { POOL_TRY if (!n_in_flight) throw HipError("null argument"); *n_in_flight = pool->k; return 0; POOL_CATCH }

int32_t bfhip_pool_ctx(bfhip_pool* pool, uint32_t i, bfhip_ctx** out) {
    POOL_TRY
    if (!out) throw HipError("null argument");
    if (i >= pool->k) throw HipError("bfhip_pool_ctx: index out of range");
    *out = pool->subs[i];
    return 0;
    POOL_CATCH
}

int32_t bfhip_pool_set_conventions(bfhip_pool* pool, const bfhip_conventions* conv) {
    POOL_TRY
    std::lock_guard<std::mutex> call(pool->call_mu);
    for (auto* s : pool->subs) if (bfhip_ctx_set_conventions(s, conv) != 0) return -1;
    if (bfhip_ctx_set_conventions(pool->builder, conv) != 0) return -1;
    shared_preprocessed_invalidate(pool->shared);
    return 0;
    POOL_CATCH
}

int32_t bfhip_pool_set_preprocessed(bfhip_pool* pool, int32_t mode) {
    POOL_TRY
    if (mode < 0 || mode > 2) throw HipError("bfhip_pool_set_preprocessed: 0 = per proof, 1 = per batch, 2 = kept across batches");
    std::lock_guard<std::mutex> call(pool->call_mu);
    pool->pre_mode = mode;
    if (mode != 2) shared_preprocessed_invalidate(pool->shared);
    return 0;
    POOL_CATCH
}

int32_t bfhip_prove_batch(bfhip_pool* pool, const bfhip_trace* const* traces, uint32_t n, uint32_t log_max_rows, char** proofs_json, size_t* proof_lens,
                          int32_t* statuses, double* seconds) {
    POOL_TRY
    if (n && !traces) throw HipError("null argument");
    for (uint32_t i = 0; i < n; i++) if (!traces[i]) throw HipError("null trace in the batch");
    Job j; j.traces = traces; j.n = n; j.log_max_rows = log_max_rows; j.json = proofs_json; j.len = proof_lens; j.status = statuses; j.seconds = seconds;
    return pool->run(j);
    POOL_CATCH
}

int32_t bfhip_prove_batch_brainfuck(bfhip_pool* pool, const char* const* codes, const uint8_t* const* inputs_h, const size_t* n_inputs, uint32_t n,
                                    uint32_t log_max_rows, char** proofs_json, size_t* proof_lens, int32_t* statuses, double* seconds) {
    POOL_TRY
    if (n && !codes) throw HipError("null argument");
    if (inputs_h && !n_inputs) throw HipError("inputs without their lengths");
    for (uint32_t i = 0; i < n; i++) {
        if (!codes[i]) throw HipError("null program text in the batch");
        if (inputs_h && n_inputs[i] && !inputs_h[i]) throw HipError("null input in the batch");
    }
    Job j; j.codes = codes; j.inputs = inputs_h; j.n_inputs = n_inputs; j.n = n; j.log_max_rows = log_max_rows; j.json = proofs_json; j.len = proof_lens;
    j.status = statuses; j.seconds = seconds;
    return pool->run(j);
    POOL_CATCH
}

}  // extern "C"
