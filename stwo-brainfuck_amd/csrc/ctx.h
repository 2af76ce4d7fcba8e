// Device context: stream, twiddle tree, arena allocator, small staging helpers. Internal to the library.
#pragma once
#include <cstdlib>
#include <thread>
#include <chrono>
#include "kernels.h"
#include "comm.h"
#include <memory>
#include <vector>
#include <string>
#include <stdexcept>
#include <cstring>

namespace bf {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
// A failed runtime call also leaves its code in the thread's "last error"; it is cleared here, or the next BF_HIP(hipGetLastError()) behind a perfectly
// good launch would report it (r06: an out-of-memory context creation made the NEXT creation on that thread fail with "out of memory").
#define BF_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { (void)hipGetLastError(); throw bf::HipError(std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)

// Bump allocator over large HBM chunks: the prover allocates hundreds of columns per proof and frees them all at once.
struct Arena {
    struct Chunk { char* base; size_t size, used; };
    std::vector<Chunk> chunks;
    size_t chunk_bytes = size_t(1) << 30;
    size_t total_used = 0, peak = 0;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        for (auto& c : chunks) if (c.size - c.used >= bytes) { void* p = c.base + c.used; c.used += bytes; total_used += bytes; if (total_used > peak) peak = total_used; return p; }
        size_t sz = bytes > chunk_bytes ? bytes : chunk_bytes;
        char* p = nullptr;
        BF_HIP(hipMalloc((void**)&p, sz));
        chunks.push_back({p, sz, bytes});
        total_used += bytes; if (total_used > peak) peak = total_used;
        return p;
    }
    void reset() { for (auto& c : chunks) c.used = 0; total_used = 0; }
    void release() { for (auto& c : chunks) (void)hipFree(c.base); chunks.clear(); total_used = 0; }
};

// One proof over several GPUs (include/bfhip.h: bfhip_ctx_join_*_group): this rank's place in the group and its transport (comm.h).
struct ShardGroup {
    u32 rank = 0, count = 1, log_count = 0;
    bool band_fusion = [] { const char* v = getenv("BFHIP_BAND_FUSION"); return !v || v[0] != '0'; }();   // A/B switch (every rank the same)
    std::shared_ptr<Comm> comm;
};

struct Ctx {
    int device = 0;
    Conventions conv;               // byte-level stwo conventions (bfhip_ctx_set_conventions)
    bool tables_on_gpu = true;      // where the 13 component tables are built (bfhip_ctx_set_table_builder)
    ShardGroup shard;
    int shard_policy = -1;          // bfhip_ctx_set_shard_policy: -1 automatic, 0 exchange columns -> rows (column-sharded transforms), 1 replicate the transforms
    bool shard_replicate = false;   // the decision for the proof in progress (HipProver::prove)
    hipStream_t stream = nullptr;
    // side stream: the trace-independent preprocessed commitment runs here, beside the main-trace phase. Created by the first proof that uses it
    // (ensure_side): a pool worker whose proofs take the pool's shared preprocessed tree never does, and every stream a process creates takes a
    // share of the few hardware queues (see aux below)
    hipStream_t stream2 = nullptr;
    void ensure_side();
    bool side_busy = false;         // work enqueued on stream2 may still read parameter blocks from the staging ring
    // Partner streams for overlap INSIDE a tree commitment: the VALU-bound Merkle layers of the largest columns run on the partner while the
    // HBM-bound transforms of the smaller columns continue on the stream itself (aux[0] beside the main stream, aux[1] beside the side stream).
    // Ordered by events only; both are joined back (event wait) before the commitment returns.
    // overlap: bit 0 = hash a tree's largest layers beside the transforms of its smaller columns; bit 1 = hash the FRI first-layer tree level by
    // level behind the quotient launches; bit 2 (shard groups) = the send-receive of a tree's largest size class on aux[0] beside the transforms of
    // its smaller columns (bfhip_ctx_set_overlap; BFHIP_OVERLAP presets it at context creation for A/B runs)
    // BFHIP_SINGLE_STREAM=1 (at context creation): the preprocessed phase stays on the main stream — ONE stream per proof, for profiler runs whose
    // per-kernel durations must not depend on how two streams share the GPU under the profiler (tools/profile_round.sh roofline). Same bytes.
    bool single_stream = false;
    bool overlap_user_set = false;   // bfhip_ctx_set_overlap / BFHIP_OVERLAP decided the mask: no default is applied on top of it
    // bit 2 is ON BY DEFAULT for a shard group whose ranks sit on different GPUs (exchange_overlapped()): there an exchange is an xGMI transfer
    // that costs the stream nothing but waiting; on one shared GPU it is a copy competing for the same HBM (r03: 38.5 vs 38.8 ms) and stays off
    bool exchange_overlapped() const { return (overlap & 4u) != 0 || (!overlap_user_set && shard.count > 1 && shard.comm && shard.comm->spans_devices()); }
    u32 overlap = 0;      // measured (profiles/r03_overlap_ab*.txt): bit 1 gains 0-0.3 ms on fib19 box to box, bit 0 nothing — both sides of either overlap are
                          // VALU-limited (co-running kernels stretch each other), and the dominant kernel's event-timed roofline would include the interference
    // Created on demand (r05): HIP hands every stream one of a few hardware queues (GPU_MAX_HW_QUEUES, default 4) when it is created; with four
    // streams per context the main streams of two contexts shared a queue and two proofs in flight serialised (tools/inflight_history.py).
    hipStream_t aux[2] = {nullptr, nullptr};
    void ensure_aux();
    hipStream_t id_main = nullptr;  // the main stream's handle (stream and stream2 are swapped while the preprocessed phase is enqueued)
    hipStream_t aux_of(hipStream_t s) const { return s == id_main ? aux[0] : aux[1]; }
    hipEvent_t evp[32] = {};        // ordering events (no timing), handed out round robin: a wait captures the event's state when it is enqueued
    u32 evp_next = 0;
    hipEvent_t next_event() { return evp[evp_next++ % 32]; }
    // Ticket counters of the kernels whose last workgroup finishes a tree (merkle.hip: k_merkle_small_end, k_fri_layer): one zeroed word per
    // stream that may run such a kernel (main, side and the two partners), reset by the kernels themselves.
    u32* d_counters = nullptr;
    u32* merkle_counter() { return d_counters + 64 * (stream == id_main ? 0 : stream == aux[0] ? 1 : stream == aux[1] ? 2 : 3); }
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // phase boundaries (GPU-side phase times without extra host syncs)
    u32 tw_root_log = 0;           // twiddle tree rooted at Coset::half_odds(tw_root_log)
    u32* d_tw = nullptr; u32* d_itw = nullptr;
    // false: twiddle tree and point tables are another context's (the sub-contexts of a pool share the first one's: include/bfhip.h
    // bfhip_pool_create) — read-only after creation, never freed here
    bool owns_tables = true;
    // Set by a pool for the duration of a batch (pool.hip): the preprocessed tree every proof of the batch takes instead of committing its own.
    const struct SharedPreprocessed* shared_pre = nullptr;
    uint2* d_tlo = nullptr; uint2* d_thi = nullptr;   // G^a (a < 2^16) and G^(b << 16) (b < 2^15) point tables
    Arena arena;
    // pinned staging for pointer arrays / small parameter blocks
    // Pinned scratch for device->host results. [0, 4096): fixed slots for deferred tiny results (tree roots, FRI roots, channel
    // state); [4096, h_small_bytes): bounce buffer of read_back(). A pageable destination would make every such copy a blocking,
    // internally staged transfer.
    char* h_small = nullptr; size_t h_small_bytes = 256 << 10;
    // The same pinned memory as the device sees it (hipHostGetDevicePointer): kernels write small results — roots, claimed sums, sampled
    // values, decommitment words — straight into it, and the gather kernel reads its request list from the staging ring's host side. A
    // result that crosses PCIe by the kernel's own stores needs no copy command behind the kernel (r04: ~8 us per Fiat-Shamir round trip:
    // a blit dispatch and its barrier); the host reads it once the event recorded behind the kernel has completed (system-scope release).
    char* d_small_alias = nullptr; char* d_hstage_alias = nullptr;
    template <class T> T* small_alias(T* host_ptr) const { return reinterpret_cast<T*>(d_small_alias + (reinterpret_cast<char*>(host_ptr) - h_small)); }
    char* h_stage = nullptr; char* d_stage = nullptr; size_t stage_bytes = 8 << 20, stage_used = 0;
    // ---- mailboxes and stamps (mailbox.hip, r04): the launches behind a Fiat-Shamir point are enqueued before the host knows the challenge ----
    u32 last_proof_flags = 0;         // bfhip_ctx_last_proof_flags: what the last completed proof of this context did
    u32 proof_seq = 0;                // flags and stamps of a proof carry its number (never 0), so a slot needs no reset between proofs
    // Measured over three boxes the order gains 1-3 % on 2^20-row proofs and nothing (fib19: +0.1..+0.5 %) on large ones, where the saved
    // idle time is below the noise of a VALU-limited proof (finding iii). So by default it is used for proofs with
    // LOG_MAX_ROWS <= 21 only (mailbox_mode -1, decided per proof in HipProver::prove); BFHIP_MAILBOX=1 / 0 forces it on / off.
    int mailbox_mode = -1;
    bool use_mailbox = false;         // the decision for the proof in progress
    int mailboxes_pending = 0;        // armed and not yet posted: the stream must not be waited for as a whole
    double mailbox_timeout = 10.0;    // seconds a mailbox kernel waits for the host before it gives up (BFHIP_MAILBOX_TIMEOUT_MS)
    int mailbox_test_delay_ms = 0;    // tests only (BFHIP_MAILBOX_TEST_DELAY_MS): the host sleeps this long before every post — a late host
    // pinned slots inside h_small's fixed area: flags (host writes, a kernel polls) and stamps (a kernel writes, the host polls), 64 bytes apart
    u32* flag_host(int k) const { return reinterpret_cast<u32*>(h_small + 3072 + 64 * k); }
    u32* stamp_host(int k) const { return reinterpret_cast<u32*>(h_small + 3584 + 64 * k); }
    u32* mailbox_err_host() const { return reinterpret_cast<u32*>(h_small + 3520); }   // behind the six flag slots in use; stamp 7 ends at 4096
    // Reaping: the runtime releases the bookkeeping of completed launches when the host asks about an event behind them. With stamps the host
    // never asks, ~170 launches pile up per proof and are then released by the runtime's own handler thread at the worst moment — concurrently
    // with the host's decommitment planning (measured: planning 36 -> 95 us at 2^20 rows). So the enqueue code drops events along the stream
    // (reap_point) and every host wait queries them in order while it has nothing else to do (reap_some).
    hipEvent_t reap_ev[64] = {}; u32 reap_head = 0, reap_tail = 0;        // FIFO over a ring of events; head == tail: empty
    void reap_point() {
        if (!use_mailbox || reap_tail - reap_head >= 64 || !reap_ev[0]) return;
        if (hipEventRecord(reap_ev[reap_tail % 64], stream) == hipSuccess) reap_tail++;
    }
    void reap_some() { while (reap_head != reap_tail && hipEventQuery(reap_ev[reap_head % 64]) == hipSuccess) reap_head++; }
    void reap_reset() { reap_head = reap_tail = 0; }
    void post_stamp(int k);           // enqueue: "everything before this point on the stream is done" -> stamp k = proof_seq
    void wait_stamp(int k);           // host: poll stamp k (with the stream's health checked now and then)

    // tables_from != nullptr: a context on the same device whose twiddle tree (at least as large) and point tables this one borrows
    void init(int dev, u32 max_log_domain, const Ctx* tables_from = nullptr);
    void destroy();                 // idempotent: also what the destructor and a failed init() run
    Ctx() = default;
    Ctx(const Ctx&) = delete; Ctx& operator=(const Ctx&) = delete;
    ~Ctx() { destroy(); }
    // HIP's current device is per host thread: every C-ABI entry binds the calling thread to this context's GPU first, so that
    // allocations (arena chunks, hipMalloc) land on the device the stream belongs to whichever thread drives the context.
    void bind() { BF_HIP(hipSetDevice(device)); }
    // Host waits for the stream. A proof has ~10 of these on its critical path (roots, samples, nonce: the Fiat-Shamir points), each followed
    // by a few microseconds of host work and the next launches, so the wake-up latency of a blocking wait is paid ~10 times per proof:
    // poll an event instead (BFHIP_SYNC=block restores hipStreamSynchronize).
    hipEvent_t sync_ev = nullptr;
    // Waits: a few tens of microseconds of polling (the Fiat-Shamir round trips are that short: no wake-up latency), then yielding polls;
    // `sync_blocking` (bfhip_ctx_set_sync_policy, or BFHIP_SYNC=block at context creation) goes to hipStreamSynchronize at once — for hosts
    // with more waiting contexts than cores. Inside a shard group every wait is BOUNDED: the transport is polled for asynchronous errors and
    // after comm_timeout_seconds() the group is aborted and the wait fails, instead of a stream that hangs on a peer that diverged.
    bool sync_blocking = false;
    // how long a wait polls before it goes to sleep on the blocking event. Outside a proof 200 us; INSIDE a proof every wait is a Fiat-Shamir
    // round trip with the GPU idle behind it, and the wake-up of a sleeping thread costs 10-15 us each time: the prover raises the limit to
    // 8 ms for its duration (r04; one core spins while a proof is in flight, which is what a host thread per GPU is there for)
    double spin_seconds = 200e-6;
    hipEvent_t block_ev = nullptr;   // hipEventBlockingSync: the only event a host thread SLEEPS on (sync_ev is polled)
    void sync() {
        const bool grouped = shard.count > 1 && shard.comm;
        if ((sync_blocking && !grouped) || !sync_ev) { BF_HIP(hipStreamSynchronize(stream)); return; }
        BF_HIP(hipEventRecord(sync_ev, stream));
        const auto t0 = std::chrono::steady_clock::now();
        for (u32 polls = 0;; polls++) {
            hipError_t e = hipEventQuery(sync_ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) BF_HIP(e);
            if (polls < 64) continue;
            if ((polls & 255u) == 0) {
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (grouped) {
                    // Inside a group the wait stays a bounded poll (the transport must be asked for asynchronous errors and a peer that
                    // diverged must not hang this rank), so set_sync_policy(blocking) does not apply; once the wait is longer than any
                    // Fiat-Shamir round trip of a sharded proof (4 ms) it backs off with short sleeps instead of spinning — N ranks of one
                    // process on an oversubscribed host otherwise starve each other. (Shorter waits keep polling: a sleep's wake-up
                    // latency, ~60 us, would be paid ten times per proof.)
                    try { shard.comm->check_async(); } catch (...) { shard.comm->abort(); throw; }
                    if (waited > comm_timeout_seconds()) { shard.comm->abort(); throw HipError("shard group: the stream did not complete within the communication timeout (a peer failed or diverged)"); }
                    if (waited > 4e-3) { std::this_thread::sleep_for(std::chrono::microseconds(waited > 50e-3 ? 200 : 50)); continue; }
                } else if (waited > spin_seconds) {
                    // a long wait (a whole proof, a big trace): stop burning a core — sleep on an event created for blocking waits
                    // (an event without hipEventBlockingSync may be waited for by spinning inside the runtime)
                    if (block_ev) { BF_HIP(hipEventRecord(block_ev, stream)); BF_HIP(hipEventSynchronize(block_ev)); }
                    else BF_HIP(hipStreamSynchronize(stream));
                    return;
                }
            }
            std::this_thread::yield();      // several contexts may be waiting on as many host threads
        }
    }
    // Copy a small host block to device scratch (valid until the ring is recycled by stage_checkpoint()). Stream-ordered.
    // Between stage_begin() and stage_end() the blocks are only written to the pinned ring and ONE copy moves them all at
    // stage_end(): every separate copy is a ~5 us blit on the GPU timeline, and a proof stages ~150 blocks.
    int stage_batch_depth = 0; size_t stage_batch_lo = 0;
    void stage_begin() { if (stage_batch_depth++ == 0) stage_batch_lo = stage_used; }
    void stage_end() {
        if (--stage_batch_depth == 0 && stage_used > stage_batch_lo)
            BF_HIP(hipMemcpyAsync(d_stage + stage_batch_lo, h_stage + stage_batch_lo, stage_used - stage_batch_lo, hipMemcpyHostToDevice, stream));
    }
    template <class T>
    T* stage(const T* host, size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        if (stage_used + bytes > stage_bytes) throw HipError("staging buffer exhausted (call stage_checkpoint() between operations)");
        memcpy(h_stage + stage_used, host, n * sizeof(T));
        if (stage_batch_depth == 0) BF_HIP(hipMemcpyAsync(d_stage + stage_used, h_stage + stage_used, n * sizeof(T), hipMemcpyHostToDevice, stream));
        T* r = reinterpret_cast<T*>(d_stage + stage_used);
        stage_used += bytes;
        return r;
    }
    // Called at the start of every high-level operation: recycles the staging ring once it is half full (after a sync, so no
    // in-flight kernel still reads parameter blocks from it).
    void stage_checkpoint() {
        if (stage_batch_depth == 0 && stage_used > stage_bytes / 2) {
            if (mailboxes_pending) {      // a mailbox kernel is waiting for this very thread: waiting for the stream would never return
                if (stage_used > stage_bytes - (size_t(1) << 20)) throw HipError("staging buffer exhausted while a mailbox is pending");
                return;
            }
            // every stream of the context may still read parameter blocks from the ring (`stream` may currently be a partner stream)
            sync();
            if (stream2) BF_HIP(hipStreamSynchronize(stream2));
            if (id_main) BF_HIP(hipStreamSynchronize(id_main));
            for (auto a : aux) if (a) BF_HIP(hipStreamSynchronize(a));
            stage_used = 0;
        }
    }
    u32* alloc_u32(size_t n) { return (u32*)arena.alloc(n * sizeof(u32)); }
    // Stream-ordered device -> host read of a small result through the pinned bounce buffer; returns after the data has arrived.
    void read_back(void* dst_h, const void* src_d, size_t bytes) {
        const size_t cap = h_small_bytes - 4096;
        for (size_t o = 0; o < bytes; o += cap) {
            size_t n = bytes - o < cap ? bytes - o : cap;
            BF_HIP(hipMemcpyAsync(h_small + 4096, (const char*)src_d + o, n, hipMemcpyDeviceToHost, stream));
            sync();
            memcpy((char*)dst_h + o, h_small + 4096, n);
        }
    }
};

// Scope of one staging batch: blocks staged inside are moved by one copy at end(); an exception unwinds the batch without copying.
void preprocessed_cache_invalidate(Ctx* c);   // prover.hip: called when the context joins or leaves a shard group
// prover.hip, for pool.hip: the preprocessed tree a pool's builder context commits once for all proofs of a batch (Ctx::shared_pre)
struct SharedPreprocessed;
SharedPreprocessed* shared_preprocessed_create(Ctx& builder);
void shared_preprocessed_destroy(SharedPreprocessed* sp);
bool shared_preprocessed_matches(const SharedPreprocessed* sp, const Ctx& c, u32 log_max_rows);
void shared_preprocessed_build(SharedPreprocessed* sp, Ctx& builder, u32 log_max_rows);      // enqueues, records sp->ready, returns without waiting
void shared_preprocessed_invalidate(SharedPreprocessed* sp);

void mailbox_launch(hipStream_t s, const u32* d_flag, u32 expect, const void* src_pinned_alias, void* dst, size_t bytes, u32* d_err, double timeout_seconds);
void post_stamp_launch(hipStream_t s, u32* d_stamp, u32 value);
inline void Ctx::post_stamp(int k) { post_stamp_launch(stream, small_alias(stamp_host(k)), proof_seq); }
inline void Ctx::wait_stamp(int k) {
    u32* p = stamp_host(k);
    const auto t0 = std::chrono::steady_clock::now();
    double next_check = 50e-3;
    for (u32 polls = 1;; polls++) {
        if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == proof_seq) return;
        if ((polls & 1023u) == 0) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            // A queue that faulted never writes the stamp: ask the runtime — but rarely: a stream query makes the runtime put a barrier packet
            // with a completion signal into the queue (launches carry none), in the middle of whatever latency chain is running. A drained
            // stream without the stamp is a bug, not a wait.
            if (waited > next_check) {
                next_check = waited + 50e-3;
                hipError_t e = hipStreamQuery(stream);
                if (e == hipSuccess) { if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == proof_seq) return; throw HipError("the stream drained without writing the stamp the host waits for"); }
                if (e != hipErrorNotReady) BF_HIP(e);
            }
            // past the polling budget (200 us under the blocking sync policy, 8 ms inside a proof otherwise): sleep between looks
            if (waited > spin_seconds) std::this_thread::sleep_for(std::chrono::microseconds(sync_blocking ? 200 : 50));
        }
    }
}

// One staging batch whose copy is made by a mailbox kernel (mailbox.hip): begin(), stage the phase's parameter blocks (structurally complete,
// challenge words still empty), arm() — enqueues the kernel; the phase's launches follow it on the stream —, later patch the blocks in the
// ring through host() and post(). Going out of scope posts (an abandoned proof must not leave the queue waiting).
struct Mailbox {
    Ctx& c; int slot; size_t lo = 0, hi = 0; bool open = false, armed = false, posted = false;
    Mailbox(Ctx& c_, int slot_) : c(c_), slot(slot_) {}
    Mailbox(const Mailbox&) = delete; Mailbox& operator=(const Mailbox&) = delete;
    void begin() {
        if (c.stage_batch_depth != 0) throw HipError("mailbox: opened inside a staging batch");
        c.stage_begin(); open = true; lo = c.stage_used;
    }
    void arm() {
        hi = c.stage_used; open = false; c.stage_batch_depth = 0;
        // err word of this slot: [0] gave up, [1] ticks waited — slot k reports at mailbox_err_host() + 2 k (slot 0 is the proof's error word)
        mailbox_launch(c.stream, c.small_alias(c.flag_host(slot)), c.proof_seq, c.d_hstage_alias + lo, c.d_stage + lo, hi - lo, c.small_alias(c.mailbox_err_host()) + 2 * slot, c.mailbox_timeout);
        armed = true; c.mailboxes_pending++;
    }
    template <class T> T* host(const T* dev) const { return reinterpret_cast<T*>(c.h_stage + (reinterpret_cast<const char*>(dev) - c.d_stage)); }
    void post() { if (armed && !posted) { if (c.mailbox_test_delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(c.mailbox_test_delay_ms)); __atomic_store_n(c.flag_host(slot), c.proof_seq, __ATOMIC_RELEASE); posted = true; c.mailboxes_pending--; } }
    ~Mailbox() { if (open) c.stage_batch_depth = 0; post(); }
};

struct StageBatch {
    Ctx& c; bool open = true;
    explicit StageBatch(Ctx& c_) : c(c_) { c.stage_begin(); }
    void end() { if (open) { open = false; c.stage_end(); } }
    ~StageBatch() { if (open) { if (--c.stage_batch_depth < 0) c.stage_batch_depth = 0; } }
};

}  // namespace bf

struct bfhip_ctx { bf::Ctx c; };
void bfhip_set_error(const std::string& s);
