// Shard-group transports (comm.h): RCCL over xGMI for one process per GPU, and an in-process transport for N contexts of one process
// (on one GPU — how a 1-GPU box tests the group logic — or on one GPU each: N host threads driving N GPUs).
#include "ctx.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <algorithm>

namespace bf {

double comm_timeout_seconds() {
    static const double t = [] { const char* v = getenv("BFHIP_COMM_TIMEOUT_S"); double x = v ? atof(v) : 0.0; return x > 0.0 ? x : 300.0; }();
    return t;
}
Comm::~Comm() { for (auto& t : pending_) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); } for (auto e : pool_) (void)hipEventDestroy(e); }
hipEvent_t Comm::timing_event() {
    if (!pool_.empty()) { hipEvent_t e = pool_.back(); pool_.pop_back(); return e; }
    hipEvent_t e = nullptr;
    BF_HIP(hipEventCreate(&e));
    return e;
}
// Completed pairs at the front of the list are folded into the totals and their events recycled, so a group that proves for ever without
// anybody asking for bfhip_ctx_group_times holds only the pairs still in flight (a proof's worth at most) and, once warm, creates no event.
void Comm::drain_completed() {
    size_t done = 0;
    while (done < pending_.size() && hipEventQuery(pending_[done].e1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pending_[done].e0, pending_[done].e1) == hipSuccess) {
            ms_[pending_[done].kind] += ms;
            auto& v = gpu_us_[pending_[done].kind];
            if (v.size() >= MAX_LATENCIES) v.erase(v.begin(), v.begin() + (long)(MAX_LATENCIES / 2));
            v.push_back(ms * 1e3f);
        }
        pool_.push_back(pending_[done].e0); pool_.push_back(pending_[done].e1);
        done++;
    }
    (void)hipGetLastError();          // hipErrorNotReady of the first unfinished pair is not an error
    if (done) pending_.erase(pending_.begin(), pending_.begin() + (long)done);
    // a stream that is never synchronised between collectives (not how the prover runs) must not grow the list without bound either
    while (pending_.size() > MAX_PENDING) { pool_.push_back(pending_.front().e0); pool_.push_back(pending_.front().e1); pending_.erase(pending_.begin()); }
}
static double host_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
Comm::Timed::Timed(Comm& c_, hipStream_t s_, int kind_) : c(c_), s(s_), kind(kind_), t0(host_now_us()) {
    if (!s) return;
    c.drain_completed();
    hipEvent_t e0 = c.timing_event(), e1_ = nullptr;
    try { e1_ = c.timing_event(); BF_HIP(hipEventRecord(e0, s)); }
    catch (...) { c.pool_.push_back(e0); if (e1_) c.pool_.push_back(e1_); throw; }
    e1 = e1_;
    c.pending_.push_back({e0, e1, kind});
}
Comm::Timed::~Timed() {
    if (e1) (void)hipEventRecord(e1, s);
    auto& v = c.host_us_[kind];
    if (v.size() >= MAX_LATENCIES) v.erase(v.begin(), v.begin() + (long)(MAX_LATENCIES / 2));
    v.push_back((float)(host_now_us() - t0));
}
void Comm::times_ms(double out[3]) {
    for (auto& t : pending_) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            ms_[t.kind] += ms;
            auto& v = gpu_us_[t.kind];
            if (v.size() >= MAX_LATENCIES) v.erase(v.begin(), v.begin() + (long)(MAX_LATENCIES / 2));
            v.push_back(ms * 1e3f);
        }
        pool_.push_back(t.e0); pool_.push_back(t.e1);
    }
    pending_.clear();
    for (int k = 0; k < 3; k++) out[k] = ms_[k];
}
void Comm::latency_us(double out[3][7], bool reset) {
    double tmp[3];
    times_ms(tmp);                       // folds the pairs still pending (the stream has been synchronised)
    auto stats = [](std::vector<float> v, double* o) {      // by value: sorted copy
        if (v.empty()) { o[0] = o[1] = o[2] = 0.0; return; }
        std::sort(v.begin(), v.end());
        o[0] = v[v.size() / 2]; o[1] = v[(v.size() * 9) / 10 < v.size() ? (v.size() * 9) / 10 : v.size() - 1]; o[2] = v.back();
    };
    for (int k = 0; k < 3; k++) {
        out[k][0] = (double)gpu_us_[k].size();
        stats(gpu_us_[k], &out[k][1]);
        stats(host_us_[k], &out[k][4]);
        if (reset) { gpu_us_[k].clear(); host_us_[k].clear(); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// LocalComm: the ranks are host threads of one process; their contexts may share a device or own one each. A transfer is a
// device-to-device copy enqueued on the RECEIVER's stream after the sender's "ready" event (events and copies work across devices; peer
// access is enabled on first use so that the copies go over xGMI instead of through the host); the sender's stream then waits for the
// receivers' "done" events before it may touch the source again. A host rendezvous (barrier) separates publishing the pointers from
// reading them. No kernel reads another rank's memory.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct LocalGroup {
    u32 count;
    std::mutex mu; std::condition_variable cv; u32 arrived = 0; std::atomic<u64> generation{0};
    std::atomic<bool> failed{false};     // a member failed or left mid-collective: every rendezvous throws from then on
    std::vector<bool> taken;             // ranks currently held by a LocalComm
    struct Slot { void* buf = nullptr; hipEvent_t ready = nullptr, done = nullptr; std::vector<Xfer> sends; int device = -1; };
    std::vector<Slot> slots;
    // Whether the members sit on more than one GPU. Group-wide state (not per member): the prover picks its collective sequence by it
    // (Ctx::exchange_overlapped), so every member — also one that joins a rank another context left — must read the same value. Recomputed
    // from the slots' devices by every member after the first rendezvous it takes part in (all slots of live members are filled by then).
    std::atomic<int> spans{-1};          // -1 unknown (no collective yet), 0 one GPU, 1 several
    std::atomic<u32> epoch{0};           // bumped by every join: members re-check their peers (and peer access) when the membership changed
    explicit LocalGroup(u32 n) : count(n), taken(n, false), slots(n) {}
    void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
    // The ranks of a proof reach a rendezvous within tens of microseconds of each other ~60 times per proof, and a thread that sleeps on the
    // condition variable pays a futex wake-up (50-100 us, the woken threads then queue for the mutex) every time. So a waiting rank first
    // polls the generation counter for up to 300 us and only then goes to sleep (r04; on ONE time-shared GPU it changes nothing measurable —
    // 67.7 ms for 8 ranks either way —, the GPUs of a multi-device group are what would otherwise wait for the wake-ups).
    void barrier() {
        u64 gen;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (failed) throw HipError("shard group: another rank of the group failed");
            gen = generation.load(std::memory_order_relaxed);
            if (++arrived == count) { arrived = 0; generation.store(gen + 1, std::memory_order_release); cv.notify_all(); return; }
        }
        {
            const auto t0 = std::chrono::steady_clock::now();
            for (u32 polls = 1;; polls++) {
                if (generation.load(std::memory_order_acquire) != gen) return;
                if (failed.load(std::memory_order_relaxed)) break;
                if ((polls & 63u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300e-6) break;
                __builtin_ia32_pause();
            }
        }
        std::unique_lock<std::mutex> lk(mu);
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(comm_timeout_seconds()), [&] { return generation.load() != gen || failed.load(); });
        if (generation != gen) return;                 // released (even if the group failed right afterwards: the next rendezvous reports it)
        arrived--;                                     // this rank leaves the rendezvous it did not complete
        if (!ok) { failed = true; cv.notify_all(); throw HipError("shard group: a rank did not reach the rendezvous (another rank failed or diverged)"); }
        throw HipError("shard group: another rank of the group failed");
    }
};
std::shared_ptr<LocalGroup> local_group_create(u32 count) { return std::make_shared<LocalGroup>(count); }

// Transfers between contexts that share ONE device (how a one-GPU box runs a group, and what tools/shard_local.py times) go through one
// launch per collective instead of one runtime copy per block: a proof issues ~280 blocks per rank, and every hipMemcpyAsync is a blit
// dispatch behind a cache write-back with ~15-20 us of idle GPU in front of it (r04: 24 of the 81 ms of an 8-rank fib19 proof on one GPU
// were such gaps, profiles/r04_shard_timeline_before.txt) — latency of the stand-in transport, not of the proof.
struct CopyBlock { const void* src; void* dst; unsigned long long bytes; };
static constexpr u32 COPY_BATCH = 64, COPY_CHUNK_LOG = 16;      // <= 64 blocks per launch (1.8 KiB of kernel arguments), 64 KiB per workgroup
struct CopyBatch { CopyBlock b[COPY_BATCH]; u32 chunk0[COPY_BATCH + 1]; u32 n; };
__global__ void __launch_bounds__(256) k_copy_blocks(const CopyBatch a) {
    u32 k = 0;
    while (k + 1 < a.n && a.chunk0[k + 1] <= blockIdx.x) k++;           // uniform: scalar loads from the kernel arguments
    const unsigned long long off = (unsigned long long)(blockIdx.x - a.chunk0[k]) << COPY_CHUNK_LOG;
    const unsigned long long left = a.b[k].bytes - off;
    const u32 len = left < (1ull << COPY_CHUNK_LOG) ? (u32)left : (1u << COPY_CHUNK_LOG);
    const char* src = (const char*)a.b[k].src + off;
    char* dst = (char*)a.b[k].dst + off;
    if ((((unsigned long long)src | (unsigned long long)dst | len) & 15ull) == 0) {
        for (u32 i = threadIdx.x * 16; i < len; i += 256 * 16) *(uint4*)(dst + i) = *(const uint4*)(src + i);
    } else {
        for (u32 i = threadIdx.x * 4; i < len; i += 256 * 4) *(u32*)(dst + i) = *(const u32*)(src + i);      // every block is a whole number of words
    }
}
// out[i] = max over the ranks' buffers, read in place (same device)
struct MaxSources { const u32* p[64]; u32 n; };
__global__ void __launch_bounds__(256) k_max_u32_direct(u32* __restrict__ out, const MaxSources srcs, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 m = 0;
    for (u32 b = 0; b < srcs.n; b++) { u32 v = srcs.p[b][i]; m = v > m ? v : m; }
    out[i] = m;
}
// out[i] = max over the nbufs gathered copies (copy b at gathered + b * n)
__global__ void k_max_u32_n(u32* __restrict__ out, const u32* __restrict__ gathered, u32 nbufs, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 m = 0;
    for (u32 b = 0; b < nbufs; b++) { u32 v = gathered[b * n + i]; m = v > m ? v : m; }
    out[i] = m;
}

struct LocalComm : Comm {
    std::shared_ptr<LocalGroup> g;
    u32* scratch = nullptr; size_t scratch_words = 0;     // (count + 1) * n words: the gathered copies, then the result
    bool multi_device = false; u32 checked_epoch = ~0u;
    LocalComm(const std::shared_ptr<LocalGroup>& g_, u32 r) : g(g_) {
        rank = r; count = g_->count;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (g->taken[r]) throw HipError("shard group: this rank of the group is already taken by another context");
            g->taken[r] = true;
        }
        int dev = -1;
        BF_HIP(hipGetDevice(&dev));                       // the C-ABI entry has bound this thread to the context's GPU
        {
            // Whether the group spans several GPUs is decided at a RENDEZVOUS, where every member is present and takes the same decision
            // (check_peers, behind the first barrier) — not here: a member that joined early and started its first proof would read "one
            // device" while a later joiner on another GPU read "several", and the two would issue different collective sequences for the
            // first tree (one exchange against first_wave + second_wave: 'unmatched send/receive'). Until that rendezvous every
            // member answers spans_devices() == false. The slot is written under the lock the peers read it under.
            std::lock_guard<std::mutex> lk(g->mu);
            g->slots[r].device = dev;
            g->epoch.fetch_add(1, std::memory_order_release);
        }
        BF_HIP(hipEventCreateWithFlags(&g->slots[r].ready, hipEventDisableTiming));
        BF_HIP(hipEventCreateWithFlags(&g->slots[r].done, hipEventDisableTiming));
    }
    ~LocalComm() override {
        { std::lock_guard<std::mutex> lk(g->mu); g->taken[rank] = false; }
        (void)hipEventDestroy(g->slots[rank].ready); (void)hipEventDestroy(g->slots[rank].done);
        g->slots[rank].ready = g->slots[rank].done = nullptr;
        (void)hipFree(scratch);
    }
    void abort() override { g->fail(); }
    // the group's answer once a rendezvous has decided it (the same for every member); false before the first collective
    bool spans_devices() const override { return g->spans.load(std::memory_order_acquire) > 0; }
    const char* transport() const override {
        return multi_device ? "local (N contexts of one process on one GPU each, peer copies ordered by HIP events)"
                            : "local (N contexts of one process, device-to-device copies ordered by HIP events)";
    }
    // first collective (every rank has joined by its rendezvous): best-effort peer access to the other ranks' GPUs
    void check_peers() {
        const u32 e = g->epoch.load(std::memory_order_acquire);
        if (checked_epoch == e) return;
        checked_epoch = e;
        std::lock_guard<std::mutex> lk(g->mu);             // the slots' devices are written under this lock (constructor)
        const int mine = g->slots[rank].device;
        for (u32 p = 0; p < count; p++) {
            const int d = g->slots[p].device;
            if (d == mine || d < 0) continue;
            multi_device = true;
            g->spans.store(1, std::memory_order_release);
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, mine, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);   // "already enabled" is fine
            (void)hipGetLastError();
        }
        if (!multi_device) { int unknown = -1; g->spans.compare_exchange_strong(unknown, 0); }
    }
    // Blocks to copy from other ranks' memory into mine on my stream. One device for the whole group: one launch per <= 64 blocks
    // (k_copy_blocks); several devices: a runtime peer copy per block.
    bool same_device() const { for (u32 p = 0; p < count; p++) if (g->slots[p].device != g->slots[rank].device) return false; return true; }
    struct Pending { u32 peer; void* dst; const void* src; size_t bytes; };
    void run_copies(const std::vector<Pending>& blocks, hipStream_t s) {
        bool words = true;
        for (auto& b : blocks) words = words && b.bytes % 4 == 0 && ((uintptr_t)b.dst | (uintptr_t)b.src) % 4 == 0;
        if (!same_device() || !words) {
            const int mine = g->slots[rank].device;
            for (auto& b : blocks) {
                if (!b.bytes) continue;
                const int theirs = g->slots[b.peer].device;
                if (theirs != mine) BF_HIP(hipMemcpyPeerAsync(b.dst, mine, b.src, theirs, b.bytes, s));
                else BF_HIP(hipMemcpyAsync(b.dst, b.src, b.bytes, hipMemcpyDeviceToDevice, s));
            }
            return;
        }
        CopyBatch cb{};
        u32 chunks = 0;
        auto flush = [&]() {
            if (!cb.n) return;
            cb.chunk0[cb.n] = chunks;
            hipLaunchKernelGGL(k_copy_blocks, dim3(chunks), dim3(256), 0, s, cb);
            cb.n = 0; chunks = 0;
        };
        for (auto& b : blocks) {
            if (!b.bytes) continue;
            if (b.bytes >> (COPY_CHUNK_LOG + 30)) throw HipError("shard group: block too large");
            const u64 need = (u64)((b.bytes + (size_t(1) << COPY_CHUNK_LOG) - 1) >> COPY_CHUNK_LOG);     // < 2^30 by the check above
            if ((u64)chunks + need > 0x7fffffffull) flush();       // the grid (and chunk0) count chunks in 31 bits: start a new launch first
            cb.b[cb.n] = CopyBlock{b.src, b.dst, (unsigned long long)b.bytes};
            cb.chunk0[cb.n] = chunks;
            chunks += (u32)need;
            if (++cb.n == COPY_BATCH) flush();
        }
        flush();
        BF_HIP(hipGetLastError());
    }
    void wait_ready(hipStream_t s, u32 p) { if (p != rank) BF_HIP(hipStreamWaitEvent(s, g->slots[p].ready, 0)); }
    void publish(hipStream_t s, void* buf) { g->slots[rank].buf = buf; BF_HIP(hipEventRecord(g->slots[rank].ready, s)); g->barrier(); check_peers(); }
    void finish(hipStream_t s) {
        BF_HIP(hipEventRecord(g->slots[rank].done, s));
        g->barrier();
        for (u32 p = 0; p < count; p++) if (p != rank) BF_HIP(hipStreamWaitEvent(s, g->slots[p].done, 0));   // peers have read my buffer
    }
    void all_gather(hipStream_t s, void* buf, size_t bpr) override {
        n_all_gather++; bytes_sent += bpr * (count - 1);
        Timed tm(*this, s, T_ALL_GATHER);
        publish(s, buf);
        std::vector<Pending> blocks;
        for (u32 p = 0; p < count; p++) {
            if (p == rank) continue;
            wait_ready(s, p);
            blocks.push_back({p, (char*)buf + p * bpr, (const char*)g->slots[p].buf + p * bpr, bpr});
        }
        run_copies(blocks, s);
        finish(s);
    }
    void all_reduce_max_u32(hipStream_t s, u32* buf, size_t n) override {
        n_all_reduce++; bytes_sent += n * sizeof(u32);
        if (n == 0) { g->barrier(); g->barrier(); return; }
        Timed tm(*this, s, T_ALL_REDUCE);
        const bool direct = same_device() && count <= 64;        // one device: the maximum is taken over the ranks' buffers where they lie
        const size_t need = direct ? n : (size_t)(count + 1) * n;
        if (scratch_words < need) { BF_HIP(hipStreamSynchronize(s)); (void)hipFree(scratch); scratch = nullptr; scratch_words = 0; BF_HIP(hipMalloc((void**)&scratch, need * sizeof(u32))); scratch_words = need; }
        publish(s, buf);
        for (u32 p = 0; p < count; p++) wait_ready(s, p);
        u32* result;
        if (direct) {
            result = scratch;
            MaxSources ms{};
            ms.n = count;
            for (u32 p = 0; p < count; p++) ms.p[p] = (const u32*)g->slots[p].buf;
            hipLaunchKernelGGL(k_max_u32_direct, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, result, ms, n);
        } else {
            std::vector<Pending> blocks;       // gather every rank's (unmodified) input into this rank's scratch, then reduce locally
            for (u32 p = 0; p < count; p++) blocks.push_back({p, scratch + (size_t)p * n, g->slots[p].buf, n * sizeof(u32)});
            run_copies(blocks, s);
            result = scratch + (size_t)count * n;
            hipLaunchKernelGGL(k_max_u32_n, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, result, scratch, count, n);
        }
        finish(s);                         // every rank has read the unmodified inputs
        run_copies({Pending{rank, buf, result, n * sizeof(u32)}}, s);
    }
    void exchange(hipStream_t s, const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs) override {
        n_exchange++; for (auto& x : sends) if (x.peer != rank) bytes_sent += x.bytes;
        Timed tm(*this, s, T_EXCHANGE);
        g->slots[rank].sends = sends;
        publish(s, nullptr);
        std::vector<size_t> next(count, 0);   // per peer: position in that peer's send list of the next block addressed to me
        std::vector<Pending> blocks;
        std::vector<bool> waited(count, false);
        for (const Xfer& r : recvs) {
            const std::vector<Xfer>& ps = g->slots[r.peer].sends;
            size_t& k = next[r.peer];
            while (k < ps.size() && ps[k].peer != rank) k++;
            if (k >= ps.size() || ps[k].bytes != r.bytes) { g->fail(); throw HipError("shard group: unmatched send/receive"); }
            if (!waited[r.peer]) { wait_ready(s, r.peer); waited[r.peer] = true; }
            if (r.bytes) blocks.push_back({r.peer, r.ptr, ps[k].ptr, r.bytes});
            k++;
        }
        run_copies(blocks, s);
        finish(s);
    }
};
std::unique_ptr<Comm> local_comm_join(const std::shared_ptr<LocalGroup>& g, u32 rank) {
    if (!g || rank >= g->count) throw HipError("shard group: bad rank");
    return std::unique_ptr<Comm>(new LocalComm(g, rank));
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// RcclComm: RCCL (backend "nccl" of torch.distributed is the same library) on the context's stream. Loaded lazily.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
#ifdef BFHIP_TEST_HOOKS
    // libbfhip_testhooks.so only. A test double exports this (tests/mock_rccl.c, selected with BFHIP_RCCL_LIBRARY): copies between HOST buffers,
    // so that RcclComm's bookkeeping can be driven on a box without a GPU. The real transport copies a block to oneself with hipMemcpyAsync.
    int (*MockSelfCopy)(void*, const void*, size_t) = nullptr;
#endif
};
// The default build loads librccl and nothing else. The test-hooks build (-DBFHIP_TEST_HOOKS, libbfhip_testhooks.so) lets the environment
// name a stand-in for the RCCL entry points: an environment variable that makes a library dlopen an arbitrary path has no place in a release.
static const char* rccl_library_override() {
#ifdef BFHIP_TEST_HOOKS
    return getenv("BFHIP_RCCL_LIBRARY");
#else
    return nullptr;
#endif
}
static RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char* over = rccl_library_override()) api.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);      // a test double of the 10 entry points
        else for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
        if (!api.lib) return;
        auto sym = [&](const char* n) { return dlsym(api.lib, n); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.CommGetAsyncError = (decltype(api.CommGetAsyncError))sym("ncclCommGetAsyncError");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
#ifdef BFHIP_TEST_HOOKS
        api.MockSelfCopy = (decltype(api.MockSelfCopy))sym("bfhip_mock_self_copy");
#endif
    });
    if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.AllReduce || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
        throw HipError("RCCL is not available (librccl.so.1 could not be loaded): a multi-process shard group needs it");
    return api;
}
#define BF_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) throw HipError(std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r__) : "RCCL error")); } while (0)

void rccl_unique_id(unsigned char id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    BF_NCCL(rccl().GetUniqueId(&u));
    memcpy(id, &u, 128);
}

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    RcclComm(const unsigned char id[128], u32 r, u32 n) {
        rank = r; count = n;
        ncclUniqueId u; memcpy(&u, id, 128);
        BF_NCCL(rccl().CommInitRank(&comm, (int)n, u, (int)r));
    }
    bool aborted = false;
    ~RcclComm() override { if (comm && !aborted && rccl().CommDestroy) (void)rccl().CommDestroy(comm); }
    void check_async() override {
        if (!comm || !rccl().CommGetAsyncError) return;
        ncclResult_t st = ncclSuccess;
        if (rccl().CommGetAsyncError(comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress)
            throw HipError(std::string("shard group: RCCL reported an asynchronous error: ") + (rccl().GetErrorString ? rccl().GetErrorString(st) : "?"));
    }
    void abort() override { if (comm && !aborted && rccl().CommAbort) { aborted = true; (void)rccl().CommAbort(comm); } }
    bool spans_devices() const override { return count > 1; }
    const char* transport() const override {
        // an overridden library path is visible to whoever asks which transport a group runs on (bfhip_ctx_group_info)
        static const std::string over = [] { const char* v = rccl_library_override(); return v ? std::string("RCCL entry points from BFHIP_RCCL_LIBRARY=") + v : std::string(); }();
        return over.empty() ? "RCCL (one process per GPU, collectives on the context's stream over xGMI)" : over.c_str();
    }
    void all_gather(hipStream_t s, void* buf, size_t bpr) override {
        n_all_gather++; bytes_sent += bpr * (count - 1);
        Timed tm(*this, s, T_ALL_GATHER);
        BF_NCCL(rccl().AllGather((const char*)buf + rank * bpr, buf, bpr, ncclUint8, comm, s));   // in place: send block = my block of the receive buffer
    }
    void all_reduce_max_u32(hipStream_t s, u32* buf, size_t n) override {
        n_all_reduce++; bytes_sent += n * sizeof(u32);
        Timed tm(*this, s, T_ALL_REDUCE);
        if (n) BF_NCCL(rccl().AllReduce(buf, buf, n, ncclUint32, ncclMax, comm, s));
    }
    void exchange(hipStream_t s, const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs) override {
        n_exchange++; for (auto& x : sends) if (x.peer != rank) bytes_sent += x.bytes;
        Timed tm(*this, s, T_EXCHANGE);
        // blocks to oneself are plain copies, matched in order
        std::vector<const Xfer*> self_s, self_r;
        for (auto& x : sends) if (x.peer == rank) self_s.push_back(&x);
        for (auto& x : recvs) if (x.peer == rank) self_r.push_back(&x);
        if (self_s.size() != self_r.size()) throw HipError("shard group: unmatched self transfer");
        for (size_t i = 0; i < self_s.size(); i++) {
            if (self_s[i]->bytes != self_r[i]->bytes) throw HipError("shard group: unmatched self transfer");
            if (!self_s[i]->bytes) continue;
            // host-memory blocks exist only behind the raw test entry (bfhip_rccl_exchange_raw: null stream) with a test double loaded; a
            // prover's exchange always carries its stream and always copies on the device, whatever library BFHIP_RCCL_LIBRARY named
#ifdef BFHIP_TEST_HOOKS
            if (!s && rccl().MockSelfCopy) { if (rccl().MockSelfCopy(self_r[i]->ptr, self_s[i]->ptr, self_s[i]->bytes) != 0) throw HipError("shard group: self copy failed"); continue; }
#endif
            BF_HIP(hipMemcpyAsync(self_r[i]->ptr, self_s[i]->ptr, self_s[i]->bytes, hipMemcpyDeviceToDevice, s));
        }
        BF_NCCL(rccl().GroupStart());
        for (auto& x : sends) if (x.peer != rank && x.bytes) BF_NCCL(rccl().Send(x.ptr, x.bytes, ncclUint8, (int)x.peer, comm, s));
        for (auto& x : recvs) if (x.peer != rank && x.bytes) BF_NCCL(rccl().Recv(x.ptr, x.bytes, ncclUint8, (int)x.peer, comm, s));
        BF_NCCL(rccl().GroupEnd());
    }
};
std::unique_ptr<Comm> rccl_comm_join(const unsigned char id[128], u32 rank, u32 count) { return std::unique_ptr<Comm>(new RcclComm(id, rank, count)); }

}  // namespace bf
