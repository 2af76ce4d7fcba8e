// C ABI (include/bfhip.h) over the gfx950 kernels. No torch types, plain pointers and sizes.
#include "../../include/bfhip.h"
#include "ctx.h"
#include <cstdlib>
#include "host/circle.h"
#include "host/quotients.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>

using namespace bf;

static thread_local std::string g_err;
void bfhip_set_error(const std::string& s) { g_err = s; }

#define API_TRY try {
#define API_CTX(ctx) try { if (!(ctx)) throw HipError("null context"); (ctx)->c.bind();
#define API_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -1; }

namespace bf {

void Ctx::ensure_aux() { for (auto& a : aux) if (!a) BF_HIP(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); }
void Ctx::ensure_side() { if (!stream2) BF_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking)); }

// A throw anywhere in here leaves a partially built context: the caller (bfhip_ctx_create, pool.hip) lets the destructor run destroy(), which
// releases exactly what exists (r06; before, a failed creation — e.g. out of memory at the twiddle tree — leaked its streams, events, pinned
// buffers and device allocations).
void Ctx::init(int dev, u32 max_log_domain, const Ctx* tables_from) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw HipError("no HIP device: the bfhip backend has no CPU fallback");
    if (dev < 0 || dev >= n) throw HipError("bad device id");
    // columns are addressed with 32-bit byte offsets (kernels.h: ld_col): at most 2^29 cells per column
    if (max_log_domain < 6 || max_log_domain > 29) throw HipError("max_log_domain out of range [6, 29]");
    device = dev;
    BF_HIP(hipSetDevice(dev));
    BF_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    id_main = stream;
    if (const char* v = getenv("BFHIP_SYNC")) sync_blocking = v[0] == 'b';
    if (const char* v = getenv("BFHIP_SINGLE_STREAM")) single_stream = atoi(v) != 0;
    if (const char* v = getenv("BFHIP_OVERLAP")) { overlap = (u32)atoi(v) & 7u; overlap_user_set = true; }
    if (const char* v = getenv("BFHIP_MAILBOX")) mailbox_mode = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("BFHIP_MAILBOX_TIMEOUT_MS")) mailbox_timeout = std::max(1, atoi(v)) * 1e-3;
#ifdef BFHIP_TEST_HOOKS      // libbfhip_testhooks.so only (Makefile): the late-host path of the mailboxes needs a host that is late on purpose
    if (const char* v = getenv("BFHIP_MAILBOX_TEST_DELAY_MS")) mailbox_test_delay_ms = std::max(0, atoi(v));
#endif
    if (overlap) ensure_aux();      // the partner streams exist only for contexts that use them (ctx.h: ensure_aux)
    for (auto& e : evp) BF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : reap_ev) BF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ev) BF_HIP(hipEventCreate(&e));
    BF_HIP(hipEventCreateWithFlags(&sync_ev, hipEventDisableTiming));
    BF_HIP(hipEventCreateWithFlags(&block_ev, hipEventDisableTiming | hipEventBlockingSync));
    BF_HIP(hipHostMalloc((void**)&h_stage, stage_bytes));
    BF_HIP(hipHostMalloc((void**)&h_small, h_small_bytes));
    memset(h_small, 0, 4096);       // flags, stamps and the mailbox error word start at zero; proof numbers start at one
    BF_HIP(hipHostGetDevicePointer((void**)&d_small_alias, h_small, 0));
    BF_HIP(hipHostGetDevicePointer((void**)&d_hstage_alias, h_stage, 0));
    BF_HIP(hipMalloc((void**)&d_stage, stage_bytes));
    BF_HIP(hipMalloc((void**)&d_counters, 4 * 64 * sizeof(u32)));
    BF_HIP(hipMemset(d_counters, 0, 4 * 64 * sizeof(u32)));
    tw_root_log = max_log_domain - 1;
    if (tables_from) {
        // sub-context of a pool: the twiddle tree and the point tables are read-only after creation and the same for every context of a device
        if (tables_from->device != dev) throw HipError("shared tables live on another device");
        if (tables_from->tw_root_log < tw_root_log || !tables_from->d_tw) throw HipError("shared twiddle tree is too small");
        owns_tables = false;
        tw_root_log = tables_from->tw_root_log;      // a tree rooted higher serves every smaller domain (the layers nest: SURVEY.md B.2 item 9)
        d_tw = tables_from->d_tw; d_itw = tables_from->d_itw; d_tlo = tables_from->d_tlo; d_thi = tables_from->d_thi;
        return;
    }
    // point tables: G^a and G^(b << 16) for the M31 circle generator G = (2, 1268011823)
    std::vector<uint2> tlo(1 << 16), thi(1 << 15);
    auto mulp = [](uint2 p, uint2 q) { return uint2{m_sub(m_mul(p.x, q.x), m_mul(p.y, q.y)), m_add(m_mul(p.x, q.y), m_mul(p.y, q.x))}; };
    uint2 g{2u, 1268011823u}, cur{1u, 0u};
    for (u32 i = 0; i < (1u << 16); i++) { tlo[i] = cur; cur = mulp(cur, g); }
    uint2 g16 = cur;  // G^(2^16)
    cur = uint2{1u, 0u};
    for (u32 i = 0; i < (1u << 15); i++) { thi[i] = cur; cur = mulp(cur, g16); }
    BF_HIP(hipMalloc((void**)&d_tlo, tlo.size() * sizeof(uint2)));
    BF_HIP(hipMalloc((void**)&d_thi, thi.size() * sizeof(uint2)));
    BF_HIP(hipMemcpy(d_tlo, tlo.data(), tlo.size() * sizeof(uint2), hipMemcpyHostToDevice));
    BF_HIP(hipMemcpy(d_thi, thi.data(), thi.size() * sizeof(uint2), hipMemcpyHostToDevice));
    BF_HIP(hipMalloc((void**)&d_tw, sizeof(u32) << tw_root_log));
    BF_HIP(hipMalloc((void**)&d_itw, sizeof(u32) << tw_root_log));
    gen_twiddles(stream, d_tw, d_itw, tw_root_log, d_tlo, d_thi);
    BF_HIP(hipGetLastError());
    sync();
}

void Ctx::destroy() {
    if (stream || stream2 || h_stage || h_small || d_stage) (void)hipSetDevice(device);      // may run on any thread (a destructor)
    if (stream) (void)hipStreamSynchronize(stream);
    if (stream2) (void)hipStreamSynchronize(stream2);
    for (auto& a : aux) if (a) { (void)hipStreamSynchronize(a); prof_forget(a); (void)hipStreamDestroy(a); a = nullptr; }
    for (auto& e : evp) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : reap_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    shard = ShardGroup();
    shared_pre = nullptr;
    if (stream) prof_forget(stream);
    if (stream2) prof_forget(stream2);
    arena.release();
    auto dfree = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
    dfree(d_counters); dfree(d_stage);
    if (owns_tables) { dfree(d_tw); dfree(d_itw); dfree(d_tlo); dfree(d_thi); }
    else d_tw = d_itw = nullptr, d_tlo = d_thi = nullptr;
    if (h_stage) { (void)hipHostFree(h_stage); h_stage = nullptr; }
    if (h_small) { (void)hipHostFree(h_small); h_small = nullptr; }
    d_small_alias = d_hstage_alias = nullptr;
    for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (sync_ev) { (void)hipEventDestroy(sync_ev); sync_ev = nullptr; }
    if (block_ev) { (void)hipEventDestroy(block_ev); block_ev = nullptr; }
    if (stream2) { (void)hipStreamDestroy(stream2); stream2 = nullptr; }
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    id_main = nullptr;
}

}  // namespace bf

extern "C" {

const char* bfhip_last_error(void) { return g_err.c_str(); }

int32_t bfhip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int32_t bfhip_device_memory(int32_t device_id, uint64_t* free_bytes, uint64_t* total_bytes) {
    API_TRY
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw HipError("no HIP device: the bfhip backend has no CPU fallback");
    if (device_id < 0 || device_id >= n) throw HipError("bad device id");
    BF_HIP(hipSetDevice(device_id));
    size_t f = 0, t = 0;
    BF_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
    API_CATCH
}

// Diagnostic: the shader clock the device sustains under the dominant kernel's instruction mix (merkle.hip: k_clock_probe).
int32_t bfhip_clock_probe(bfhip_ctx* ctx, double seconds, double out[6]) {
    API_CTX(ctx)
    if (!out) throw HipError("null argument");
    if (!(seconds > 0.0) || seconds > 30.0) throw HipError("bfhip_clock_probe: seconds must be in (0, 30]");
    Ctx& c = ctx->c;
    c.sync();
    const u32 blocks = 256 * 8, iters = 2048;          // 8 workgroups of 4 waves per CU; ~2.5 ms per launch
    uint4* d_stamps = nullptr; u32* d_sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<uint4> st(blocks);
    u64 launches = 0; float ms = 0.f;
    try {
        BF_HIP(hipMalloc((void**)&d_stamps, blocks * sizeof(uint4)));
        BF_HIP(hipMalloc((void**)&d_sink, (size_t)blocks * 256 * sizeof(u32)));
        BF_HIP(hipEventCreate(&e0)); BF_HIP(hipEventCreate(&e1));
        const auto t0 = std::chrono::steady_clock::now();
        // back-to-back launches (a few in the queue at any time) until `seconds` have passed; the LAST launch's stamps are read
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int k = 0; k < 8; k++) { clock_probe_launch(c.stream, d_stamps, d_sink, blocks, iters); launches++; }
            BF_HIP(hipGetLastError());
            c.sync();
        }
        BF_HIP(hipEventRecord(e0, c.stream));
        for (int k = 0; k < 8; k++) clock_probe_launch(c.stream, d_stamps, d_sink, blocks, iters);
        BF_HIP(hipEventRecord(e1, c.stream));
        BF_HIP(hipMemcpyAsync(st.data(), d_stamps, blocks * sizeof(uint4), hipMemcpyDeviceToHost, c.stream));
        c.sync();
        BF_HIP(hipEventElapsedTime(&ms, e0, e1));
    } catch (...) { (void)hipFree(d_stamps); (void)hipFree(d_sink); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); throw; }
    (void)hipFree(d_stamps); (void)hipFree(d_sink); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    std::vector<double> ghz;
    for (auto& s4 : st) {
        const double cyc = (double)(((u64)s4.y << 32) | s4.x), ticks = (double)(((u64)s4.w << 32) | s4.z);
        if (ticks > 0) ghz.push_back(cyc / ticks * 0.1);       // cycles per 10 ns tick -> GHz
    }
    if (ghz.empty()) throw HipError("bfhip_clock_probe: no stamps");
    std::sort(ghz.begin(), ghz.end());
    out[0] = ghz[ghz.size() / 2]; out[1] = ghz.front(); out[2] = ghz.back();
    out[3] = ms > 0 ? 8.0 * blocks * 256.0 * iters / (ms * 1e-3) / 1e9 : 0.0;      // G compressions/s of the last 8 launches
    out[4] = (double)(launches + 8);
    out[5] = ms > 0 ? ms / 8.0 : 0.0;                                                 // ms per launch
    return 0;
    API_CATCH
}

// Diagnostic: the clock the device holds under the REAL Merkle kernel (VALU + its memory traffic), and that kernel's rate on a fixed shape.
int32_t bfhip_clock_probe_mix(bfhip_ctx* ctx, double seconds, uint32_t log_nodes, double out[6]) {
    API_CTX(ctx)
    if (!out) throw HipError("null argument");
    if (!(seconds > 0.0) || seconds > 10.0) throw HipError("bfhip_clock_probe_mix: seconds must be in (0, 10]");
    Ctx& c = ctx->c;
    c.ensure_side();
    c.sync();
    BF_HIP(hipStreamSynchronize(c.stream2));
    if (log_nodes < 16 || log_nodes > 26) throw HipError("bfhip_clock_probe_mix: log_nodes must be in [16, 26]");
    const u32 log = log_nodes;                            // an inner layer of 2^log nodes without columns: one compression per node, 96 B of traffic per node
    const size_t prev_words = (size_t(2) << log) * 8, out_words = (size_t(1) << log) * 8;
    u32 *d_prev = nullptr, *d_out = nullptr; uint4* d_stamp = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    u32* stop = reinterpret_cast<u32*>(c.h_small + 3456);     // a pinned word between the flag and stamp slots (unused by proofs)
    uint4 st[2] = {};
    u64 launches = 0; float ms = 0.f;
    try {
        BF_HIP(hipMalloc((void**)&d_prev, prev_words * sizeof(u32)));
        BF_HIP(hipMalloc((void**)&d_out, out_words * sizeof(u32)));
        BF_HIP(hipMalloc((void**)&d_stamp, 2 * sizeof(uint4)));
        BF_HIP(hipEventCreate(&e0)); BF_HIP(hipEventCreate(&e1));
        fill_mix(c.stream, d_prev, prev_words);
        for (int k = 0; k < 8; k++) merkle_layer(c.stream, d_out, d_prev, nullptr, 0, log, 0.0, 0, 0, 0);      // warm
        BF_HIP(hipGetLastError());
        c.sync();
        __atomic_store_n(stop, 0u, __ATOMIC_RELEASE);
        // the sampler first (it is resident before the wide launches arrive), bounded by 4 x the window whatever happens
        clock_sampler_launch(c.stream2, d_stamp, c.small_alias(stop), (unsigned long long)(seconds * 4.0 * 1e8));
        BF_HIP(hipEventRecord(e0, c.stream));
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int k = 0; k < (log >= 24 ? 4 : 32); k++) { merkle_layer(c.stream, d_out, d_prev, nullptr, 0, log, 0.0, 0, 0, 0); launches++; }
            BF_HIP(hipGetLastError());
            c.sync();
        }
        BF_HIP(hipEventRecord(e1, c.stream));
        c.sync();
        __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
        BF_HIP(hipStreamSynchronize(c.stream2));
        BF_HIP(hipMemcpy(st, d_stamp, sizeof st, hipMemcpyDeviceToHost));
        BF_HIP(hipEventElapsedTime(&ms, e0, e1));
    } catch (...) {
        __atomic_store_n(stop, 1u, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(c.stream2);
        (void)hipFree(d_prev); (void)hipFree(d_out); (void)hipFree(d_stamp); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
        throw;
    }
    (void)hipFree(d_prev); (void)hipFree(d_out); (void)hipFree(d_stamp); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double cyc = (double)(((u64)st[0].y << 32) | st[0].x), ticks = (double)(((u64)st[0].w << 32) | st[0].z);
    const double comp = (double)launches * (double)(size_t(1) << log);
    out[0] = ticks > 0 ? cyc / ticks * 0.1 : 0.0;                    // GHz under the real kernel's mix
    out[1] = ms > 0 ? comp / (ms * 1e-3) / 1e9 : 0.0;                // G compressions/s of k_merkle_layer on this shape (includes the host's sync gaps: 32 launches per sync)
    out[2] = (double)launches;
    out[3] = ms > 0 ? ms / (double)launches * 1e3 : 0.0;             // us per launch
    out[4] = (double)st[1].x;                                        // 1: the sampler was stopped by the host (it spanned the window); 0: it ran into its own bound
    out[5] = ticks * 1e-8;                                           // seconds the sampler covered
    return 0;
    API_CATCH
}

int32_t bfhip_ctx_create(int32_t device_id, uint32_t max_log_domain, bfhip_ctx** out) {
    API_TRY
    if (!out) throw HipError("null argument");
    auto* c = new bfhip_ctx();
    try { c->c.init(device_id, max_log_domain); } catch (...) { delete c; throw; }     // ~Ctx releases whatever init() had created
    *out = c;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_destroy(bfhip_ctx* ctx) { API_TRY if (ctx) { bfhip_ctx_reuse_preprocessed(ctx, 0); delete ctx; } return 0; API_CATCH }
// ---- shard groups: one proof over several GPUs (comm.h) ------------------------------------------------------------------------------
struct bfhip_local_group { std::shared_ptr<LocalGroup> g; uint32_t count; };
static void join_group(bfhip_ctx* ctx, std::unique_ptr<Comm> comm) {
    u32 count = comm->count, lc = 0;
    while ((1u << lc) < count) lc++;
    ctx->c.sync();
    preprocessed_cache_invalidate(&ctx->c);
    ShardGroup g; g.rank = comm->rank; g.count = count; g.log_count = lc; g.comm = std::shared_ptr<Comm>(std::move(comm));
    ctx->c.shard = g;
}
static void check_group_size(uint32_t rank, uint32_t count) {
    if (count < 2 || (count & (count - 1)) != 0 || count > 64) throw HipError("shard count must be a power of two in [2, 64]");
    if (rank >= count) throw HipError("shard rank out of range");
}
int32_t bfhip_local_group_create(uint32_t count, bfhip_local_group** out) {
    API_TRY
    check_group_size(0, count);
    *out = new bfhip_local_group{local_group_create(count), count};
    return 0;
    API_CATCH
}
int32_t bfhip_local_group_destroy(bfhip_local_group* g) { delete g; return 0; }   // members that have not left yet keep the rendezvous alive
int32_t bfhip_ctx_join_local_group(bfhip_ctx* ctx, bfhip_local_group* group, uint32_t rank) {
    API_CTX(ctx)
    if (!group) throw HipError("null group");
    check_group_size(rank, group->count);
    ctx->c.ensure_aux();                  // a group's exchange may run on the partner stream (Ctx::exchange_overlapped)
    join_group(ctx, local_comm_join(group->g, rank));
    return 0;
    API_CATCH
}
int32_t bfhip_rccl_unique_id(uint8_t id[128]) { API_TRY rccl_unique_id(id); return 0; API_CATCH }
int32_t bfhip_ctx_join_rccl_group(bfhip_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t count) {
    API_CTX(ctx)
    check_group_size(rank, count);
    ctx->c.ensure_aux();
    join_group(ctx, rccl_comm_join(id, rank, count));
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_group_stats(bfhip_ctx* ctx, uint64_t out[4]) {
    API_CTX(ctx)
    const Comm* m = ctx->c.shard.comm.get();
    out[0] = m ? m->n_all_gather : 0; out[1] = m ? m->n_all_reduce : 0; out[2] = m ? m->n_exchange : 0; out[3] = m ? m->bytes_sent : 0;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_group_times(bfhip_ctx* ctx, double out_ms[3]) {
    API_CTX(ctx)
    out_ms[0] = out_ms[1] = out_ms[2] = 0.0;
    if (Comm* m = ctx->c.shard.comm.get()) { ctx->c.sync(); m->times_ms(out_ms); }
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_set_shard_policy(bfhip_ctx* ctx, int32_t policy) {
    API_CTX(ctx)
    if (policy < -1 || policy > 1) throw HipError("bfhip_ctx_set_shard_policy: -1 = automatic, 0 = exchange columns -> rows, 1 = replicate the transforms");
    ctx->c.sync();
    if (policy != ctx->c.shard_policy) preprocessed_cache_invalidate(&ctx->c);
    ctx->c.shard_policy = policy;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_group_latency(bfhip_ctx* ctx, int32_t reset, double out_us[21]) {
    API_CTX(ctx)
    if (!out_us) throw HipError("null argument");
    for (int i = 0; i < 21; i++) out_us[i] = 0.0;
    if (Comm* m = ctx->c.shard.comm.get()) { ctx->c.sync(); m->latency_us(reinterpret_cast<double(*)[7]>(out_us), reset != 0); }
    return 0;
    API_CATCH
}
// Test entry (tests/test_abi_and_replicas.py with tests/mock_rccl.c as BFHIP_RCCL_LIBRARY): joins an RCCL group and runs ONE grouped
// send-receive whose blocks live in HOST memory — RcclComm's bookkeeping (self blocks, zero-byte blocks, several blocks per peer, matching
// order) driven through the 10 RCCL entry points without a GPU. With the real librccl the pointers must be device memory.
int32_t bfhip_rccl_exchange_raw(const uint8_t id[128], uint32_t rank, uint32_t count, uint32_t n_sends, const uint32_t* send_peer, void* const* send_ptr, const size_t* send_bytes,
                                uint32_t n_recvs, const uint32_t* recv_peer, void* const* recv_ptr, const size_t* recv_bytes, uint64_t stats_out[4]) {
    API_TRY
    if (!id || (n_sends && (!send_peer || !send_ptr || !send_bytes)) || (n_recvs && (!recv_peer || !recv_ptr || !recv_bytes))) throw HipError("null argument");
    check_group_size(rank, count);
    std::unique_ptr<Comm> comm = rccl_comm_join(id, rank, count);
    std::vector<Xfer> sends, recvs;
    for (u32 i = 0; i < n_sends; i++) sends.push_back({send_peer[i], send_ptr[i], send_bytes[i]});
    for (u32 i = 0; i < n_recvs; i++) recvs.push_back({recv_peer[i], recv_ptr[i], recv_bytes[i]});
    comm->exchange(nullptr, sends, recvs);
    if (stats_out) { stats_out[0] = comm->n_all_gather; stats_out[1] = comm->n_all_reduce; stats_out[2] = comm->n_exchange; stats_out[3] = comm->bytes_sent; }
    return 0;
    API_CATCH
}
// Exercises the RCCL transport with a communicator of ONE rank on this context's GPU: library load, communicator creation, an in-place
// all-gather, a max-reduce and a grouped exchange (a block to oneself). What a single-GPU box can check of the multi-process path.
int32_t bfhip_rccl_selftest(bfhip_ctx* ctx) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    unsigned char id[128];
    rccl_unique_id(id);
    std::unique_ptr<Comm> comm = rccl_comm_join(id, 0, 1);
    const u32 n = 1024;
    u32 *a = nullptr, *b = nullptr;
    BF_HIP(hipMalloc((void**)&a, n * sizeof(u32)));
    hipError_t e = hipMalloc((void**)&b, n * sizeof(u32));
    if (e != hipSuccess) { (void)hipFree(a); BF_HIP(e); }
    std::vector<u32> h(n), out(n);
    for (u32 i = 0; i < n; i++) h[i] = i * 2654435761u;
    try {
        BF_HIP(hipMemcpyAsync(a, h.data(), n * sizeof(u32), hipMemcpyHostToDevice, c.stream));
        comm->all_gather(c.stream, a, n * sizeof(u32));
        comm->all_reduce_max_u32(c.stream, a, n);
        comm->exchange(c.stream, {Xfer{0, a, n * sizeof(u32)}}, {Xfer{0, b, n * sizeof(u32)}});
        BF_HIP(hipMemcpyAsync(out.data(), b, n * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
        c.sync();
    } catch (...) { (void)hipFree(a); (void)hipFree(b); throw; }
    (void)hipFree(a); (void)hipFree(b);
    if (out != h) throw HipError("RCCL self-test: data mismatch");
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_leave_group(bfhip_ctx* ctx) {
    API_CTX(ctx)
    ctx->c.sync();
    preprocessed_cache_invalidate(&ctx->c);
    ctx->c.shard = ShardGroup();
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_group_info(bfhip_ctx* ctx, uint32_t* rank, uint32_t* count, const char** transport) {
    API_CTX(ctx)
    if (rank) *rank = ctx->c.shard.rank;
    if (count) *count = ctx->c.shard.count;
    if (transport) *transport = ctx->c.shard.comm ? ctx->c.shard.comm->transport() : "none";
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_set_conventions(bfhip_ctx* ctx, const bfhip_conventions* conv) {
    API_CTX(ctx)
    Conventions cv;
    if (conv) {
        if (conv->merkle_node_hash > 1 || conv->mix_u64 > 1 || conv->logup_mask_order > 1 || conv->merkle_channel > 1) throw HipError("unknown convention value");
        cv.merkle_node_hash = conv->merkle_node_hash; cv.mix_u64 = conv->mix_u64; cv.logup_mask_order = conv->logup_mask_order; cv.merkle_channel = conv->merkle_channel;
    }
    ctx->c.sync();
    // a kept preprocessed tree is keyed on the hasher it was built with (prover.hip: PreprocessedCache::matches) and dropped here as well
    if (cv.merkle_node_hash != ctx->c.conv.merkle_node_hash || cv.merkle_channel != ctx->c.conv.merkle_channel) preprocessed_cache_invalidate(&ctx->c);
    ctx->c.conv = cv;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_set_sync_policy(bfhip_ctx* ctx, int32_t blocking) { API_CTX(ctx) ctx->c.sync_blocking = blocking != 0; return 0; API_CATCH }
int32_t bfhip_ctx_set_mailbox(bfhip_ctx* ctx, int32_t mode, uint32_t timeout_ms, int32_t test_delay_ms) {
    API_CTX(ctx)
    if (mode < -2 || mode > 1) throw HipError("bfhip_ctx_set_mailbox: mode must be -2 (keep), -1, 0 or 1");
    if (mode != -2) ctx->c.mailbox_mode = mode;
    if (timeout_ms) ctx->c.mailbox_timeout = timeout_ms * 1e-3;
#ifdef BFHIP_TEST_HOOKS
    if (test_delay_ms >= 0) ctx->c.mailbox_test_delay_ms = test_delay_ms;
#else
    if (test_delay_ms > 0) throw HipError("bfhip_ctx_set_mailbox: test_delay_ms needs the test-hooks build of the library (libbfhip_testhooks.so)");
#endif
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_set_overlap(bfhip_ctx* ctx, uint32_t mask) {
    API_CTX(ctx)
    if (mask > 7) throw HipError("overlap mask: bit 0 = tree commitment, bit 1 = quotients / FRI first layer, bit 2 = shard-group exchanges");
    ctx->c.sync();
    if (mask) ctx->c.ensure_aux();
    ctx->c.overlap = mask;
    ctx->c.overlap_user_set = true;      // also switches OFF the default exchange overlap of a multi-GPU shard group when bit 2 is clear
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_memory(bfhip_ctx* ctx, uint64_t out[4]) {
    API_CTX(ctx)
    if (!out) throw HipError("null argument");
    uint64_t reserved = 0;
    for (auto& ch : ctx->c.arena.chunks) reserved += ch.size;
    out[0] = reserved; out[1] = ctx->c.arena.peak; out[2] = ctx->c.owns_tables ? (uint64_t)(2 * sizeof(u32)) << ctx->c.tw_root_log : 0;      // a pool's sub-contexts borrow the first one's out[3] = ctx->c.arena.total_used;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_get_conventions(bfhip_ctx* ctx, bfhip_conventions* out) {
    API_CTX(ctx)
    bfhip_conventions r{};
    r.merkle_node_hash = ctx->c.conv.merkle_node_hash; r.mix_u64 = ctx->c.conv.mix_u64; r.logup_mask_order = ctx->c.conv.logup_mask_order; r.merkle_channel = ctx->c.conv.merkle_channel;
    *out = r;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_sync(bfhip_ctx* ctx) { API_CTX(ctx) ctx->c.sync(); return 0; API_CATCH }

int32_t bfhip_malloc(bfhip_ctx* ctx, size_t bytes, void** out_d) { API_CTX(ctx) BF_HIP(hipSetDevice(ctx->c.device)); BF_HIP(hipMalloc(out_d, bytes ? bytes : 4)); return 0; API_CATCH }
int32_t bfhip_free(bfhip_ctx* ctx, void* p) { API_CTX(ctx) (void)ctx; BF_HIP(hipFree(p)); return 0; API_CATCH }
int32_t bfhip_upload(bfhip_ctx* ctx, void* dst_d, const void* src_h, size_t bytes) {
    API_CTX(ctx) BF_HIP(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, ctx->c.stream)); ctx->c.sync(); return 0; API_CATCH
}
int32_t bfhip_download(bfhip_ctx* ctx, void* dst_h, const void* src_d, size_t bytes) {
    API_CTX(ctx) BF_HIP(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, ctx->c.stream)); ctx->c.sync(); return 0; API_CATCH
}
int32_t bfhip_memset_zero(bfhip_ctx* ctx, void* dst_d, size_t bytes) { API_CTX(ctx) BF_HIP(hipMemsetAsync(dst_d, 0, bytes, ctx->c.stream)); return 0; API_CATCH }

int32_t bfhip_twiddles(bfhip_ctx* ctx, const uint32_t** tw_d, const uint32_t** itw_d, uint32_t* root_log) {
    API_CTX(ctx) *tw_d = ctx->c.d_tw; *itw_d = ctx->c.d_itw; *root_log = ctx->c.tw_root_log; return 0; API_CATCH
}

int32_t bfhip_interpolate(bfhip_ctx* ctx, uint32_t* const* src_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, int32_t replicated) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    if (replicated && log_size < 4) throw HipError("replicated columns need log_size >= 4");
    if (!replicated && log_size < 3) throw HipError("circle transforms need log_size >= 3");
    if (log_size > c.tw_root_log + 1) throw HipError("log_size exceeds the context's twiddle tree");
    u32 log = replicated ? log_size - 4 : log_size;
    c.stage_checkpoint();
    StageBatch sb(c);
    auto* s = c.stage(src_cols_h, n_cols);
    auto* d = c.stage(dst_cols_h, n_cols);
    const FftJob job{(const u32* const*)s, (u32* const*)d, n_cols, log, log, !replicated};
    FftPlan plan;
    fft_plan(plan, true, &job, 1, c.d_tw, c.d_itw, c.tw_root_log);
    plan.d_groups = c.stage(plan.groups.data(), plan.groups.size());
    sb.end();
    fft_run(c.stream, plan);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}

int32_t bfhip_evaluate(bfhip_ctx* ctx, uint32_t* const* coeff_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, uint32_t log_eval, int32_t replicated) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    if (log_eval < log_size) throw HipError("log_eval < log_size");
    if (replicated && log_size < 4) throw HipError("replicated columns need log_size >= 4");
    if (!replicated && log_eval < 3) throw HipError("circle transforms need log_eval >= 3");
    if (log_eval > c.tw_root_log + 1) throw HipError("log_eval exceeds the context's twiddle tree");
    u32 sh = replicated ? 4 : 0;
    c.stage_checkpoint();
    StageBatch sb(c);
    auto* s = c.stage(coeff_cols_h, n_cols);
    auto* d = c.stage(dst_cols_h, n_cols);
    const FftJob job{(const u32* const*)s, (u32* const*)d, n_cols, log_eval - sh, log_size - sh, !replicated};
    FftPlan plan;
    fft_plan(plan, false, &job, 1, c.d_tw, c.d_itw, c.tw_root_log);
    plan.d_groups = c.stage(plan.groups.data(), plan.groups.size());
    sb.end();
    fft_run(c.stream, plan);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}


int32_t bfhip_is_first_coeffs(bfhip_ctx* ctx, uint32_t log_min, uint32_t log_max, uint32_t* const* dst_cols_h) {
    API_CTX(ctx)
    if (!dst_cols_h) throw HipError("null argument");
    if (log_min < 4 || log_max < log_min || log_max - log_min >= 28) throw HipError("is_first_coeffs: need 4 <= log_min <= log_max < log_min + 28");
    if (log_max > ctx->c.tw_root_log + 1) throw HipError("log_max exceeds the context's twiddle tree");
    IsFirstCols a{}; a.log_min = log_min; a.log_max = log_max;
    for (u32 n = log_min; n <= log_max; n++) a.ptr[n - log_min] = dst_cols_h[n - log_min];
    is_first_coeffs(ctx->c.stream, a, ctx->c.d_itw, ctx->c.tw_root_log);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}
int32_t bfhip_broadcast16(bfhip_ctx* ctx, const uint32_t* rows_d, uint32_t* dst_d, size_t n_rows) {
    API_CTX(ctx) broadcast16(ctx->c.stream, rows_d, dst_d, (u32)(n_rows * 16)); BF_HIP(hipGetLastError()); return 0; API_CATCH
}
int32_t bfhip_bit_reverse(bfhip_ctx* ctx, const uint32_t* src_d, uint32_t* dst_d, uint32_t log_size) {
    API_CTX(ctx) if (src_d == dst_d) throw HipError("bit_reverse is out of place"); bit_reverse(ctx->c.stream, src_d, dst_d, log_size); BF_HIP(hipGetLastError()); return 0; API_CATCH
}
int32_t bfhip_batch_inverse_m31(bfhip_ctx* ctx, const uint32_t* src_d, uint32_t* dst_d, size_t n) {
    API_CTX(ctx) batch_inverse_m31(ctx->c.stream, src_d, dst_d, (u32)n); BF_HIP(hipGetLastError()); return 0; API_CATCH
}
int32_t bfhip_batch_inverse_qm31(bfhip_ctx* ctx, const uint32_t* const src_d[4], uint32_t* const dst_d[4], size_t n) {
    API_CTX(ctx) batch_inverse_qm31(ctx->c.stream, src_d, dst_d, (u32)n); BF_HIP(hipGetLastError()); return 0; API_CATCH
}
int32_t bfhip_accumulate(bfhip_ctx* ctx, uint32_t* dst_d, const uint32_t* src_d, size_t n) {
    API_CTX(ctx) accumulate(ctx->c.stream, dst_d, src_d, (u32)n); BF_HIP(hipGetLastError()); return 0; API_CATCH
}
int32_t bfhip_eval_at_point(bfhip_ctx* ctx, const uint32_t* coeffs_d, uint32_t log_size, int32_t replicated, const uint32_t point_h[8], uint32_t out_h[4]) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    if (replicated && log_size < 4) throw HipError("replicated columns need log_size >= 4");
    c.stage_checkpoint();
    uint4 factors[32];
    Q31 x = q_make(point_h[0], point_h[1], point_h[2], point_h[3]);
    factors[0] = make_uint4(point_h[4], point_h[5], point_h[6], point_h[7]);
    for (u32 b = 1; b < 32; b++) { factors[b] = make_uint4(x.a.a, x.a.b, x.b.a, x.b.b); Q31 s = q_mul(x, x); x = q_subm(q_add(s, s), 1); }
    EvalJob job{coeffs_d, replicated ? log_size - 4 : log_size, 0, replicated ? 4u : 0u, 0};
    const EvalJob* dj = c.stage(&job, 1);
    const uint4* df = c.stage(factors, 32);
    u32 nchunks = job.log_n > 12 ? 1u << (job.log_n - 12) : 1u;
    uint4* partials = nullptr; uint4* dout = nullptr;
    BF_HIP(hipMalloc((void**)&partials, sizeof(uint4) * (nchunks + 1)));
    dout = partials + nchunks;
    eval_at_points(c.stream, dj, 1, nchunks, df, partials, dout);
    uint4 r;
    hipError_t e = hipMemcpyAsync(&r, dout, sizeof(uint4), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    (void)hipFree(partials);
    BF_HIP(e);
    out_h[0] = r.x; out_h[1] = r.y; out_h[2] = r.z; out_h[3] = r.w;
    return 0;
    API_CATCH
}
int32_t bfhip_merkle_commit_layer(bfhip_ctx* ctx, uint32_t log_size, const void* prev_layer_d, const uint32_t* const* cols_h, const uint32_t* col_shifts_h,
                                  uint32_t n_cols, void* out_hashes_d) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    c.stage_checkpoint();
    std::vector<ColDesc> d(n_cols);
    for (u32 k = 0; k < n_cols; k++) d[k] = ColDesc{cols_h[k], col_shifts_h ? col_shifts_h[k] : 0u, 0};
    const ColDesc* dd = n_cols ? c.stage(d.data(), n_cols) : nullptr;
    merkle_layer(c.stream, out_hashes_d, prev_layer_d, dd, n_cols, log_size, 0.0, 0, 0, c.conv.merkle_node_hash);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}
int32_t bfhip_merkle_commit_layer_poseidon252(bfhip_ctx* ctx, uint32_t log_size, const void* prev_layer_d, const uint32_t* const* cols_h, const uint32_t* col_shifts_h,
                                             uint32_t n_cols, void* out_hashes_d) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    c.stage_checkpoint();
    std::vector<ColDesc> d(n_cols);
    for (u32 k = 0; k < n_cols; k++) d[k] = ColDesc{cols_h[k], col_shifts_h ? col_shifts_h[k] : 0u, 0};
    const ColDesc* dd = n_cols ? c.stage(d.data(), n_cols) : nullptr;
    merkle_layer_poseidon(c.stream, out_hashes_d, prev_layer_d, dd, n_cols, log_size);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}
int32_t bfhip_hades_permutation(bfhip_ctx* ctx, const uint32_t in_h[24], uint32_t out_h[24]) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    c.stage_checkpoint();
    u32* din = (u32*)c.stage(in_h, 24);
    u32* dout = (u32*)c.stage(in_h, 24);
    hades_once(c.stream, din, dout);
    BF_HIP(hipMemcpyAsync(out_h, dout, 24 * sizeof(u32), hipMemcpyDeviceToHost, c.stream));
    c.sync();
    return 0;
    API_CATCH
}
static Q31 q_from_h(const uint32_t v[4]) { return q_make(v[0], v[1], v[2], v[3]); }
static const u32* stage_alpha(Ctx& c, const uint32_t alpha_h[4]) {
    Q31 a = q_from_h(alpha_h), sq = q_mul(a, a);
    u32 w[8] = {a.a.a, a.a.b, a.b.a, a.b.b, sq.a.a, sq.a.b, sq.b.a, sq.b.b};
    c.stage_checkpoint();
    return c.stage(w, 8);
}
int32_t bfhip_fold_line(bfhip_ctx* ctx, const uint32_t* const src_d[4], uint32_t* const dst_d[4], uint32_t log_size, const uint32_t alpha_h[4]) {
    API_CTX(ctx)
    if (log_size < 1 || log_size > ctx->c.tw_root_log) throw HipError("fold_line: log_size outside the twiddle tree");
    fold_line(ctx->c.stream, dst_d, src_d, stage_alpha(ctx->c, alpha_h), ctx->c.d_itw, ctx->c.tw_root_log, log_size); BF_HIP(hipGetLastError()); return 0;
    API_CATCH
}
int32_t bfhip_fold_circle_into_line(bfhip_ctx* ctx, uint32_t* const dst_d[4], const uint32_t* const src_d[4], uint32_t log_size, const uint32_t alpha_h[4]) {
    API_CTX(ctx)
    if (log_size < 3 || log_size > ctx->c.tw_root_log + 1) throw HipError("fold_circle_into_line: log_size outside the twiddle tree");
    fold_circle_into_line(ctx->c.stream, dst_d, src_d, stage_alpha(ctx->c, alpha_h), ctx->c.d_itw, ctx->c.tw_root_log, log_size); BF_HIP(hipGetLastError()); return 0;
    API_CATCH
}
int32_t bfhip_grind(bfhip_ctx* ctx, const uint8_t digest_h[32], uint32_t pow_bits, uint64_t* nonce) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    c.stage_checkpoint();
    u32* d_digest = (u32*)c.stage(digest_h, 32);
    unsigned long long init = ~0ull, best = ~0ull;
    unsigned long long* d_best = c.stage(&init, 1);
    const u32 span = 1u << 20;
    for (u64 base = 0; best == ~0ull; base += span) {
        grind_span(c.stream, d_digest, base, span, pow_bits, d_best, c.conv.mix_u64);
        BF_HIP(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, c.stream));
        c.sync();
        if (base > (u64(1) << 40)) throw HipError("grind: no nonce found below 2^40");
    }
    *nonce = best;
    return 0;
    API_CATCH
}
int32_t bfhip_gather(bfhip_ctx* ctx, const uint32_t* col_d, const uint64_t* idx_h, size_t n, uint32_t* out_h) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    std::vector<GatherReq> req(n);
    for (size_t i = 0; i < n; i++) req[i] = GatherReq{col_d, idx_h[i], (u32)i, 1u};
    GatherReq* dreq = nullptr; u32* dout = nullptr;
    BF_HIP(hipMalloc((void**)&dreq, sizeof(GatherReq) * (n + 1)));
    hipError_t e = hipMalloc((void**)&dout, sizeof(u32) * (n + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(dreq, req.data(), sizeof(GatherReq) * n, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) { gather_u32(c.stream, dreq, (u32)n, dout); e = hipMemcpyAsync(out_h, dout, sizeof(u32) * n, hipMemcpyDeviceToHost, c.stream); }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    (void)hipFree(dreq); (void)hipFree(dout);
    BF_HIP(e);
    return 0;
    API_CATCH
}

// ---- per-component AIR operations ---------------------------------------------------------------------------------------------------
static Lookups lookups_from_h(const uint32_t v[24]) {
    Lookups el;
    el.memory = make_lookup(q_make(v[0], v[1], v[2], v[3]), q_make(v[4], v[5], v[6], v[7]));
    el.instruction = make_lookup(q_make(v[8], v[9], v[10], v[11]), q_make(v[12], v[13], v[14], v[15]));
    el.processor = make_lookup(q_make(v[16], v[17], v[18], v[19]), q_make(v[20], v[21], v[22], v[23]));
    return el;
}
static void check_component(int32_t component, uint32_t log_size, const Ctx& c, uint32_t extra_log) {
    if (component < 0 || component >= N_COMPONENTS) throw HipError("unknown component");
    if (log_size < LOG_N_LANES) throw HipError("component log_size below LOG_N_LANES (4)");
    if (log_size + extra_log > c.tw_root_log + 1) throw HipError("log_size exceeds the context's twiddle tree");
}
int32_t bfhip_component_shape(int32_t component, uint32_t* n_main, uint32_t* n_logup, uint32_t* n_cons) {
    if (component < 0 || component >= N_COMPONENTS) { g_err = "unknown component"; return -1; }
    if (n_main) *n_main = n_main_cols(component);
    if (n_logup) *n_logup = n_logup_cols(component);
    if (n_cons) *n_cons = n_constraints(component);
    return 0;
}
int32_t bfhip_logup_generate(bfhip_ctx* ctx, int32_t component, uint32_t log_size, const uint32_t* const* main_rows_h, const uint32_t lookup_h[24],
                             uint32_t* const* out_cols_h, uint32_t claimed_sum_h[4]) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    check_component(component, log_size, c, 0);
    u32 log_rows = log_size - LOG_N_LANES;
    size_t M = size_t(1) << log_rows;
    LogupLaunch L{};
    for (u32 j = 0; j < n_main_cols(component); j++) L.cols[j] = main_rows_h[j];
    u32 nl = n_logup_cols(component);
    for (u32 q = 0; q + 1 < nl; q++) for (int w = 0; w < 4; w++) L.out_rep[4 * q + w] = out_cols_h[4 * q + w];
    for (int w = 0; w < 4; w++) L.out_last[w] = out_cols_h[4 * (nl - 1) + w];
    // scratch: vrow[M], wloc[M], totals[M / 1024 + 2], claimed[1]
    size_t n_tot = M / 1024 + 2;
    uint4* scratch = nullptr;
    BF_HIP(hipMalloc((void**)&scratch, sizeof(uint4) * (2 * M + n_tot + 1)));
    L.vrow = scratch; L.wloc = scratch + M; L.totals = scratch + 2 * M; L.claimed = scratch + 2 * M + n_tot;
    L.el = lookups_from_h(lookup_h); L.log_rows = log_rows; L.comp = component;
    {
        LogupBatch lb;
        logup_batch_init(lb, L.el, &L, 1);
        c.stage_checkpoint();
        logup_batch_run(c.stream, c.stage(&lb, 1), lb);
    }
    uint4 r;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&r, L.claimed, sizeof(uint4), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    (void)hipFree(scratch);
    BF_HIP(e);
    claimed_sum_h[0] = r.x; claimed_sum_h[1] = r.y; claimed_sum_h[2] = r.z; claimed_sum_h[3] = r.w;
    return 0;
    API_CATCH
}
int32_t bfhip_eval_constraints(bfhip_ctx* ctx, int32_t component, uint32_t log_size, const uint32_t* is_first_d, const uint32_t* const* main_lde_h,
                               const uint32_t* main_shifts_h, const uint32_t* const* inter_lde_h, const uint32_t* inter_shifts_h, const uint32_t lookup_h[24],
                               const uint32_t claimed_sum_h[4], const uint32_t* coeffs_h, uint32_t* const acc_d[4]) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    check_component(component, log_size, c, 1);
    u32 eval_log = log_size + 1;
    ConstraintLaunch L{};
    L.is_first = is_first_d;
    for (u32 j = 0; j < n_main_cols(component); j++) L.trace[j] = ColDesc{main_lde_h[j], main_shifts_h ? main_shifts_h[j] : 0u, 0};
    for (u32 j = 0; j < 4 * n_logup_cols(component); j++) L.inter[j] = ColDesc{inter_lde_h[j], inter_shifts_h ? inter_shifts_h[j] : 0u, 0};
    for (int w = 0; w < 4; w++) L.acc[w] = acc_d[w];
    for (u32 j = 0; j < n_constraints(component); j++) L.coeff[j] = q_from_h(coeffs_h + 4 * j);
    L.el = lookups_from_h(lookup_h); L.total_sum = q_from_h(claimed_sum_h); L.log_size = log_size;
    // 1 / coset_vanishing(CanonicCoset(log_size).coset, eval_domain.at(i)) takes two values, by the parity of the (bit-reversed) cell index
    for (u32 i = 0; i < 2; i++) L.denom_inv[i] = m_inv(coset_vanishing_m(log_size, canonic_domain_at(eval_log, i)));
    c.stage_checkpoint();
    eval_constraints(c.stream, component, c.stage(&L, 1), log_size, 0, constraint_group_rows(L, component));
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}
int32_t bfhip_accumulate_quotients(bfhip_ctx* ctx, uint32_t log_size, const uint32_t* const* cols_h, const uint32_t* col_shifts_h, uint32_t n_cols,
                                   const uint32_t* n_samples_h, const uint32_t* sample_points_h, const uint32_t* sample_values_h,
                                   const uint32_t random_coeff_h[4], uint32_t* const out_d[4]) {
    API_CTX(ctx)
    Ctx& c = ctx->c;
    if (log_size < 3 || log_size > c.tw_root_log + 1) throw HipError("accumulate_quotients: log_size outside the twiddle tree");
    std::vector<ColDesc> descs(n_cols);
    std::vector<std::vector<ColumnSample>> samples(n_cols);
    size_t si = 0;
    for (u32 k = 0; k < n_cols; k++) {
        descs[k] = ColDesc{cols_h[k], col_shifts_h ? col_shifts_h[k] : 0u, 0};
        if (descs[k].shift == 1) throw HipError("accumulate_quotients: column shift must be 0 or >= 2");
        for (u32 s = 0; s < n_samples_h[k]; s++, si++) {
            const uint32_t* p = sample_points_h + 8 * si;
            samples[k].push_back({PtQ{q_from_h(p), q_from_h(p + 4)}, q_from_h(sample_values_h + 4 * si)});
        }
    }
    std::vector<QuotientBatch> batches; std::vector<QuotientEntry> entries;
    build_quotient_batches(samples, q_from_h(random_coeff_h), batches, entries);
    quotient_entries_finish(batches.data(), batches.size(), entries.data(), descs.data());
    c.stage_checkpoint();
    QuotientArgs a{};
    a.batches = batches.empty() ? nullptr : c.stage(batches.data(), batches.size());
    a.entries = entries.empty() ? nullptr : c.stage(entries.data(), entries.size());
    a.n_batches = (u32)batches.size(); a.log = log_size; a.tw = c.d_tw; a.tw_total = 1u << c.tw_root_log;
    for (int w = 0; w < 4; w++) a.out[w] = out_d[w];
    const u32 blocks = quotient_groups_layout(&a, 1);
    accumulate_quotients(c.stream, c.stage(&a, 1), 1, blocks);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}

}  // extern "C"
