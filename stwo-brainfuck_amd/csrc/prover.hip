// Device-resident counterpart of prove_brainfuck (crates/brainfuck_prover/src/brainfuck_air/mod.rs:471-735) and of the stwo
// driver code it calls (CommitmentSchemeProver / TreeBuilder / prover::prove / FriProver). Everything that touches a column runs
// in the gfx950 kernels; the host keeps only the Fiat–Shamir channel, the sample/batch bookkeeping and the decommitment control
// flow (which depends on query positions, never on column data). There is no CPU fallback for any column operation.
#include <atomic>
#include "../../include/bfhip.h"
#include "ctx.h"
#include "host/circle.h"
#include "host/quotients.h"
#include "host/proof.h"
#include "host/verifier.h"
#include <map>
#include <set>
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <mutex>

namespace bf {

// tables.hip
void build_tables_device(Ctx& c, const std::vector<u32> trace7_soa[7], u32 n, const std::vector<u32>& code, const std::function<u32*(size_t)>& alloc,
                         std::vector<std::vector<u32*>>& cols_out, u32 log_sizes_out[13]);

struct PcsConfig { u32 pow_bits = 5, log_blowup = 1, log_last_layer_degree_bound = 0, n_queries = 3; };  // PcsConfig::default() (mod.rs:479)

static constexpr u32 OWNER_ALL = 0xFFFFFFFFu;   // a polynomial every rank of a shard group holds and transforms itself

struct DCol {
    u32* ptr = nullptr; u32 log_size = 0; u32 shift = 0;   // 2^log_size domain cells, stored as 2^(log_size - shift) u32
    // Row-sharded column of a shard group (lc = log2(ranks) > 0): this rank stores only its contiguous range of 2^(log_size - lc) rows and
    // `ptr` is a VIRTUAL BASE — the slice's address minus the range's first row — so that kernels keep addressing rows by their global
    // index; only rows of the rank's own range may be dereferenced.
    u32 lc = 0;
    bool sliced() const { return lc != 0; }
    size_t stored() const { return size_t(1) << (log_size - shift - lc); }
    ColDesc desc() const { return ColDesc{ptr, shift, 0}; }
    bool mine(u64 cell, u32 rank) const { return lc == 0 || (cell >> (log_size - lc)) == rank; }
};
// layer k: node i stored at i >> shifts[k]. In a shard group (Ctx::shard.count > 1) the layers k with band_lo < k <= band_hi hold only this
// rank's contiguous share of the nodes (node i belongs to rank i >> (k - log2 count)); layer band_lo and everything below is complete.
// cols: the tree's columns in descending size order (what the decommitment walks; kept from the commitment so that the last round trip of a
// proof does not sort 128 descriptors again)
struct DevMerkle { std::vector<u32*> layers; std::vector<u32> shifts; u32 max_log = 0; Hash32 root; int band_lo = 0, band_hi = -1; std::vector<DCol> cols; };
// owner[i]: the rank that holds polynomial i (coefficients) and computes its LDE, or OWNER_ALL. prev[i]: previous-row copy of evals[i]
// (row-sharded last logUp columns only; ptr == nullptr otherwise).
struct DTree { std::vector<DCol> polys, evals, prev; std::vector<u32> owner; DevMerkle mk; };
struct DSecure {
    u32* c[4]; u32 log_size; u32 lc = 0;     // lc > 0: row-sharded, c[] are virtual bases (see DCol)
    bool mine(u64 cell, u32 rank) const { return lc == 0 || (cell >> (log_size - lc)) == rank; }
};

struct Gather {
    std::vector<GatherReq> reqs;
    u32 n_words = 0;
    // each returns the position of the first gathered word in the output of run(); mine == false: another rank of the shard group holds
    // the word (this rank contributes a zero, the max-reduce completes it)
    size_t add(const u32* base, u64 idx, bool mine = true) { reqs.push_back({mine ? base : nullptr, idx, n_words, 1u}); n_words += 1; return n_words - 1; }
    size_t add_hash(const u32* layer, u64 node_slot, bool mine) { reqs.push_back({mine ? layer : nullptr, node_slot * 8, n_words, 8u}); n_words += 8; return n_words - 8; }
    size_t add_col(const DCol& col, u64 cell, u32 rank) { return add(col.ptr, cell >> col.shift, col.mine(cell, rank)); }
    // stamp_slot >= 0 (inside a proof, one process per proof): the host polls a stamp word written behind the gather instead of an event
    std::vector<u32> run(Ctx& c, int stamp_slot = -1) {
        std::vector<u32> out(n_words);
        if (reqs.empty()) return out;
        c.stage_checkpoint();
        if (reqs.size() * sizeof(GatherReq) > c.stage_bytes / 4) throw HipError("decommitment: too many gather requests");
        if (c.shard.count == 1 && n_words * sizeof(u32) <= c.h_small_bytes - 4096) {
            // One process per proof: the kernel reads the request list where the host wrote it (the pinned side of the staging ring) and
            // writes the words into the pinned bounce buffer — two copy commands and their barriers less on the last round trip of a proof.
            size_t bytes = (reqs.size() * sizeof(GatherReq) + 255) & ~size_t(255);
            if (c.stage_used + bytes > c.stage_bytes) throw HipError("staging buffer exhausted (call stage_checkpoint() between operations)");
            memcpy(c.h_stage + c.stage_used, reqs.data(), reqs.size() * sizeof(GatherReq));
            const GatherReq* d = reinterpret_cast<const GatherReq*>(c.d_hstage_alias + c.stage_used);
            c.stage_used += bytes;
            gather_u32(c.stream, d, (u32)reqs.size(), reinterpret_cast<u32*>(c.d_small_alias + 4096));
            if (stamp_slot >= 0 && c.use_mailbox && c.proof_seq) { c.post_stamp(stamp_slot); c.wait_stamp(stamp_slot); }
            else c.sync();
            memcpy(out.data(), c.h_small + 4096, n_words * sizeof(u32));
            return out;
        }
        GatherReq* d = c.stage(reqs.data(), reqs.size());
        u32* dout = c.alloc_u32(n_words);
        gather_u32(c.stream, d, (u32)reqs.size(), dout);
        // shard group: every word is either identical on all ranks (replicated columns, complete layers) or held by one rank and zero
        // elsewhere (rows of sharded columns, hashes of share-wise layers) — an element-wise maximum completes it everywhere
        if (c.shard.count > 1) c.shard.comm->all_reduce_max_u32(c.stream, dout, n_words);
        c.read_back(out.data(), dout, n_words * sizeof(u32));
        return out;
    }
};

// The prover's input once resident in HBM: row-granular main-trace columns of the 13 components (what the reference's
// `XTable::from(&vm_trace).trace_evaluation()` calls produce, mod.rs:511-547, minus the 16x lane broadcast).
struct TraceInput {
    std::vector<std::vector<DCol>> rows;   // [component][column]
    u32 log_sizes[N_COMPONENTS];
    u64 n_steps = 0, main_cells = 0, interaction_cells = 0;
    std::vector<u32*> owned;
    void release() { for (u32* p : owned) (void)hipFree(p); owned.clear(); rows.clear(); }
};

// Optional, off by default: the preprocessed tree (IsFirst columns) depends only on LOG_MAX_ROWS, so a deployment that proves many
// programs can commit it once per context and reuse polynomials, LDE columns and Merkle layers. The reference recomputes it in every
// prove_brainfuck call (mod.rs:495-500); bench.py's headline number does the same (reuse only with --reuse-preprocessed).
// The kept tree is only valid for the configuration it was built under: LOG_MAX_ROWS, the node-hash convention and the shard group
// (share-wise layers hold one rank's share only) — any change rebuilds it.
struct PreprocessedCache {
    bool enabled = false, valid = false, replicate = false; u32 lmr = 0, node_conv = 0, channel = 0, shard_rank = 0, shard_count = 1; DTree tree; Arena keep;
    bool matches(const Ctx& c, u32 log_max_rows) const {
        // the hasher is (merkle_channel, merkle_node_hash): a Blake2s tree must never serve a Poseidon252 proof or the reverse
        return enabled && valid && lmr == log_max_rows && node_conv == c.conv.merkle_node_hash && channel == c.conv.merkle_channel &&
               shard_rank == c.shard.rank && shard_count == c.shard.count && replicate == (c.shard.count > 1 && c.shard_replicate);
    }
};
static std::mutex g_cache_mutex;   // contexts may be driven from different host threads (bench.py --inflight)
static std::map<Ctx*, PreprocessedCache>& preprocessed_caches() { static std::map<Ctx*, PreprocessedCache> m; return m; }
static PreprocessedCache& preprocessed_cache_of(Ctx* c) { std::lock_guard<std::mutex> g(g_cache_mutex); return preprocessed_caches()[c]; }   // map nodes are address-stable
// Group membership changes drop the cached tree: the ranks of a group must take the same decision (reuse or rebuild with its exchanges) in
// every proof, and they do when each starts its membership with an empty cache and then issues the group's common sequence of calls. (A cache
// that survived an earlier membership could match on some ranks only — found by tools/fuzz_campaign.py persistent, seed 60806.)
void preprocessed_cache_invalidate(Ctx* c) {
    std::lock_guard<std::mutex> g(g_cache_mutex);
    auto it = preprocessed_caches().find(c);
    if (it != preprocessed_caches().end()) it->second.valid = false;
}

// A pool's shared preprocessed tree (include/bfhip.h: bfhip_pool_*; pool.hip): committed ONCE per batch (or once per pool) by the pool's
// builder context and read by every proof of the batch instead of being recommitted by each — byte-neutral, the tree depends on LOG_MAX_ROWS
// and the hasher only. The builder enqueues the commitment and records `ready` behind it BEFORE the workers are woken; a proof copies the
// layout at its start and waits (host side) for `ready` where it would have joined its own side stream, so the commitment runs beside the
// batch's first main-trace phases like a proof's own would.
struct SharedPreprocessed {
    bool valid = false; u32 lmr = 0, node_conv = 0, channel = 0;
    DTree tree;                         // storage: the builder context's arena (not reset while a batch can read it)
    hipEvent_t ready = nullptr;         // recorded on the builder's stream behind the tree (and the root's copy into pinned memory)
    const Hash32* pinned_root = nullptr;
    bool matches(const Ctx& c, u32 log_max_rows) const {
        return valid && lmr == log_max_rows && node_conv == c.conv.merkle_node_hash && channel == c.conv.merkle_channel && c.shard.count == 1;
    }
};

struct PhaseTimes { double preprocessed = 0, main_trace = 0, interaction = 0, composition = 0, oods = 0, quotients = 0, fri = 0, decommit = 0, tables = 0, total = 0; };

struct HipProver {
    Ctx& c;
    PcsConfig cfg;
    u32 log_max_rows;
    Channel ch;
    PhaseTimes tm;
    std::string transcript;   // "name:hexdigest\n" per stage, for divergence hunting against the oracle
    bool want_transcript = false;

    HipProver(Ctx& ctx, u32 lmr) : c(ctx), log_max_rows(lmr) {}

    void tap(const char* name) {
        if (!want_transcript) return;
        char buf[3]; transcript += name; transcript += ':';
        for (int i = 0; i < 32; i++) { snprintf(buf, sizeof buf, "%02x", ch.digest.b[i]); transcript += buf; }
        transcript += '\n';
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // BFHIP_TRACE_HOST=1: host-side timestamps of the Fiat-Shamir round trips of a proof (label, microseconds since the proof started),
    // printed to stderr when the proof is done — what the host does while the GPU waits for a challenge (tools: point.py)
    std::vector<std::pair<const char*, double>> host_marks;
    bool trace_host = [] { const char* v = getenv("BFHIP_TRACE_HOST"); return v && (v[0] == '1' || v[0] == '2'); }();
    bool trace_host_mean = [] { const char* v = getenv("BFHIP_TRACE_HOST"); return v && v[0] == '2'; }();      // 2: means over every 20 proofs instead of every proof
    double mark_t0 = 0;
    void mark(const char* label) { if (trace_host) host_marks.push_back({label, (now() - mark_t0) * 1e6}); }
    void print_marks() {
        if (!trace_host) return;
        if (trace_host_mean) {
            static std::mutex mu; static std::vector<std::pair<const char*, double>> sum; static int n = 0;
            std::lock_guard<std::mutex> g(mu);
            if (sum.size() != host_marks.size()) { sum.assign(host_marks.size(), {nullptr, 0.0}); n = 0; }
            double prev = 0;
            for (size_t i = 0; i < host_marks.size(); i++) { sum[i].first = host_marks[i].first; sum[i].second += host_marks[i].second - prev; prev = host_marks[i].second; }
            if (++n == 20) {
                double at = 0;
                for (auto& m : sum) { at += m.second / n; fprintf(stderr, "[bfhip host mean of 20] %10.1f us  (+%7.1f)  %s\n", at, m.second / n, m.first); }
                sum.clear(); n = 0;
            }
            return;
        }
        double prev = 0;
        for (auto& m : host_marks) { fprintf(stderr, "[bfhip host] %10.1f us  (+%7.1f)  %s\n", m.second, m.second - prev, m.first); prev = m.second; }
        fprintf(stderr, "[bfhip host] staging ring: %zu bytes in use after this proof\n", c.stage_used);
    }

    // ---- batched FFT over heterogeneous columns: group by (size, storage) --------------------------------------------------
    // Two steps so that a caller can put the plan's staging into a batch shared with what follows (fft_prepare inside a StageBatch,
    // fft_launch after its end()): pointer arrays and the pass table of every size group reach the device in one copy.
    FftPlan fft_prepare(bool inverse, const std::vector<DCol>& src, const std::vector<DCol>& dst) {
        std::map<std::pair<u32, u32>, std::vector<size_t>> groups;   // (dst log_size, shift) -> indices
        for (size_t i = 0; i < dst.size(); i++) groups[{dst[i].log_size, dst[i].shift}].push_back(i);
        struct Job { size_t off, n; u32 log, src_log, sh; };
        std::vector<Job> jobs; std::vector<const u32*> ptrs;
        for (auto it = groups.rbegin(); it != groups.rend(); ++it) {
            const auto& idx = it->second;
            u32 log = it->first.first, sh = it->first.second;
            // split further by source size (forward transforms may extend from different coefficient sizes)
            std::map<u32, std::vector<size_t>> by_src;
            for (size_t i : idx) by_src[src[i].log_size].push_back(i);
            for (auto& kv : by_src) {
                jobs.push_back({ptrs.size(), kv.second.size(), log, kv.first, sh});
                for (size_t i : kv.second) ptrs.push_back(src[i].ptr);
                for (size_t i : kv.second) ptrs.push_back(dst[i].ptr);
            }
        }
        FftPlan plan;
        if (jobs.empty()) return plan;
        StageBatch sb(c);
        const u32* const* d_ptrs = c.stage(ptrs.data(), ptrs.size());
        std::vector<FftJob> fj;
        for (auto& j : jobs) fj.push_back({d_ptrs + j.off, (u32* const*)(d_ptrs + j.off + j.n), (u32)j.n, j.log - j.sh, j.src_log - j.sh, j.sh == 0});
        fft_plan(plan, inverse, fj.data(), fj.size(), c.d_tw, c.d_itw, c.tw_root_log);
        plan.d_groups = c.stage(plan.groups.data(), plan.groups.size());
        sb.end();
        return plan;
    }
    void fft_launch(const FftPlan& plan) { fft_run(c.stream, plan); BF_HIP(hipGetLastError()); }
    // ---- batched FFT over heterogeneous columns: one launch per pass and kernel kind, whatever the number of sizes ---------------
    void fft_cols(bool inverse, const std::vector<DCol>& src, const std::vector<DCol>& dst) {
        c.stage_checkpoint();
        fft_launch(fft_prepare(inverse, src, dst));
    }

    // ---- Merkle (a4) -----------------------------------------------------------------------------------------------------------
    // defer_root: leave the 32-byte root copy pending in pinned memory (*pinned_root) instead of synchronising the stream.
    // no_readback: the root stays on the device (the caller collects it; FRI commit phase).
    // step: FRI commit phase — the device channel mixes the root and draws the next alpha right behind the tree (fused into the top kernel).
    struct ChannelStep { u32* chan; u32* alpha8; u32* root_copy; };
    // A tree is committed in two steps so that several trees can share ONE staging copy (the FRI commit phase plans all its layers first:
    // every separate copy is a ~9 us blit in front of the kernels that need it): merkle_plan = host-side layout (levels, replication shifts,
    // arena storage, column descriptors written to the staging ring — inside the caller's StageBatch), merkle_run = the launches.
    struct MerklePlan {
        DevMerkle mk; std::vector<DCol> cols; std::vector<size_t> off; std::vector<double> bytes; size_t n_all = 0;
        const ColDesc* d_all = nullptr; MerkleTreeDesc tree{}; bool poseidon = false;
        // launches: levels [max_log .. sub_hi + 1] one each (k_merkle_layer), [sub_hi .. 9] one (k_merkle_subtree; sub_hi == 0: none and the
        // single-level launches go down to fused_top), [fused_top - 1 .. 0] one (k_merkle_top; fused_top == 0: none)
        u32 fused_top = 0, sub_hi = 0;
        double top_bytes = 0, top_comp = 0, sub_bytes = 0, sub_comp = 0;
        // shard group: the band's last levels [band_lo .. band_fuse_hi] as ONE launch over this rank's share (k_merkle_subtree in its general
        // form, a workgroup per 2^band_fuse_r nodes of level band_lo); band_fuse_hi < 0: single-level launches all the way down
        int band_fuse_hi = -1; u32 band_fuse_r = 0; double band_bytes = 0, band_comp = 0;
    };
    MerklePlan merkle_plan(const std::vector<DCol>& cols_in) {
        if (cols_in.empty()) throw HipError("merkle_commit: no columns");
        MerklePlan p;
        p.cols = cols_in;
        std::vector<DCol>& cols = p.cols;
        std::stable_sort(cols.begin(), cols.end(), [](const DCol& a, const DCol& b) { return a.log_size > b.log_size; });
        DevMerkle& mk = p.mk;
        mk.cols = cols;
        mk.max_log = cols[0].log_size;
        mk.layers.resize(mk.max_log + 1);
        mk.shifts.assign(mk.max_log + 1, 0);
        u32 min_col_log = cols.back().log_size;
        // one staging copy for the column descriptors of every level; replication shift of every level
        std::vector<ColDesc> all; p.off.assign(mk.max_log + 2, 0); p.bytes.assign(mk.max_log + 1, 0.0);
        {
            size_t ci = 0;
            for (int log = (int)mk.max_log; log >= 0; log--) {
                p.off[log] = all.size();
                u32 sh = log < (int)mk.max_log ? (mk.shifts[log + 1] ? mk.shifts[log + 1] - 1 : 0) : 32;
                while (ci < cols.size() && cols[ci].log_size == (u32)log) { sh = std::min(sh, cols[ci].shift); p.bytes[log] += 4.0 * cols[ci].stored(); all.push_back(cols[ci++].desc()); }
                mk.shifts[log] = std::min<u32>(sh == 32 ? 0 : sh, (u32)log);
            }
        }
        p.n_all = all.size();
        p.poseidon = c.conv.merkle_channel == 1;   // Poseidon252MerkleHasher: layer kernel of poseidon.hip, no fused top, host channel
        (void)min_col_log;
        // The top kernel takes levels [fused_top - 1 .. 0] (<= 512 nodes in its first level, columns included), all un-replicated, and reads
        // the children of its first level from level fused_top (also un-replicated) unless it starts at the leaves.
        u32 fused_top = std::min<u32>(mk.max_log + 1, 10);
        while (fused_top > 0 && mk.shifts[std::min(fused_top, mk.max_log)] != 0) fused_top--;
        if (p.poseidon) fused_top = 0;
        p.fused_top = fused_top;
        // Shard group: the big layers are hashed SHARE-WISE — rank r takes the stored slots [r * stored / count, (r + 1) * stored / count) of
        // every layer with at least 2^SHARE_MIN_LOG_PER_RANK stored nodes per rank, replicated ones included (a slot's children are slots of
        // the same rank in the layer below, whether that layer is stored at the same replication or one step finer). log - shift is
        // non-decreasing in log, so these layers form one band [band_lo, max_log]; the band's lowest layer is completed on every rank by one
        // all-gather and everything below it is hashed by every rank, redundantly, with the same two launches as a one-GPU proof (subtree +
        // top): a share of fewer than 2^14 nodes is a launch that costs more than it computes (r04: 155 of a rank's 217 layer launches per
        // fib19 proof had <= 512 workgroups and took 0.85 ms per rank: profiles/r04_shard_redundancy_before.txt).
        const ShardGroup& sg = c.shard;
        if (sg.count > 1) {
            // (Poseidon252: a node costs ~40x a Blake2s node and there is no multi-level kernel below the band — shares down to 256 nodes)
            const int share_min = p.poseidon ? 8 : (int)SHARE_MIN_LOG_PER_RANK;
            int lo = std::max<int>((int)sg.log_count + share_min, (int)fused_top);
            while (lo <= (int)mk.max_log && (int)lo - (int)mk.shifts[lo] < (int)sg.log_count + share_min) lo++;
            // a tree with fewer than 2^14 stored leaves per rank is hashed whole by every rank: cheaper than the latency of its all-gather
            const bool worth = (int)mk.max_log - (int)mk.shifts[mk.max_log] >= (int)sg.log_count + (int)SLICE_MIN_LOG_PER_RANK;
            if (worth && lo <= (int)mk.max_log) { mk.band_hi = (int)mk.max_log; mk.band_lo = lo; }
            // row-sharded columns can only be hashed share-wise: their layers must lie inside the band
            for (auto& col : cols) if (col.sliced() && ((int)col.log_size < mk.band_lo || (int)col.log_size > mk.band_hi)) throw HipError("shard group: a row-sharded column lies outside the share-wise Merkle band");
        }
        const bool banded = mk.band_hi >= mk.band_lo;
        // Storage of the levels. A share-wise level above the band's lowest holds only this rank's nodes (its children and its decommitment
        // reads are its own), addressed through a virtual base like a row-sharded column: a 2^26-row trace has ~40 GB of hashes, and eight
        // ranks each reserving all of them would not fit one GPU's — or, on eight GPUs, waste seven eighths of — memory.
        for (int log = (int)mk.max_log; log >= 0; log--) {
            const size_t stored = (size_t(1) << log) >> mk.shifts[log];
            if (banded && log > mk.band_lo && log <= mk.band_hi) {
                const size_t per_rank = stored >> sg.log_count;
                mk.layers[log] = reinterpret_cast<u32*>(reinterpret_cast<uintptr_t>(c.arena.alloc(32 * per_rank)) - 32 * per_rank * sg.rank);
            } else mk.layers[log] = (u32*)c.arena.alloc(32 * stored);
        }
        // levels [sub_hi .. 9]: one launch, a workgroup per node of level 9 — over complete, un-replicated levels only (below the band of a
        // shard group's tree); complete levels above sub_hi are single-level launches
        if (fused_top == 10 && mk.max_log >= 11) {
            u32 hi = std::min<u32>(banded ? (u32)mk.band_lo - 1 : mk.max_log, 17);
            while (hi > 10 && mk.shifts[hi] != 0) hi--;
            if (hi > 10) { p.sub_hi = hi; p.fused_top = fused_top = MERKLE_SUBTREE_ROOT_LEVEL; }     // the top starts below the subtree roots
        }
        auto level_cols = [&](int log) { return (log > 0 ? p.off[log - 1] : all.size()) - p.off[log]; };
        auto level_cost = [&](int log, double& bytes, double& comp) {
            const double nodes = (double)(1u << log), nc = (double)level_cols(log);
            const bool has = log < (int)mk.max_log;
            bytes += nodes * ((has ? 64.0 : 0.0) + 32.0) + p.bytes[log];
            comp += nodes * ((has ? 1.0 : 0.0) + (double)(((u32)nc + 15) / 16) + ((!has && nc == 0) ? 1.0 : 0.0));
        };
        if (banded && !p.poseidon && fused_top > 0 && mk.band_hi > mk.band_lo && c.shard.band_fusion) {
            // The shares of the band's last levels are 2^14..2^17 nodes: 64..512 workgroups each, four launches of ~6-10 us for ~2 us of work
            // each (r04: 50 such launches per rank and fib19 proof, 0.35 ms per rank). Levels band_lo + 3 .. band_lo go into one launch.
            const int hi = std::min(mk.band_hi, mk.band_lo + 3);
            bool plain = true;
            for (int lg = mk.band_lo; lg <= std::min(hi + 1, (int)mk.max_log); lg++) plain = plain && mk.shifts[lg] == 0;
            const u32 r = 8 - (u32)(hi - mk.band_lo);
            if (plain && (u32)mk.band_lo >= sg.log_count + r) {
                p.band_fuse_hi = hi; p.band_fuse_r = r;
                for (int lg = hi; lg >= mk.band_lo; lg--) level_cost(lg, p.band_bytes, p.band_comp);
                p.band_bytes /= sg.count; p.band_comp /= sg.count;
            }
        }
        for (int log = (int)fused_top - 1; log >= 0; log--) level_cost(log, p.top_bytes, p.top_comp);
        if (p.sub_hi) for (int log = (int)p.sub_hi; log >= (int)MERKLE_SUBTREE_ROOT_LEVEL; log--) level_cost(log, p.sub_bytes, p.sub_comp);
        p.d_all = all.empty() ? nullptr : c.stage(all.data(), all.size());
        if (fused_top > 0) {
            MerkleTreeDesc td{};
            if (mk.max_log >= 32) throw HipError("merkle: tree too deep");
            for (u32 lg = 0; lg <= mk.max_log; lg++) { td.layers[lg] = (uint4*)mk.layers[lg]; td.shifts[lg] = mk.shifts[lg]; td.col_off[lg] = (u32)p.off[lg]; }
            td.cols = p.d_all; td.n_cols = (u32)all.size(); td.max_log = mk.max_log;
            p.tree = td;
        }
        return p;
    }
    // waits: before the level `level` (and everything below it) is hashed the stream waits for `ev` — the columns of that size are produced
    // on another stream while the larger layers are being hashed. Sorted by descending level.
    struct LevelWait { int level; hipEvent_t ev; };
    // stamp_slot >= 0 (with a pinned root written by the top kernel itself): the top kernel also writes the proof's number into that stamp slot
    // behind the root; *stamped tells the caller whether it did (a tree without a fused top needs a k_post_stamp launch instead)
    DevMerkle merkle_run(MerklePlan& p, Hash32* pinned_root = nullptr, bool no_readback = false, const ChannelStep* step = nullptr, const std::vector<LevelWait>* waits = nullptr,
                         int stamp_slot = -1, bool* stamped = nullptr) {
        DevMerkle& mk = p.mk;
        const ShardGroup& sg = c.shard;
        const bool poseidon = p.poseidon;
        const u32 fused_top = p.fused_top;
        if (poseidon && step) throw HipError("the device-side channel step is a Blake2s path");
        const char* layer_kernel = poseidon ? "k_merkle_layer_poseidon" : "k_merkle_layer";
        prof_run_begin(c.stream, layer_kernel);
        const int single_lo = p.sub_hi ? (int)p.sub_hi + 1 : (int)fused_top;
        size_t wi = 0;
        auto apply_waits = [&](int log) { while (waits && wi < waits->size() && (*waits)[wi].level >= log) BF_HIP(hipStreamWaitEvent(c.stream, (*waits)[wi++].ev, 0)); };
        for (int log = (int)mk.max_log; log >= single_lo; log--) {
            apply_waits(log);
            if (log == p.band_fuse_hi) {
                // the rest of the band in one launch over this rank's share, then the all-gather of its lowest level
                prof_run_end(c.stream);
                const u32 n_wg = ((1u << mk.band_lo) >> sg.log_count) >> p.band_fuse_r;
                merkle_subtree_share(c.stream, p.tree, (u32)log, (u32)mk.band_lo, (u32)mk.band_lo - p.band_fuse_r, sg.rank * n_wg, n_wg, c.conv.merkle_node_hash, p.band_bytes, p.band_comp);
                sg.comm->all_gather(c.stream, mk.layers[mk.band_lo], (size_t(32) << mk.band_lo) >> sg.log_count);
                prof_run_begin(c.stream, layer_kernel);
                log = mk.band_lo;
                continue;
            }
            size_t n = (log > 0 ? p.off[log - 1] : p.n_all) - p.off[log];
            const bool share = log >= mk.band_lo && log <= mk.band_hi;
            const u32 per_rank = share ? ((1u << (log - mk.shifts[log])) >> sg.log_count) : 0u;   // in stored slots
            if (poseidon)
                merkle_layer_poseidon(c.stream, mk.layers[log], log < (int)mk.max_log ? mk.layers[log + 1] : nullptr, n ? p.d_all + p.off[log] : nullptr, (u32)n, (u32)log,
                                      mk.shifts[log], log < (int)mk.max_log ? mk.shifts[log + 1] : 0, sg.rank * per_rank, per_rank);
            else
                merkle_layer(c.stream, mk.layers[log], log < (int)mk.max_log ? mk.layers[log + 1] : nullptr, n ? p.d_all + p.off[log] : nullptr, (u32)n, (u32)log, p.bytes[log],
                             mk.shifts[log], log < (int)mk.max_log ? mk.shifts[log + 1] : 0, c.conv.merkle_node_hash, sg.rank * per_rank, per_rank);
            if (share && log == mk.band_lo) {
                // the smallest share-wise layer is completed on every rank by one all-gather on the device buffer (rank r's block = its
                // contiguous node range); the levels below are hashed redundantly, so every rank obtains the same root
                prof_run_end(c.stream);
                sg.comm->all_gather(c.stream, mk.layers[log], ((size_t(32) << log) >> mk.shifts[log]) >> sg.log_count);
                prof_run_begin(c.stream, layer_kernel);
            }
        }
        prof_run_end(c.stream);
        apply_waits(0);
        if (p.sub_hi) merkle_subtree(c.stream, p.tree, p.sub_hi, c.conv.merkle_node_hash, p.sub_bytes, p.sub_comp);
        // a deferred root goes to its pinned slot by the top kernel's own stores (no copy command behind the tree)
        u32* root_direct = (fused_top > 0 && !step && pinned_root) ? reinterpret_cast<u32*>(c.small_alias(pinned_root)) : nullptr;
        const bool stamp_here = root_direct && stamp_slot >= 0;
        if (stamped) *stamped = stamp_here;
        if (fused_top > 0) merkle_top(c.stream, p.tree, fused_top - 1, c.conv.merkle_node_hash, step ? step->chan : nullptr, step ? step->alpha8 : nullptr, step ? step->root_copy : root_direct, p.top_bytes, p.top_comp,
                                      stamp_here ? c.small_alias(c.stamp_host(stamp_slot)) : nullptr, c.proof_seq);
        else if (step) channel_mix_root_draw(c.stream, step->chan, mk.layers[0], step->alpha8, step->root_copy);
        BF_HIP(hipGetLastError());
        if (no_readback) return mk;
        if (pinned_root) { if (!root_direct) BF_HIP(hipMemcpyAsync(pinned_root->b, mk.layers[0], 32, hipMemcpyDeviceToHost, c.stream)); return mk; }
        c.read_back(mk.root.b, mk.layers[0], 32);
        return mk;
    }
    DevMerkle merkle_commit(const std::vector<DCol>& cols_in, Hash32* pinned_root = nullptr, bool no_readback = false, const ChannelStep* step = nullptr) {
        c.stage_checkpoint();
        StageBatch sb(c);
        MerklePlan p = merkle_plan(cols_in);
        sb.end();
        return merkle_run(p, pinned_root, no_readback, step);
    }

    // MerkleProver::decommit — control flow on the host, data through one gather.
    // Requests go into the shared gather `g`; the returned closure fills the outputs once the gathered words are available.
    typedef std::function<void(const std::vector<u32>&)> Finisher;
    Finisher decommit(Gather& g, const DevMerkle& mk, const std::vector<DCol>& cols_in, const std::map<u32, std::vector<size_t>>& queries_per_log,
                      std::vector<u32>* queried_values, MerkleDecommitment* dec) {
        // host time matters here (the GPU is idle while the decommitment is planned): no per-layer allocations, no copy when the
        // columns already come in descending size order
        auto by_size = [](const DCol& a, const DCol& b) { return a.log_size > b.log_size; };
        std::vector<DCol> sorted_copy;
        const bool kept = mk.cols.size() == cols_in.size();          // the sorted list of the commitment (same columns, same stable order)
        if (!kept && !std::is_sorted(cols_in.begin(), cols_in.end(), by_size)) { sorted_copy = cols_in; std::stable_sort(sorted_copy.begin(), sorted_copy.end(), by_size); }
        const std::vector<DCol>& cols = kept ? mk.cols : sorted_copy.empty() ? cols_in : sorted_copy;
        struct Slot { int kind; size_t first; };   // kind 0: hash witness (8 words), 1: column witness, 2: queried value
        std::vector<Slot> slots;
        slots.reserve(64 + 4 * cols.size());
        std::vector<DCol> lc;
        lc.reserve(cols.size());
        size_t ci = 0;
        std::vector<size_t> last, total;
        static const std::vector<size_t> empty;
        for (int log = (int)mk.max_log; log >= 0; log--) {
            lc.clear();
            while (ci < cols.size() && cols[ci].log_size == (u32)log) lc.push_back(cols[ci++]);
            const u32* prev_hashes = log < (int)mk.max_log ? mk.layers[log + 1] : nullptr;
            const u32 prev_shift = log < (int)mk.max_log ? mk.shifts[log + 1] : 0;
            auto it = queries_per_log.find((u32)log);
            const std::vector<size_t>& colq = it == queries_per_log.end() ? empty : it->second;
            total.clear();
            size_t pi = 0, qi = 0;
            while (pi < last.size() || qi < colq.size()) {
                size_t node;
                if (pi < last.size() && qi < colq.size()) node = std::min(last[pi] / 2, colq[qi]);
                else if (pi < last.size()) node = last[pi] / 2;
                else node = colq[qi];
                if (prev_hashes) {
                    for (size_t child = 2 * node; child <= 2 * node + 1; child++) {
                        if (pi < last.size() && last[pi] == child) pi++;
                        else {
                            // in a shard group a hash of a share-wise layer is held by one rank only; the others request a zero
                            const bool shared_layer = log + 1 > mk.band_lo && log + 1 <= mk.band_hi;
                            const bool mine = !shared_layer || (child >> (log + 1 - c.shard.log_count)) == c.shard.rank;
                            slots.push_back({0, g.add_hash(prev_hashes, child >> prev_shift, mine)});
                        }
                    }
                }
                bool queried = qi < colq.size() && colq[qi] == node;
                if (queried) qi++;
                for (auto& col : lc) { size_t f = g.add_col(col, node, c.shard.rank); slots.push_back({queried ? 2 : 1, f}); }
                total.push_back(node);
            }
            std::swap(last, total);
        }
        return [slots = std::move(slots), queried_values, dec](const std::vector<u32>& data) {
            for (auto& s : slots) {
                if (s.kind == 0) { Hash32 h; memcpy(h.b, &data[s.first], 32); dec->hash_witness.push_back(h); }
                else if (s.kind == 1) dec->column_witness.push_back(data[s.first]);
                else if (queried_values) queried_values->push_back(data[s.first]);
            }
        };
    }

    // ---- shard group (one proof over several GPUs): which columns are cut into row ranges ---------------------------------------------
    // A full-size column of the preprocessed / interaction / composition trees, a quotient column or an FRI layer with at least 2^14 rows per rank is
    // ROW-sharded: rank r holds rows [r * 2^(log - lc), (r + 1) * 2^(log - lc)) — a contiguous range of a bit-reversed circle domain, i.e.
    // a sub-coset, so Merkle subtrees, offset-0 masks, quotient rows and FRI sibling pairs are all local. Smaller columns, the 16x-replicated
    // (row-granular) columns and the preprocessed / main trees stay complete on every rank.
    // 2^14 rows per rank: below that a transform, a fold or a subtree costs less than the latency of the exchange that would divide it
    static constexpr u32 SLICE_MIN_LOG_PER_RANK = 14;
    // a Merkle layer is hashed share-wise while a rank's share has at least 2^14 stored nodes (64 workgroups); see merkle_plan
    static constexpr u32 SHARE_MIN_LOG_PER_RANK = 14;
    bool sharded() const { return c.shard.count > 1; }
    u32 lc() const { return c.shard.log_count; }
    bool slice_log(u32 log) const { return sharded() && log >= lc() + SLICE_MIN_LOG_PER_RANK; }
    // Shard policy "replicate the transforms" (bfhip_ctx_set_shard_policy, r06): every rank interpolates and extends EVERY column itself and evaluates the
    // constraints on every row — no column -> row exchange, no rows -> columns exchange of the composition accumulators — while the Merkle band, the quotient
    // rows and the FRI folds stay divided by row range through VIRTUALLY sliced columns: DCol::lc is set and the storage is the whole column, so the virtual
    // base of DCol is the real base. What is exchanged shrinks to the per-tree all-gather of 256 nodes per rank and the max-reduces. For groups whose
    // exchange would cross ONE xGMI link (N = 2: 1.03 GB per proof and rank at 76 GB/s = 13.5 ms against 5.6 ms of transforms) — DESIGN.md section 7.
    bool replicate() const { return sharded() && c.shard_replicate; }
    // does this rank hold the coefficients / compute the LDE of a polynomial with this owner entry?
    bool transforms_here(u32 owner) const { return owner == OWNER_ALL || owner == c.shard.rank || replicate(); }
    // A 16x-replicated (row-granular, shift = 4) column is cut into row ranges when a rank's range still holds 2^14 STORED words — then every
    // layer it enters lies inside the share-wise Merkle band (merkle_plan) and its rows' constraint / quotient launches are range-restricted.
    bool slice_col(u32 log, u32 shift) const { return sharded() && log >= shift + lc() + SLICE_MIN_LOG_PER_RANK; }
    size_t slice_cells(u32 log) const { return size_t(1) << (log - lc()); }
    size_t slice_first(u32 log) const { return (size_t)c.shard.rank << (log - lc()); }
    // storage for this rank's row range of a 2^log column (stored at one word per 2^shift rows), returned as a virtual base (see DCol)
    u32* alloc_slice(u32 log, u32 shift = 0) {
        return reinterpret_cast<u32*>(reinterpret_cast<uintptr_t>(c.alloc_u32(slice_cells(log) >> shift)) - sizeof(u32) * (slice_first(log) >> shift));
    }
    // Column-sharding of the transforms: the biggest column goes to the least loaded rank (greedy by 2^log, deterministic on every rank).
    std::vector<u32> assign_owners(const std::vector<DCol>& polys, u32 log_blowup) const {
        std::vector<u32> owner(polys.size(), OWNER_ALL);
        if (!sharded()) return owner;
        std::vector<size_t> idx;
        for (size_t i = 0; i < polys.size(); i++) if (slice_col(polys[i].log_size + log_blowup, polys[i].shift)) idx.push_back(i);
        // by transform work = stored words (a replicated column is a 16x smaller transform)
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return polys[a].log_size - polys[a].shift > polys[b].log_size - polys[b].shift; });
        std::vector<u64> load(c.shard.count, 0);
        for (size_t i : idx) {
            u32 best = 0;
            for (u32 r = 1; r < c.shard.count; r++) if (load[r] < load[best]) best = r;
            owner[i] = best; load[best] += u64(1) << (polys[i].log_size - polys[i].shift);
        }
        return owner;
    }

    // CommitmentTreeProver::new: LDE by the blowup factor, Merkle, mix_root.
    // Shard group: a polynomial with t.owner[i] != OWNER_ALL is extended by its owner only; one grouped send-receive then hands every rank
    // its row range of the LDE column (with_prev: and of the column's previous-row copy, which the constraint kernel needs for the mask
    // offset -1 of the last logUp column — that neighbour is a reflection in bit-reversed storage, not a halo).
    void commit_tree(DTree& t, Hash32* pinned_root = nullptr, bool with_prev = false) {
        const size_t n = t.polys.size();
        if (t.owner.size() != n) t.owner.assign(n, OWNER_ALL);
        t.evals.resize(n); t.prev.assign(n, DCol());
        std::vector<DCol> fsrc, fdst, full(n), fullprev(n);
        for (size_t i = 0; i < n; i++) {
            DCol e; e.log_size = t.polys[i].log_size + cfg.log_blowup; e.shift = t.polys[i].shift;
            if (t.owner[i] == OWNER_ALL) { e.ptr = c.alloc_u32(e.stored()); fsrc.push_back(t.polys[i]); fdst.push_back(e); }
            else if (replicate()) {
                // virtually sliced: the whole column is here (virtual base = real base), consumers are restricted to this rank's row range by `lc`;
                // the mask offset -1 of a last logUp column is read from the column itself (prev stays null)
                e.ptr = c.alloc_u32(e.stored()); fsrc.push_back(t.polys[i]); fdst.push_back(e);
                e.lc = lc();
            } else {
                if (t.owner[i] == c.shard.rank) {
                    full[i] = e; full[i].ptr = c.alloc_u32(e.stored()); fsrc.push_back(t.polys[i]); fdst.push_back(full[i]);
                    if (with_prev && e.shift == 0) { fullprev[i] = e; fullprev[i].ptr = c.alloc_u32(e.stored()); }
                }
                e.lc = lc(); e.ptr = alloc_slice(e.log_size, e.shift);
                // previous-row copies: of the full-size columns only (the last logUp column of each component; a replicated column has no
                // mask offset -1)
                if (with_prev && e.shift == 0) { t.prev[i] = e; t.prev[i].ptr = alloc_slice(e.log_size); }
            }
            t.evals[i] = e;
        }
        // the grouped send-receive that hands every rank its row range of the owned columns `idx` (after the owner's previous-row copies)
        auto exchange_columns = [&](hipStream_t comm_stream, const std::vector<size_t>& idx, hipEvent_t after_copies) {
            std::vector<Xfer> sends, recvs;
            for (size_t i : idx) {
                const u32 el = t.evals[i].log_size, sh = t.evals[i].shift;
                const size_t cells = slice_cells(el) >> sh, bytes = cells * sizeof(u32), first = slice_first(el) >> sh;      // in stored words
                const bool wp = with_prev && sh == 0;
                if (t.owner[i] == c.shard.rank) {
                    if (wp) prev_row_copy(c.stream, fullprev[i].ptr, full[i].ptr, t.polys[i].log_size);
                    for (u32 r = 0; r < c.shard.count; r++) {
                        sends.push_back({r, full[i].ptr + r * cells, bytes});
                        if (wp) sends.push_back({r, fullprev[i].ptr + r * cells, bytes});
                    }
                }
                recvs.push_back({t.owner[i], t.evals[i].ptr + first, bytes});
                if (wp) recvs.push_back({t.owner[i], t.prev[i].ptr + first, bytes});
            }
            if (recvs.empty()) return;
            if (comm_stream != c.stream) { BF_HIP(hipEventRecord(after_copies, c.stream)); BF_HIP(hipStreamWaitEvent(comm_stream, after_copies, 0)); }
            c.shard.comm->exchange(comm_stream, sends, recvs);
        };
        std::vector<size_t> owned;
        u32 big = 0;
        for (size_t i = 0; i < n; i++) if (t.owner[i] != OWNER_ALL && !replicate()) { owned.push_back(i); big = std::max(big, t.evals[i].log_size); }
        std::vector<size_t> first_wave, second_wave;
        for (size_t i : owned) (t.evals[i].log_size == big ? first_wave : second_wave).push_back(i);
        // bfhip_ctx_set_overlap bit 2 (shard groups): the largest size class is transformed first and travels on the partner stream while the
        // remaining columns are being transformed; the second send-receive follows on the same partner stream (every rank issues the group's
        // collectives in one order), and the main stream resumes behind both. Needs at least two size classes among the owned columns.
        if (sharded() && c.exchange_overlapped() && !second_wave.empty() && c.aux[0]) {
            std::vector<DCol> sa, da, sb2, db2;
            {
                size_t k = 0;   // fsrc / fdst hold, in index order, every column this rank transforms
                for (size_t i = 0; i < n; i++) {
                    const bool mine = t.owner[i] == OWNER_ALL || t.owner[i] == c.shard.rank;
                    if (!mine) continue;
                    const bool wave_a = t.owner[i] != OWNER_ALL && t.evals[i].log_size == big;
                    (wave_a ? sa : sb2).push_back(fsrc[k]); (wave_a ? da : db2).push_back(fdst[k]);
                    k++;
                }
            }
            c.stage_checkpoint();
            FftPlan pa = fft_prepare(false, sa, da), pb = fft_prepare(false, sb2, db2);
            try {
                fft_launch(pa);
                exchange_columns(c.aux[0], first_wave, c.next_event());
                fft_launch(pb);
                exchange_columns(c.aux[0], second_wave, c.next_event());
                hipEvent_t done = c.next_event();
                BF_HIP(hipEventRecord(done, c.aux[0]));
                BF_HIP(hipStreamWaitEvent(c.stream, done, 0));
            } catch (...) {
                // copies and receives still queued on the partner stream write into arena memory the next proof hands out again
                (void)hipStreamSynchronize(c.aux[0]);
                throw;
            }
        } else {
            fft_cols(false, fsrc, fdst);
            if (sharded() && !replicate()) exchange_columns(c.stream, owned, nullptr);
        }
        BF_HIP(hipGetLastError());
        t.mk = merkle_commit(t.evals, pinned_root);
        if (!pinned_root) ch.mix_root(t.mk.root);
    }

    // One process per proof: the same commitment with its two bounds overlapped. The transforms are HBM-bound, the Blake2s layers
    // VALU-bound, and layer L of a mixed-degree tree needs only the columns of size L and layer L + 1 — so the largest size class is
    // transformed first and its layers are hashed on the partner stream while the smaller classes are still being transformed.
    // interp_src (optional, one entry per polynomial): evaluations still to be interpolated into t.polys (extend_evals), wave by wave.
    // stamp_slot >= 0: behind the root (in its pinned slot) the proof's number is written into that stamp slot (ctx.h: wait_stamp)
    void commit_tree_overlapped(DTree& t, Hash32* pinned_root, const std::vector<DCol>* interp_src = nullptr, int stamp_slot = -1) {
        const size_t n = t.polys.size();
        t.owner.assign(n, OWNER_ALL);
        t.evals.resize(n); t.prev.assign(n, DCol());
        u32 max_log = 0;
        for (size_t i = 0; i < n; i++) {
            DCol e; e.log_size = t.polys[i].log_size + cfg.log_blowup; e.shift = t.polys[i].shift; e.ptr = c.alloc_u32(e.stored());
            t.evals[i] = e; max_log = std::max(max_log, e.log_size);
        }
        std::vector<DCol> src[2], pol[2], ev[2];
        u32 next_log = 0;                       // largest size among the second wave
        for (size_t i = 0; i < n; i++) {
            const int w = t.evals[i].log_size == max_log ? 0 : 1;
            if (interp_src) src[w].push_back((*interp_src)[i]);
            pol[w].push_back(t.polys[i]); ev[w].push_back(t.evals[i]);
            if (w) next_log = std::max(next_log, t.evals[i].log_size);
        }
        // below ~2^18 leaves the whole tree is a latency chain: nothing to hide, one stream
        const bool overlap = (c.overlap & 1u) && !ev[1].empty() && max_log >= 19;
        c.stage_checkpoint();
        FftPlan fi[2], fe[2];
        MerklePlan mp;
        {
            StageBatch sb(c);
            for (int w = 0; w < 2; w++) {
                if (interp_src) fi[w] = fft_prepare(true, src[w], pol[w]);
                fe[w] = fft_prepare(false, pol[w], ev[w]);
            }
            mp = merkle_plan(t.evals);
            sb.end();
        }
        hipStream_t main = c.stream, aux = c.aux_of(main);
        if (interp_src) fft_launch(fi[0]);
        fft_launch(fe[0]);
        if (!overlap) {
            if (interp_src) fft_launch(fi[1]);
            fft_launch(fe[1]);
            bool stamped = false;
            t.mk = merkle_run(mp, pinned_root, false, nullptr, nullptr, stamp_slot, &stamped);
            if (stamp_slot >= 0 && !stamped) c.post_stamp(stamp_slot);
        } else {
            hipEvent_t e1 = c.next_event(), e2 = c.next_event(), e3 = c.next_event();
            BF_HIP(hipEventRecord(e1, main));
            if (interp_src) fft_launch(fi[1]);
            fft_launch(fe[1]);
            BF_HIP(hipEventRecord(e2, main));
            BF_HIP(hipStreamWaitEvent(aux, e1, 0));
            std::vector<LevelWait> waits = {{(int)next_log, e2}};
            c.stream = aux;
            try { t.mk = merkle_run(mp, nullptr, /*no_readback=*/true, nullptr, &waits); } catch (...) { c.stream = main; (void)hipStreamSynchronize(aux); throw; }
            c.stream = main;
            BF_HIP(hipEventRecord(e3, aux));
            BF_HIP(hipStreamWaitEvent(main, e3, 0));      // joined: whatever follows on this stream sees the tree
            if (pinned_root) BF_HIP(hipMemcpyAsync(pinned_root->b, t.mk.layers[0], 32, hipMemcpyDeviceToHost, main));
            else c.read_back(t.mk.root.b, t.mk.layers[0], 32);
            if (stamp_slot >= 0) c.post_stamp(stamp_slot);
        }
        if (!pinned_root) ch.mix_root(t.mk.root);
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // Host table build + upload (outside the metric's timed region: "inputs already resident in HBM").
    // Prover-input preparation from the VM trace. on_gpu (default): the 13 table builders run on the device (tables.hip);
    // otherwise the host builders (host/tables.h) fill the columns and they are uploaded.
    // use_arena: take the column storage from the per-proof arena (no hipMalloc, which would synchronise the device) — only valid for
    // the duration of the current proof; otherwise the columns live in their own device allocations owned by `in`.
    static void upload_trace(Ctx& c, const std::vector<Registers>& vm_trace, const std::vector<u32>& code, TraceInput& in, bool use_arena = false, bool on_gpu = true) {
        in.rows.assign(N_COMPONENTS, {});
        in.n_steps = vm_trace.size();
        in.main_cells = in.interaction_cells = 0;
        auto alloc = [&](size_t words) -> u32* {
            if (use_arena) return c.alloc_u32(words);
            u32* p = nullptr; BF_HIP(hipMalloc((void**)&p, (words ? words : 1) * sizeof(u32))); in.owned.push_back(p); return p;
        };
        if (on_gpu) {
            std::vector<u32> soa[7];
            size_t n = vm_trace.size();
            for (auto& v : soa) v.resize(n);
            for (size_t i = 0; i < n; i++) { const Registers& r = vm_trace[i]; soa[0][i] = r.clk; soa[1][i] = r.ip; soa[2][i] = r.ci; soa[3][i] = r.ni; soa[4][i] = r.mp; soa[5][i] = r.mv; soa[6][i] = r.mvi; }
            std::vector<std::vector<u32*>> cols;
            build_tables_device(c, soa, (u32)n, code, alloc, cols, in.log_sizes);
            for (int k = 0; k < N_COMPONENTS; k++)
                for (u32 j = 0; j < n_main_cols(k); j++) { DCol r; r.log_size = in.log_sizes[k]; r.shift = LOG_N_LANES; r.ptr = cols[k][j]; in.rows[k].push_back(r); }
        } else {
            std::vector<Table> tables = build_tables(vm_trace, code);
            for (int k = 0; k < N_COMPONENTS; k++) {
                in.log_sizes[k] = tables[k].log_size();
                for (u32 j = 0; j < n_main_cols(k); j++) {
                    DCol r; r.log_size = in.log_sizes[k]; r.shift = LOG_N_LANES;
                    r.ptr = alloc(r.stored());
                    BF_HIP(hipMemcpyAsync(r.ptr, tables[k].cols[j].data(), r.stored() * sizeof(u32), hipMemcpyHostToDevice, c.stream));
                    in.rows[k].push_back(r);
                }
            }
        }
        for (int k = 0; k < N_COMPONENTS; k++) {
            in.main_cells += (u64)n_main_cols(k) << in.log_sizes[k];
            in.interaction_cells += (u64)(4 * n_logup_cols(k)) << in.log_sizes[k];
        }
        c.sync();
    }

    // Phase 0 of prove_brainfuck: the preprocessed tree IsFirst(LOG_MAX_ROWS ..= LOG_N_LANES) (mod.rs:495-500) — polynomials in closed form,
    // LDE, Merkle tree; the root goes to *pinned_root behind the tree (no host wait). Storage from the context's current arena.
    void build_preprocessed(DTree& tree, Hash32* pinned_root) {
        for (u32 log = log_max_rows; log >= LOG_N_LANES; log--) {
            DCol p; p.log_size = log; p.shift = 0; p.ptr = nullptr;
            tree.polys.push_back(p);
        }
        // shard group: the big IsFirst columns are column-sharded like the interaction tree's (owner interpolates and extends,
        // every rank receives its row range of the LDE)
        tree.owner = assign_owners(tree.polys, cfg.log_blowup);
        // interpolate(gen_is_first(log)) for every size in closed form, one launch (fft.hip: k_is_first_coeffs)
        IsFirstCols ifc{}; ifc.log_min = LOG_N_LANES; ifc.log_max = log_max_rows;
        if (log_max_rows - LOG_N_LANES >= 28) throw HipError("log_max_rows too large");
        for (size_t i = 0; i < tree.polys.size(); i++) {
            if (!transforms_here(tree.owner[i])) continue;
            DCol& p = tree.polys[i];
            p.ptr = c.alloc_u32(p.stored());
            ifc.ptr[p.log_size - LOG_N_LANES] = p.ptr;
        }
        is_first_coeffs(c.stream, ifc, c.d_itw, c.tw_root_log);
        if (sharded()) commit_tree(tree, pinned_root); else commit_tree_overlapped(tree, pinned_root);
    }
    // The pool's builder (pool.hip): commits the preprocessed tree on this (otherwise idle) context for every proof of a batch and returns
    // without waiting — sp.ready is recorded behind the tree and the root's store into pinned memory.
    void build_shared_preprocessed(SharedPreprocessed& sp) {
        if (sharded()) throw HipError("pool: the builder context must not be a member of a shard group");
        if (log_max_rows < LOG_N_LANES) throw HipError("log_max_rows must be at least LOG_N_LANES (4)");
        if (log_max_rows + cfg.log_blowup + 1 > c.tw_root_log + 1) throw HipError("context twiddle tree too small for log_max_rows");
        sp.valid = false;
        c.sync();                       // nothing of an earlier batch's build is in flight (its readers are done: the pool's batches are serial)
        c.arena.reset(); c.stage_used = 0; c.use_mailbox = false;
        BF_HIP(hipMemsetAsync(c.d_counters, 0, 4 * 64 * sizeof(u32), c.stream));
        sp.tree = DTree();
        Hash32* root = reinterpret_cast<Hash32*>(c.h_small);
        build_preprocessed(sp.tree, root);
        BF_HIP(hipEventRecord(sp.ready, c.stream));
        sp.pinned_root = root; sp.lmr = log_max_rows; sp.node_conv = c.conv.merkle_node_hash; sp.channel = c.conv.merkle_channel; sp.valid = true;
    }

    BrainfuckProof prove(const TraceInput& in) { return prove([&]() -> const TraceInput& { return in; }); }

    // get_input() runs on the host AFTER the (trace-independent) preprocessed phase has been enqueued, so a caller that still has to
    // run the VM and build the tables overlaps that host work with GPU work (bfhip_prove_brainfuck does).
    BrainfuckProof prove(const std::function<const TraceInput&()>& get_input) {
        double t_start = now();
        mark_t0 = t_start;
        struct SpinScope { Ctx& c; double saved; ~SpinScope() { c.spin_seconds = saved; } } spin_scope{c, c.spin_seconds};
        if (!c.sync_blocking) c.spin_seconds = 8e-3;      // bfhip_ctx_set_sync_policy(blocking): hosts with more contexts than cores keep the short poll
        c.arena.reset();
        // shard policy of this proof (every rank of a group resolves it the same way: same setting, same group size, and "spans several GPUs" is a
        // rendezvous decision): -1 = automatic = replicate the transforms for a group of TWO ranks on different GPUs (one xGMI link would carry 1 GB)
        c.shard_replicate = sharded() && (c.shard_policy == 1 || (c.shard_policy < 0 && c.shard.count == 2 && c.shard.comm && c.shard.comm->spans_devices()));
        // Mailboxes (mailbox.hip): one process per proof only — a shard group's exchanges are rendezvous points of their own. The ring must
        // not need recycling while a mailbox kernel waits for this thread, so it is recycled here, where nothing of this context is in flight.
        // Not by default under the blocking sync policy either: a mailbox kernel spins on the GPU until this thread posts, and a host that
        // asked for sleeping waits is one whose threads may be descheduled for long (BFHIP_MAILBOX=1 still forces it; wait_stamp then sleeps).
        c.use_mailbox = c.mailbox_mode < 0 ? (log_max_rows <= 21 && !c.sync_blocking) : c.mailbox_mode != 0;
        // At most ONE proof of the process runs in the mailbox order at a time (r05). HIP multiplexes the streams of all contexts onto a few
        // hardware queues; with two proofs in that order, A's stamp-producing kernels can sit behind B's spinning mailbox kernel on a shared
        // queue while B's sit behind A's — each host thread waits for a stamp that cannot be written until the other posts: both give up after
        // BFHIP_MAILBOX_TIMEOUT_MS (seen with three 2^20-row proofs in flight, profiles/r05_bench.json of the first pass). A proof that does not
        // get the token keeps the order wait -> draw -> copy -> launch: behind one spinning kernel it is merely later, never stuck.
        // The token is per GPU (hardware queues are per device), and a proof that unwinds on an error keeps it until its streams have drained:
        // its mailbox kernels may still be spinning (the Mailbox destructors, which run first, post them) and must not meet another proof's.
        struct MailboxToken {
            Ctx& c; bool held = false; int entered = std::uncaught_exceptions();
            explicit MailboxToken(Ctx& c_) : c(c_) {}
            static std::atomic<int>& flag(int device) { static std::atomic<int> f[64]; return f[device & 63]; }
            bool acquire() { int z = 0; held = flag(c.device).compare_exchange_strong(z, 1); return held; }
            ~MailboxToken() {
                if (!held) return;
                if (std::uncaught_exceptions() > entered) for (hipStream_t st : {c.stream, c.id_main, c.stream2}) if (st) (void)hipStreamSynchronize(st);
                flag(c.device).store(0);
            }
        } mailbox_token(c);
        if (c.use_mailbox && !sharded() && !(c.overlap & 2u) && !mailbox_token.acquire()) c.use_mailbox = false;
        const bool mb = c.use_mailbox && !sharded() && !(c.overlap & 2u);
        c.last_proof_flags = 0;       // completed below: what this proof actually did (bfhip_ctx_last_proof_flags)
        if (++c.proof_seq == 0) c.proof_seq = 1;
        c.reap_some();
        if (mb && c.stage_used > c.stage_bytes / 4) {
            c.sync(); if (c.stream2) BF_HIP(hipStreamSynchronize(c.stream2)); for (auto a : c.aux) if (a) BF_HIP(hipStreamSynchronize(a));
            c.stage_used = 0;
        }
        for (int k = 0; k <= 5; k++) c.mailbox_err_host()[2 * k] = 0;
        Mailbox mb_logup(c, 1), mb_constraints(c, 2), mb_samples(c, 3), mb_quot0(c, 4), mb_quot1(c, 5);
        ch = Channel(c.conv);
        if (log_max_rows < LOG_N_LANES) throw HipError("log_max_rows must be at least LOG_N_LANES (4)");
        if (log_max_rows + cfg.log_blowup + 1 > c.tw_root_log + 1) throw HipError("context twiddle tree too small for log_max_rows");
        std::vector<DTree> trees(4);
        BrainfuckProof bp;

        // ---- Phase 0: preprocessed IsFirst(LOG_MAX_ROWS ..= LOG_N_LANES) (mod.rs:495-500) ---------------------------------
        // Trace independent, and nothing on the GPU depends on its root: it is enqueued on the side stream and runs beside the
        // caller's trace preparation (get_input) and the main-trace phase; both roots are read after ONE synchronisation and mixed
        // in protocol order (root0, claim, root1).
        double t0 = now();
        PreprocessedCache& cache = preprocessed_cache_of(&c);
        Hash32* pinned_root0 = reinterpret_cast<Hash32*>(c.h_small);
        Hash32* pinned_root1 = pinned_root0 + 1;
        // a pool's batch (pool.hip): the tree its builder context commits once for all proofs of the batch
        const SharedPreprocessed* shared = c.shared_pre && c.shared_pre->matches(c, log_max_rows) ? c.shared_pre : nullptr;
        const bool reuse = shared || cache.matches(c, log_max_rows);
        BF_HIP(hipMemsetAsync(c.d_counters, 0, 4 * 64 * sizeof(u32), c.stream));      // ticket counters (a failed proof may have left one mid-count)
        BF_HIP(hipEventRecord(c.ev[0], c.stream));
        // In a shard group everything stays on the main stream: the group's exchanges are issued in one order on one stream per rank.
        const bool use_side = !sharded() && !c.single_stream;
        if (shared) trees[0] = shared->tree;        // layout only: complete (and its root known) once shared->ready has completed — awaited below
        else if (reuse) trees[0] = cache.tree;
        else {
            if (use_side) {
                c.ensure_side();
                BF_HIP(hipStreamWaitEvent(c.stream2, c.ev[0], 0));   // stream2 starts after whatever preceded this proof on the main stream
                std::swap(c.stream, c.stream2); c.side_busy = true;
            }
            try {
                if (cache.enabled) { cache.keep.reset(); std::swap(c.arena, cache.keep); }   // build the tree in memory that survives arena.reset()
                try { build_preprocessed(trees[0], pinned_root0); } catch (...) { if (cache.enabled) std::swap(c.arena, cache.keep); throw; }
                if (cache.enabled) std::swap(c.arena, cache.keep);
                BF_HIP(hipEventRecord(c.ev[1], c.stream));
            } catch (...) { if (use_side) { std::swap(c.stream, c.stream2); (void)hipStreamSynchronize(c.stream2); c.side_busy = false; } throw; }
            if (use_side) std::swap(c.stream, c.stream2);
        }
        // the side stream's work ends with ev[1]: when the event has completed there is nothing to wait for (a hipStreamSynchronize call
        // costs ~13 us even then — on the critical path of the first Fiat-Shamir round trip)
        auto join_side = [&]() {
            if (!c.side_busy) return;
            if (reuse || hipEventQuery(c.ev[1]) != hipSuccess) (void)hipStreamSynchronize(c.stream2);
            (void)hipGetLastError();
            c.side_busy = false;
        };
        const TraceInput* in_p = nullptr;
        try { in_p = &get_input(); } catch (...) { join_side(); throw; }
        const TraceInput& in = *in_p;

        // ---- Phase 1: main trace (mod.rs:506-583) -----------------------------------------------------------------------------
        const std::vector<std::vector<DCol>>& rows = in.rows;   // row-granular table columns (also feed the logUp pass)
        size_t main_off[N_COMPONENTS], inter_off[N_COMPONENTS];
        {
            size_t mo = 0, io = 0;
            for (int k = 0; k < N_COMPONENTS; k++) { main_off[k] = mo; inter_off[k] = io; mo += n_main_cols(k); io += 4 * n_logup_cols(k); }
        }
        // Everything the logUp launches need apart from the lookup elements (storage of the 60 interaction columns, the scratch of the scans,
        // the launch table) is laid out while the GPU still hashes the main-trace tree (r04: 27 us off the first Fiat-Shamir round trip).
        uint4* d_claimed = nullptr;
        CompositionPlan composition_plan;
        SamplePlan sample_plan;
        std::vector<DCol> inter_vals;
        std::vector<LogupLaunch> logups(N_COMPONENTS);
        std::function<bool(size_t)> kept = [](size_t) { return true; };
        auto prepare_logup = [&]() {
            for (int k = 0; k < N_COMPONENTS; k++) bp.log_sizes[k] = in.log_sizes[k];
            // the claimed sums: in HBM for a shard group (read back at once), else written by the scan kernel into their pinned slot
            d_claimed = sharded() ? (uint4*)c.arena.alloc(sizeof(uint4) * N_COMPONENTS) : c.small_alias(reinterpret_cast<uint4*>(c.h_small + 2048));
            // the interaction columns in commit order (mod.rs:690-702): per component its logUp columns but the last row-granular (4 coordinates
            // each), then the last one full-size
            for (int k = 0; k < N_COMPONENTS; k++) {
                const u32 log = bp.log_sizes[k], nl = n_logup_cols(k);
                for (u32 j = 0; j < 4 * nl; j++) { DCol col; col.log_size = log; col.shift = j + 4 < 4 * nl ? LOG_N_LANES : 0; inter_vals.push_back(col); }
            }
            // Shard group: the full-size columns (each component's last logUp column, 4 coordinates) are column-sharded — only the owner of a
            // coordinate column keeps, interpolates and extends it, so only the owner has the logUp kernel write it (the others pass a null
            // pointer: no storage, no store); the row-granular columns and the small ones are written and transformed by every rank.
            if (sharded()) trees[2].owner = assign_owners(inter_vals, cfg.log_blowup);
            kept = [&](size_t i) { return !sharded() || transforms_here(trees[2].owner[i]); };
            for (size_t i = 0; i < inter_vals.size(); i++) if (kept(i)) inter_vals[i].ptr = c.alloc_u32(inter_vals[i].stored());
            for (int k = 0; k < N_COMPONENTS; k++) {
                u32 log = bp.log_sizes[k], log_rows = log - LOG_N_LANES;
                size_t M = size_t(1) << log_rows;
                LogupLaunch L{};
                for (u32 j = 0; j < n_main_cols(k); j++) L.cols[j] = rows[k][j].ptr;
                u32 nl = n_logup_cols(k);
                for (u32 j = 0; j + 4 < 4 * nl; j++) L.out_rep[j] = inter_vals[inter_off[k] + j].ptr;
                for (int w = 0; w < 4; w++) L.out_last[w] = inter_vals[inter_off[k] + 4 * (nl - 1) + w].ptr;
                L.vrow = c.arena.alloc(sizeof(uint4) * M);
                L.wloc = c.arena.alloc(sizeof(uint4) * M);
                L.totals = c.arena.alloc(sizeof(uint4) * (M / 1024 + 2));
                L.claimed = d_claimed + k;
                L.log_rows = log_rows; L.comp = k;       // L.el: drawn after the main-trace root
                logups[k] = L;
            }
        };
        uint4* pinned_claimed = reinterpret_cast<uint4*>(c.h_small + 2048);
        Hash32* pinned_root2 = reinterpret_cast<Hash32*>(c.h_small + 2304);
        LogupBatch* h_lb = nullptr;           // the logUp batch in the staging ring (mailbox mode): its `el` is filled in after the draw
        auto enqueue_interaction = [&]() {
            LogupBatch lb;
            logup_batch_init(lb, Lookups{}, logups.data(), N_COMPONENTS);
            c.stage_checkpoint();
            mb_logup.begin();
            const LogupBatch* d_lb = c.stage(&lb, 1);
            mb_logup.arm();
            h_lb = mb_logup.host(d_lb);
            logup_batch_run(c.stream, d_lb, lb);          // the 13 interaction_trace_evaluation calls (mod.rs:596-687) as one batch: four launches
            BF_HIP(hipGetLastError());
            trees[2].polys = inter_vals;                  // interpolate in place
            commit_tree_overlapped(trees[2], pinned_root2, &inter_vals, 2);
        };
        try {
            for (int k = 0; k < N_COMPONENTS; k++) {
                bp.log_sizes[k] = in.log_sizes[k];
                if (bp.log_sizes[k] > log_max_rows) throw HipError("a component exceeds LOG_MAX_ROWS");
                for (u32 j = 0; j < n_main_cols(k); j++) { DCol p = rows[k][j]; p.ptr = nullptr; trees[1].polys.push_back(p); }
            }
            {
                std::vector<DCol> src;
                for (int k = 0; k < N_COMPONENTS; k++) for (auto& r : rows[k]) src.push_back(r);
                if (sharded()) {
                    // Shard group: the main-trace columns are column-sharded like the other trees' — a column of at least 2^14 stored words
                    // per rank after the extension is interpolated and extended by its owner only, which then hands every rank its row
                    // range (row-granular: a sixteenth of the bytes of a full-size column); the small ones are transformed by every rank.
                    trees[1].owner = assign_owners(trees[1].polys, cfg.log_blowup);
                    std::vector<DCol> s_mine, p_mine;
                    for (size_t i = 0; i < src.size(); i++) {
                        if (!transforms_here(trees[1].owner[i])) continue;
                        trees[1].polys[i].ptr = c.alloc_u32(trees[1].polys[i].stored());
                        s_mine.push_back(src[i]); p_mine.push_back(trees[1].polys[i]);
                    }
                    fft_cols(true, s_mine, p_mine);
                    commit_tree(trees[1], pinned_root1);
                } else {
                    for (auto& p : trees[1].polys) p.ptr = c.alloc_u32(p.stored());
                    commit_tree_overlapped(trees[1], pinned_root1, &src, mb ? 1 : -1);
                }
            }
            BF_HIP(hipEventRecord(c.ev[2], c.stream));
            mark("main tree enqueued");
            prepare_logup();                 // host work under the main tree's kernels: only the lookup elements are missing afterwards
            if (mb) {
                // the whole interaction phase goes onto the stream now, behind a mailbox that will deliver the lookup elements
                enqueue_interaction();
                mark("interaction phase enqueued behind its mailbox");
                c.wait_stamp(1);
            } else c.sync();
            mark("main root arrived");
        } catch (...) { join_side(); throw; }
        join_side();
        mark("side stream joined");
        if (shared) {
            // The batch's tree: its commitment was enqueued on the builder's stream before this proof started. Nothing of this proof has read it
            // yet; from here on the constraint, sampling, quotient and decommitment launches do, so the host waits for it here (normally long done).
            const auto w0 = std::chrono::steady_clock::now();
            for (u32 polls = 0;; polls++) {
                hipError_t e = hipEventQuery(shared->ready);
                if (e == hipSuccess) break;
                if (e != hipErrorNotReady) BF_HIP(e);
                if (polls > 64) std::this_thread::yield();
                if ((polls & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 120.0)
                    throw HipError("pool: the batch's shared preprocessed commitment did not complete");
            }
            trees[0].mk.root = *shared->pinned_root;
            mark("shared preprocessed tree ready");
        }
        if (!reuse) {
            trees[0].mk.root = *pinned_root0;
            if (cache.enabled) {
                cache.tree = trees[0]; cache.lmr = log_max_rows; cache.node_conv = c.conv.merkle_node_hash; cache.channel = c.conv.merkle_channel;
                cache.shard_rank = c.shard.rank; cache.shard_count = c.shard.count; cache.replicate = replicate(); cache.valid = true;
            }
        }
        trees[1].mk.root = *pinned_root1;
        ch.mix_root(trees[0].mk.root);
        tap("root0");
        for (int k = 0; k < N_COMPONENTS; k++) ch.mix_u64(bp.log_sizes[k]);   // claim.mix_into (mod.rs:102-116)
        ch.mix_root(trees[1].mk.root);
        tap("root1");
        // the host wall time of the two overlapped phases; split in proportion to their GPU-side durations once the events behind them have
        // certainly completed (at the end of the proof: a stamp can arrive before the event recorded behind the kernel that wrote it)
        const double wall_phase01 = now() - t0;

        // ---- Phase 2: interaction trace (mod.rs:589-723) ------------------------------------------------------------------------
        t0 = now();
        Lookups el;
        { Q31 z, a; ch.draw_two_felts(z, a); el.memory = make_lookup(z, a); }         // MemoryElements::draw
        { Q31 z, a; ch.draw_two_felts(z, a); el.instruction = make_lookup(z, a); }    // InstructionElements::draw
        { Q31 z, a; ch.draw_two_felts(z, a); el.processor = make_lookup(z, a); }      // ProcessorElements::draw
        mark("lookup elements drawn");
        for (auto& L : logups) L.el = el;
        auto take_claimed = [&](const uint4* h_claimed) {
            for (int k = 0; k < N_COMPONENTS; k++) bp.claimed_sums[k] = q_make(h_claimed[k].x, h_claimed[k].y, h_claimed[k].z, h_claimed[k].w);
            for (int k = 0; k < N_COMPONENTS; k++) ch.mix_felts(&bp.claimed_sums[k], 1);   // interaction_claim.mix_into (mod.rs:189-203)
        };
        // per tree, per column: list of point indices (Components::mask_points + composition mask); independent of the challenges
        std::vector<std::vector<std::vector<u32>>> mask(4);
        auto build_mask = [&]() {
            mask[0].assign(trees[0].polys.size(), {});
            for (int k = 0; k < N_COMPONENTS; k++) mask[0][log_max_rows - bp.log_sizes[k]] = {0};
            for (int k = 0; k < N_COMPONENTS; k++) {
                for (u32 j = 0; j < n_main_cols(k); j++) mask[1].push_back({0});
                u32 ni = 4 * n_logup_cols(k);
                // last logUp column of a component: offsets {0, -1} (LogupAtRow::finalize); which comes first is Conventions::logup_mask_order
                for (u32 j = 0; j < ni; j++) {
                    if (j + 4 >= ni) { if (c.conv.logup_mask_order == 1) mask[2].push_back({(u32)(1 + k), 0}); else mask[2].push_back({0, (u32)(1 + k)}); }
                    else mask[2].push_back({0});
                }
            }
            mask[3].assign(4, {0});
        };
        Hash32* pinned_root3 = reinterpret_cast<Hash32*>(c.h_small + 2368);
        Q31 random_coeff;
        if (mb) {
            // The logUp kernels and the interaction tree are already on the stream, behind their mailbox: hand over the lookup elements.
            h_lb->el = el;
            mb_logup.post();
            mark("lookup elements posted");
            // ---- prover::prove (mod.rs:732): the composition phase goes onto the stream behind ITS mailbox while the GPU works on the
            // interaction phase: the 13 constraint launches (coefficient powers and claimed sums still empty), the composition transforms,
            // the composition tree, a stamp
            composition_plan = composition_prepare(trees, bp, main_off, inter_off, el);
            compute_composition(trees, bp, composition_plan, q_zero(), &mb_constraints);
            build_mask();
            commit_tree_overlapped(trees[3], pinned_root3, nullptr, 3);
            sample_plan = sample_prepare(trees, mask);
            mark("composition phase enqueued behind its mailbox");
            c.wait_stamp(2);
            mark("interaction root arrived");
            take_claimed(pinned_claimed);
            trees[2].mk.root = *pinned_root2;
            ch.mix_root(trees[2].mk.root);
            tap("root2");
            tm.interaction = now() - t0;
            t0 = now();
            random_coeff = ch.draw_felt();
            composition_fill(bp, composition_plan, random_coeff);
            mb_constraints.post();
            mark("random coefficient posted");
        } else {
            {   // the 13 interaction_trace_evaluation calls (mod.rs:596-687) as one batch: four launches
                LogupBatch lb;
                logup_batch_init(lb, el, logups.data(), N_COMPONENTS);
                c.stage_checkpoint();
                logup_batch_run(c.stream, c.stage(&lb, 1), lb);
            }
            mark("logUp launched");
            BF_HIP(hipGetLastError());
            trees[2].polys = inter_vals;          // interpolate in place
            if (sharded()) {
                uint4 h_claimed[N_COMPONENTS];
                c.read_back(h_claimed, d_claimed, sizeof(h_claimed));
                take_claimed(h_claimed);
                std::vector<DCol> mine_cols;
                for (size_t i = 0; i < inter_vals.size(); i++) if (kept(i)) mine_cols.push_back(inter_vals[i]);
                fft_cols(true, mine_cols, mine_cols);
                commit_tree(trees[2], nullptr, /*with_prev=*/true);
            } else {
                // Nothing on the GPU waits for the claimed sums: they travel to their pinned slot behind the logUp kernels, the interaction tree is
                // enqueued right away, and the host mixes claim and root in protocol order after ONE synchronisation (no idle gap between the logUp
                // kernels and the transforms).
                commit_tree_overlapped(trees[2], pinned_root2, &inter_vals);
                composition_plan = composition_prepare(trees, bp, main_off, inter_off, el);      // host work under the tree's kernels
                mark("interaction tree enqueued + composition prepared");
                c.sync();
                mark("interaction root arrived");
                take_claimed(pinned_claimed);
                trees[2].mk.root = *pinned_root2;
                ch.mix_root(trees[2].mk.root);
            }
            tap("root2");
            tm.interaction = now() - t0;

            // ---- prover::prove (mod.rs:732): composition polynomial -------------------------------------------------------------
            t0 = now();
            random_coeff = ch.draw_felt();
            if (sharded()) composition_plan = composition_prepare(trees, bp, main_off, inter_off, el);
            compute_composition(trees, bp, composition_plan, random_coeff);
            mark("constraints + composition transforms launched");
            build_mask();
            if (sharded()) commit_tree(trees[3]);
            else {
                // the composition tree is enqueued; the sampling jobs (which only need the polynomials' addresses) are listed while it is hashed
                commit_tree_overlapped(trees[3], pinned_root3);
                sample_plan = sample_prepare(trees, mask);
                mark("composition tree enqueued + samples prepared");
            }
        }

        // ---- OODS sampling (a8) ------------------------------------------------------------------------------------------------------
        // sample points: index 0 = P, 1 + k = P - trace_step(component k)
        std::vector<PtQ> points(1 + N_COMPONENTS);
        PtQ oods;
        auto draw_oods = [&]() {
            Q31 t = ch.draw_felt();
            Q31 t2 = q_mul(t, t);
            Q31 d = q_inv(q_addm(t2, 1));
            oods.x = q_mul(q_sub(q_one(), t2), d);
            oods.y = q_mul(q_add(t, t), d);
            points[0] = oods;
            for (int k = 0; k < N_COMPONENTS; k++) points[1 + k] = pq_add(oods, pq_neg(to_q(index_to_point(subgroup_gen(bp.log_sizes[k])))));
        };
        auto mix_samples = [&]() {
            std::vector<Q31> flat;
            for (auto& t : bp.proof.sampled_values) for (auto& col : t) for (auto& v : col) flat.push_back(v);
            ch.mix_felts(flat.data(), flat.size());
        };
        std::vector<LevelWait> q_waits;
        std::vector<DSecure> quotients;
        if (mb) {
            // the sampling kernels behind their mailbox (the point's factor tables still empty) while the GPU evaluates the constraints
            SampleRun sr = sample_enqueue(sample_plan, points.size(), &mb_samples);
            c.post_stamp(4);
            mark("sampling enqueued behind its mailbox");
            c.wait_stamp(3);
            mark("composition root arrived");
            trees[3].mk.root = *pinned_root3;
            ch.mix_root(trees[3].mk.root);
            tap("root3");
            tm.composition = now() - t0;
            t0 = now();
            draw_oods();
            sample_fill(sr, points);
            mb_samples.post();
            mark("out-of-domain point posted");
            // the quotient kernels behind two mailboxes (largest size group; the rest): batch structure from the points, sampled values still empty
            QuotientRun qr = quotients_enqueue(trees, mask, points, &mb_quot0, &mb_quot1);
            BF_HIP(hipEventRecord(c.ev[5], c.stream));
            quotients = qr.out;
            mark("quotients enqueued behind their mailboxes");
            c.wait_stamp(4);
            sample_finish(trees, mask, bp.proof, sr);
            mark("sampled values arrived");
            mix_samples();
            tap("sampled");
            tm.oods = now() - t0;
            t0 = now();
            Q31 q_coeff = ch.draw_felt();
            mark("sampled values mixed, quotient coefficient drawn");
            quotients_fill(qr, mask, points, bp.proof, q_coeff, &mb_quot0, &mb_quot1);
            mark("quotient constants posted");
        } else {
            if (!sharded()) {
                c.sync();
                mark("composition root arrived");
                trees[3].mk.root = *pinned_root3;
                ch.mix_root(trees[3].mk.root);
            }
            tap("root3");
            tm.composition = now() - t0;
            t0 = now();
            draw_oods();
            if (sharded()) sample_plan = sample_prepare(trees, mask);
            sample(trees, mask, points, bp.proof, sample_plan);
            mark("sampled values arrived");
            mix_samples();
            tap("sampled");
            tm.oods = now() - t0;

            // ---- FRI quotients (a9) ----------------------------------------------------------------------------------------------------
            t0 = now();
            Q31 q_coeff = ch.draw_felt();
            mark("sampled values mixed, quotient coefficient drawn");
            BF_HIP(hipEventRecord(c.ev[4], c.stream));
            quotients = compute_quotients(trees, mask, points, bp.proof, q_coeff, &q_waits);
            BF_HIP(hipEventRecord(c.ev[5], c.stream));
            mark("quotients launched");
        }
        // no host wait here: the FRI phase is planned (layer storage, 26 tree layouts, one staging copy) while the quotient kernels run;
        // the phase time comes from the two events


        // ---- FRI commit (a10), proof of work (a11), decommitment (a12) -------------------------------------------------------------------
        // Sanity check of prover::prove (composition OODS value == constraints evaluated on the sampled mask values): host arithmetic on values
        // known since the sampling — done while the GPU runs the FRI commit phase, not after the proof's last kernel (r04)
        auto mailbox_gave_up = [&]() { for (int k = 1; k <= 5; k++) if (c.mailbox_err_host()[2 * k]) return true; return false; };
        const char* mailbox_msg = "a mailbox kernel gave up waiting for the host (BFHIP_MAILBOX_TIMEOUT_MS): the proof was computed from stale challenge words";
        auto sanity_check = [&]() {
            // the mailbox error words are final here (every mailbox kernel ran before stamp 4 was written): a kernel that gave up made the
            // phases run on stale challenge words, and THAT is the error to report — not the constraint mismatch it causes
            if (mb && mailbox_gave_up()) throw HipError(mailbox_msg);
            Q31 want = eval_composition_at_point(bp.log_sizes, bp.claimed_sums, log_max_rows, el, oods, bp.proof.sampled_values, random_coeff, c.conv);
            const auto& cv = bp.proof.sampled_values[3];
            std::vector<Q31> ce[4] = {cv[0], cv[1], cv[2], cv[3]};
            if (!q_eq(HostPointEval::combine(ce, 0), want)) throw HipError("ConstraintsNotSatisfied");
        };
        fri_and_decommit(trees, quotients, bp.proof, q_waits, sanity_check);
        for (int k = 1; k <= 5; k++) if (c.mailbox_err_host()[2 * k]) *c.mailbox_err_host() = c.mailbox_err_host()[2 * k];
        if (mb) {
            mark("mailbox 1 waited for the host (GPU clock, 10 ns ticks):"); if (trace_host) host_marks.back().second = host_marks[host_marks.size() - 2].second + c.mailbox_err_host()[3] * 0.01;
            mark("mailbox 2 waited"); if (trace_host) host_marks.back().second = host_marks[host_marks.size() - 2].second + c.mailbox_err_host()[5] * 0.01;
            mark("mailbox 3 waited"); if (trace_host) host_marks.back().second = host_marks[host_marks.size() - 2].second + c.mailbox_err_host()[7] * 0.01;
            mark("mailbox 4 waited"); if (trace_host) host_marks.back().second = host_marks[host_marks.size() - 2].second + c.mailbox_err_host()[9] * 0.01;
            mark("mailbox 5 waited"); if (trace_host) host_marks.back().second = host_marks[host_marks.size() - 2].second + c.mailbox_err_host()[11] * 0.01;
        }
        if (*c.mailbox_err_host()) throw HipError(mailbox_msg);
        {
            float ms0 = 0.f, ms1 = 0.f;
            if (!reuse) BF_HIP(hipEventElapsedTime(&ms0, c.ev[0], c.ev[1]));
            BF_HIP(hipEventElapsedTime(&ms1, c.ev[0], c.ev[2]));
            const double tot = (double)ms0 + (double)ms1;
            tm.preprocessed = tot > 0 ? wall_phase01 * ms0 / tot : 0.0;
            tm.main_trace = wall_phase01 - tm.preprocessed;
        }
        {
            float ms_q = 0.f;
            BF_HIP(hipEventElapsedTime(&ms_q, c.ev[4], c.ev[5]));      // both completed: fri_and_decommit ends with host waits
            tm.quotients = ms_q * 1e-3;
            tm.fri = (now() - t0) - tm.quotients;
        }

        c.last_proof_flags = (mb ? 1u : 0u) | (reuse && !shared ? 2u : 0u) | (shared ? 4u : 0u) | (replicate() ? 8u : 0u);
        tm.total = now() - t_start;
        mark("done");
        print_marks();
        return bp;
    }

    // ComponentProvers::compute_composition_polynomial + DomainEvaluationAccumulator::finalize
    // Everything about the 13 constraint launches that does not depend on the interaction phase's challenge-side results (random coefficient,
    // claimed sums): accumulators, column descriptors, vanishing inverses. Built while the GPU is still hashing the interaction tree.
    struct CompositionPlan {
        std::vector<ConstraintLaunch> launches; std::vector<DSecure> acc; std::vector<bool> have; u32 total = 0, max_log = 0;
        ConstraintLaunch* h_staged = nullptr;      // mailbox mode: the launch table in the staging ring, completed by composition_fill
    };
    CompositionPlan composition_prepare(std::vector<DTree>& trees, const BrainfuckProof& bp, const size_t* main_off, const size_t* inter_off, const Lookups& el) {
        CompositionPlan cp;
        for (int k = 0; k < N_COMPONENTS; k++) { cp.total += n_constraints(k); cp.max_log = std::max(cp.max_log, bp.log_sizes[k] + 1); }
        cp.acc.resize(cp.max_log + 1); cp.have.assign(cp.max_log + 1, false); cp.launches.resize(N_COMPONENTS);
        for (int k = 0; k < N_COMPONENTS; k++) {
            u32 log = bp.log_sizes[k], eval_log = log + 1;
            // shard group: an accumulator of a row-sharded size holds this rank's row range only (its components' interaction LDE columns
            // have the same size and are row-sharded too)
            const bool sl = slice_log(eval_log) && !replicate();      // replicate policy: every rank evaluates every row (0.7 ms of a fib19 proof) instead of exchanging rows -> columns
            if (!cp.have[eval_log]) {
                cp.acc[eval_log].log_size = eval_log; cp.acc[eval_log].lc = sl ? lc() : 0;
                for (int w = 0; w < 4; w++) cp.acc[eval_log].c[w] = sl ? alloc_slice(eval_log) : c.alloc_u32(size_t(1) << eval_log);
            }
            ConstraintLaunch L{};
            L.overwrite = cp.have[eval_log] ? 0u : 1u;      // the first component of a size writes the accumulator (no zero fill)
            cp.have[eval_log] = true;
            L.is_first = trees[0].evals[log_max_rows - log].ptr;
            for (u32 j = 0; j < n_main_cols(k); j++) L.trace[j] = trees[1].evals[main_off[k] + j].desc();
            const u32 ni = 4 * n_logup_cols(k);
            for (u32 j = 0; j < ni; j++) L.inter[j] = trees[2].evals[inter_off[k] + j].desc();
            for (int w = 0; w < 4; w++) { const DCol& pv = trees[2].prev[inter_off[k] + ni - 4 + w]; L.inter_prev[w] = pv.ptr; }   // nullptr unless row-sharded
            if (sl) { L.row0 = (u32)slice_first(eval_log); L.n_rows = (u32)slice_cells(eval_log); }
            for (int w = 0; w < 4; w++) L.acc[w] = cp.acc[eval_log].c[w];
            L.el = el; L.log_size = log;
            // denom_inv[i] = 1 / coset_vanishing(CanonicCoset(log).coset, eval_domain.at(i)), i in {0, 1} (bit-reversal of 2 entries = identity)
            for (u32 i = 0; i < 2; i++) L.denom_inv[i] = m_inv(coset_vanishing_m(log, canonic_domain_at(eval_log, i)));
            cp.launches[k] = L;
        }
        return cp;
    }
    // the challenge-side fields of the 13 launches: coefficient powers and claimed sums
    static void composition_challenge_fields(const BrainfuckProof& bp, u32 total, Q31 random_coeff, ConstraintLaunch* launches) {
        std::vector<Q31> powers(total);
        { Q31 cur = q_one(); for (u32 i = 0; i < total; i++) { powers[i] = cur; cur = q_mul(cur, random_coeff); } }
        u32 remaining = total;
        for (int k = 0; k < N_COMPONENTS; k++) {
            const u32 nc = n_constraints(k);
            // accum.columns(): this component takes the LAST nc remaining powers and uses them reversed (constraint 0 <-> highest)
            for (u32 j = 0; j < nc; j++) launches[k].coeff[j] = powers[remaining - 1 - j];
            remaining -= nc;
            launches[k].total_sum = bp.claimed_sums[k];
        }
    }
    // mailbox mode: the launch table is already in the ring and the kernels are on the stream; complete it (the caller posts)
    void composition_fill(const BrainfuckProof& bp, CompositionPlan& cp, Q31 random_coeff) {
        if (!cp.h_staged) throw HipError("composition_fill without a staged launch table");
        composition_challenge_fields(bp, cp.total, random_coeff, cp.h_staged);
    }
    // mbx != nullptr: the launches go onto the stream behind that mailbox with their challenge-side fields empty (composition_fill completes them)
    void compute_composition(std::vector<DTree>& trees, const BrainfuckProof& bp, CompositionPlan& cp, Q31 random_coeff, Mailbox* mbx = nullptr) {
        const u32 max_log = cp.max_log;
        std::vector<DSecure>& acc = cp.acc;
        std::vector<bool>& have = cp.have;
        std::vector<ConstraintLaunch>& launches = cp.launches;
        if (!mbx) composition_challenge_fields(bp, cp.total, random_coeff, launches.data());
        c.stage_checkpoint();
        {   // the 13 evaluate_constraint_quotients_on_domain calls as ONE launch (air.hip: k_constraints_batch), one staging copy
            ConstraintBatch cb;
            constraint_batch_init(cb, launches.data(), N_COMPONENTS);
            const ConstraintLaunch* d_launches; const ConstraintBatch* d_cb;
            if (mbx) {
                mbx->begin();
                d_launches = c.stage(launches.data(), launches.size());
                d_cb = c.stage(&cb, 1);
                mbx->arm();
                cp.h_staged = mbx->host(d_launches);
            } else {
                StageBatch sb(c);
                d_launches = c.stage(launches.data(), launches.size());
                d_cb = c.stage(&cb, 1);
                sb.end();
            }
            eval_constraints_batch(c.stream, d_cb, cb, d_launches);
        }
        BF_HIP(hipGetLastError());
        // finalize (DomainEvaluationAccumulator::finalize): ascending sizes; the reference evaluates the running polynomial on the next
        // populated size, adds the evaluations and interpolates the sum. Interpolation is linear and evaluating a polynomial on a larger
        // domain is zero-extension of its coefficients (CirclePoly::extend), so interpolate(values + evaluate(prev)) =
        // interpolate(values) + extend(prev): one inverse transform per size and an addition over the *smaller* size — no forward
        // transform, no full-size accumulate. Exact field arithmetic: the coefficients are the same.
        // Shard group: the 4 coordinate columns of a row-sharded accumulator are gathered whole on their owners (coordinate w on rank
        // w mod count: rows -> columns, one grouped send-receive per size), which interpolate and merge them; ranks without a coordinate idle.
        if (!sharded() || replicate()) {
            // one process (or a group that replicates the transforms: every accumulator is complete on every rank): every size's accumulator is interpolated by the SAME batch of launches (the transforms are independent), then one
            // launch adds the smaller sizes' coefficients onto the largest size's
            std::vector<DCol> all_vals;
            std::vector<u32> logs;
            for (u32 log = max_log; log >= 1; log--) {
                if (!have[log]) continue;
                logs.push_back(log);
                for (int w = 0; w < 4; w++) { DCol v; v.ptr = acc[log].c[w]; v.log_size = log; v.shift = 0; all_vals.push_back(v); }
            }
            if (logs.size() > 13) throw HipError("composition: too many distinct sizes");
            fft_cols(true, all_vals, all_vals);
            AccumulateSizes as{};
            for (int w = 0; w < 4; w++) as.dst[w] = acc[logs[0]].c[w];
            for (size_t k = 1; k < logs.size(); k++) { for (int w = 0; w < 4; w++) as.src[k - 1][w] = acc[logs[k]].c[w]; as.log[k - 1] = logs[k]; }
            as.n = (u32)logs.size() - 1;
            accumulate_sizes(c.stream, as);
            BF_HIP(hipGetLastError());
            trees[3].polys.assign(all_vals.begin(), all_vals.begin() + 4);
            trees[3].owner.assign(4, OWNER_ALL);
            // replicate policy: the composition LDE is (virtually) row-sharded like every full-size column — commit_tree keys that on an owner entry
            if (replicate() && slice_log(all_vals[0].log_size + cfg.log_blowup)) for (int w = 0; w < 4; w++) trees[3].owner[w] = (u32)w % c.shard.count;
            return;
        }
        bool cur_have = false; std::vector<DCol> cur(4);
        bool cur_owned = false;               // `cur` is complete only on the coordinate's owner
        auto owner_of = [&](int w) { return (u32)w % c.shard.count; };
        for (u32 log = 1; log <= max_log; log++) {
            if (!have[log]) continue;
            std::vector<DCol> vals(4), mine_vals;
            const bool sl = acc[log].lc != 0;
            if (sl) {
                const size_t cells = slice_cells(log), bytes = cells * sizeof(u32), first = slice_first(log);
                std::vector<Xfer> sends, recvs;
                for (int w = 0; w < 4; w++) {
                    vals[w].log_size = log; vals[w].shift = 0;
                    sends.push_back({owner_of(w), acc[log].c[w] + first, bytes});
                    if (owner_of(w) == c.shard.rank) {
                        vals[w].ptr = c.alloc_u32(size_t(1) << log);
                        for (u32 r = 0; r < c.shard.count; r++) recvs.push_back({r, vals[w].ptr + r * cells, bytes});
                        mine_vals.push_back(vals[w]);
                    }
                }
                // receive order per peer must follow that peer's send order (coordinate ascending): regroup by coordinate within a peer
                std::stable_sort(recvs.begin(), recvs.end(), [](const Xfer& a, const Xfer& b) { return a.peer < b.peer; });
                c.shard.comm->exchange(c.stream, sends, recvs);
            } else {
                if (cur_owned) throw HipError("composition: a replicated accumulator above a row-sharded one");
                for (int w = 0; w < 4; w++) { vals[w].ptr = acc[log].c[w]; vals[w].log_size = log; vals[w].shift = 0; mine_vals.push_back(vals[w]); }
            }
            fft_cols(true, mine_vals, mine_vals);
            if (cur_have)
                for (int w = 0; w < 4; w++) if (!sl || owner_of(w) == c.shard.rank) accumulate(c.stream, vals[w].ptr, cur[w].ptr, 1u << cur[w].log_size);
            cur = vals; cur_have = true; cur_owned = sl;
        }
        trees[3].polys = cur;
        trees[3].owner.assign(4, OWNER_ALL);
        // The composition LDE is one size above the largest accumulator: it can be row-sharded (quotients, FRI first layer) although no
        // accumulator was. Then every rank holds the complete coefficients and coordinate w's owner alone extends them.
        if (cur_owned || (sharded() && slice_log(cur[0].log_size + cfg.log_blowup))) for (int w = 0; w < 4; w++) trees[3].owner[w] = owner_of(w);
    }

    // PolyOps::eval_at_point for every (column, mask point).
    // sample_prepare: the job list (addresses, sizes, which point) — known before the out-of-domain point is drawn.
    struct SamplePlan { std::vector<EvalJob> jobs; u32 partial_off = 0, n_all = 0; };
    SamplePlan sample_prepare(const std::vector<DTree>& trees, const std::vector<std::vector<std::vector<u32>>>& mask) {
        // Shard group: a sample is evaluated by ONE rank — the owner of the polynomial's coefficients, or for polynomials every rank holds the
        // rank (job index mod count), which splits that work — the others leave a zero and one max-reduce completes the array everywhere.
        SamplePlan sp;
        for (size_t t = 0; t < trees.size(); t++)
            for (size_t col = 0; col < trees[t].polys.size(); col++)
                for (u32 pt : mask[t][col]) {
                    const u32 ji = sp.n_all++;
                    const u32 owner = trees[t].owner.empty() ? OWNER_ALL : trees[t].owner[col];
                    if (sharded() && ((owner == OWNER_ALL || replicate()) ? ji % c.shard.count : owner) != c.shard.rank) continue;
                    const DCol& p = trees[t].polys[col];
                    EvalJob j{}; j.coeffs = p.ptr; j.log_n = p.log_size - p.shift; j.point = pt; j.factor_shift = p.shift; j.partial_off = sp.partial_off; j.out_idx = ji;
                    sp.partial_off += j.log_n > 12 ? 1u << (j.log_n - 12) : 1u;
                    sp.jobs.push_back(j);
                }
        return sp;
    }
    // factor tables: F[0] = y, F[1] = x, F[b] = double_x^(b-1)(x); 32 entries per point
    static void sample_factors(const std::vector<PtQ>& points, uint4* factors) {
        for (size_t p = 0; p < points.size(); p++) {
            Q31 x = points[p].x;
            auto pk = [](Q31 q) { return make_uint4(q.a.a, q.a.b, q.b.a, q.b.b); };
            factors[p * 32 + 0] = pk(points[p].y);
            for (u32 b = 1; b < 32; b++) { factors[p * 32 + b] = pk(x); x = q_double_x(x); }
        }
    }
    // Mailbox mode (one process per proof): the sampling kernels go onto the stream before the point is drawn — jobs staged, factor tables
    // empty —, sample_fill writes the tables into the ring (the caller posts), sample_finish reads the values from their pinned slot.
    struct SampleRun { uint4* h_factors = nullptr; u32 n_all = 0; };
    SampleRun sample_enqueue(const SamplePlan& sp, size_t n_points, Mailbox* mbx) {
        SampleRun sr; sr.n_all = sp.n_all;
        if (sp.n_all * sizeof(uint4) > c.h_small_bytes - 4096) throw HipError("sampling: too many samples for the pinned result buffer");
        std::vector<uint4> factors(n_points * 32, make_uint4(0, 0, 0, 0));
        c.stage_checkpoint();
        mbx->begin();
        const uint4* d_factors = c.stage(factors.data(), factors.size());
        const EvalJob* d_jobs = sp.jobs.empty() ? nullptr : c.stage(sp.jobs.data(), sp.jobs.size());
        mbx->arm();
        sr.h_factors = mbx->host(d_factors);
        void* d_partials = c.arena.alloc(size_t(sp.partial_off ? sp.partial_off : 1) * sizeof(uint4));
        eval_at_points(c.stream, d_jobs, (u32)sp.jobs.size(), sp.partial_off, d_factors, d_partials, c.d_small_alias + 4096);
        BF_HIP(hipGetLastError());
        return sr;
    }
    void sample_fill(const SampleRun& sr, const std::vector<PtQ>& points) { sample_factors(points, sr.h_factors); }
    void sample_finish(const std::vector<DTree>& trees, const std::vector<std::vector<std::vector<u32>>>& mask, StarkProof& pf, const SampleRun& sr) {
        const uint4* out = reinterpret_cast<const uint4*>(c.h_small + 4096);
        pf.sampled_values.resize(trees.size());
        size_t ji = 0;
        for (size_t t = 0; t < trees.size(); t++) {
            pf.sampled_values[t].resize(trees[t].polys.size());
            for (size_t col = 0; col < trees[t].polys.size(); col++)
                for (size_t k = 0; k < mask[t][col].size(); k++, ji++) pf.sampled_values[t][col].push_back(q_make(out[ji].x, out[ji].y, out[ji].z, out[ji].w));
        }
        if (ji != sr.n_all) throw HipError("sampling: job count mismatch");
    }
    void sample(std::vector<DTree>& trees, const std::vector<std::vector<std::vector<u32>>>& mask, const std::vector<PtQ>& points, StarkProof& pf, const SamplePlan& sp) {
        std::vector<uint4> factors(points.size() * 32, make_uint4(0, 0, 0, 0));
        sample_factors(points, factors.data());
        const std::vector<EvalJob>& jobs = sp.jobs;
        const u32 partial_off = sp.partial_off, n_all = sp.n_all;
        c.stage_checkpoint();
        StageBatch sb(c);
        const uint4* d_factors = c.stage(factors.data(), factors.size());     // through the pinned staging ring (no pageable copies)
        const EvalJob* d_jobs = jobs.empty() ? nullptr : c.stage(jobs.data(), jobs.size());
        sb.end();
        void* d_partials = c.arena.alloc(size_t(partial_off ? partial_off : 1) * sizeof(uint4));
        std::vector<uint4> out(n_all);
        if (!sharded() && n_all * sizeof(uint4) <= c.h_small_bytes - 4096) {
            // the second stage writes the samples into the pinned bounce buffer itself
            eval_at_points(c.stream, d_jobs, (u32)jobs.size(), partial_off, d_factors, d_partials, c.d_small_alias + 4096);
            BF_HIP(hipGetLastError());
            c.sync();
            memcpy(out.data(), c.h_small + 4096, out.size() * sizeof(uint4));
        } else {
            uint4* d_out = (uint4*)c.arena.alloc(n_all * sizeof(uint4));
            if (sharded()) BF_HIP(hipMemsetAsync(d_out, 0, n_all * sizeof(uint4), c.stream));
            eval_at_points(c.stream, d_jobs, (u32)jobs.size(), partial_off, d_factors, d_partials, d_out);
            BF_HIP(hipGetLastError());
            if (sharded()) c.shard.comm->all_reduce_max_u32(c.stream, reinterpret_cast<u32*>(d_out), size_t(n_all) * 4);
            c.read_back(out.data(), d_out, out.size() * sizeof(uint4));
        }
        pf.sampled_values.resize(trees.size());
        size_t ji = 0;
        for (size_t t = 0; t < trees.size(); t++) {
            pf.sampled_values[t].resize(trees[t].polys.size());
            for (size_t col = 0; col < trees[t].polys.size(); col++)
                for (size_t k = 0; k < mask[t][col].size(); k++, ji++) pf.sampled_values[t][col].push_back(q_make(out[ji].x, out[ji].y, out[ji].z, out[ji].w));
        }
    }

    // Mailbox mode of compute_quotients (one process per proof). quotients_enqueue: the size groups, their storage, the batch STRUCTURE (which
    // depends on the sample points, known by now, not on the sampled values) and the launches — the largest group behind mailbox mb0, the other
    // groups behind mb1. quotients_fill: the same constants as compute_quotients, written over the staged blocks; mb0 is posted as soon as the
    // largest group's constants are in place, mb1 after the rest (computed while the first launch runs).
    struct QuotientGroup {
        u32 log = 0; std::vector<ColDesc> descs; std::vector<ColSamples> cols; std::vector<std::pair<size_t, size_t>> src;   // (tree, column) per column
        QuotientBatch* h_batches = nullptr; QuotientEntry* h_entries = nullptr; size_t n_batches = 0, n_entries = 0;
    };
    struct QuotientRun { std::vector<QuotientGroup> groups; std::vector<DSecure> out; };
    QuotientRun quotients_enqueue(std::vector<DTree>& trees, const std::vector<std::vector<std::vector<u32>>>& mask, const std::vector<PtQ>& points, Mailbox* mb0, Mailbox* mb1) {
        struct FlatCol { DCol col; size_t tree, idx; };
        std::vector<FlatCol> flat;
        for (size_t t = 0; t < trees.size(); t++) for (size_t i = 0; i < trees[t].evals.size(); i++) flat.push_back({trees[t].evals[i], t, i});
        std::stable_sort(flat.begin(), flat.end(), [](const FlatCol& a, const FlatCol& b) { return a.col.log_size > b.col.log_size; });
        QuotientRun qr;
        for (size_t i = 0; i < flat.size();) {
            size_t j = i; const u32 log = flat[i].col.log_size;
            while (j < flat.size() && flat[j].col.log_size == log) j++;
            QuotientGroup g; g.log = log;
            for (size_t k = i; k < j; k++) {
                g.descs.push_back(flat[k].col.desc());
                const auto& pts = mask[flat[k].tree][flat[k].idx];
                if (pts.size() > 2) throw HipError("quotients: more than two mask points on a column");
                ColSamples cs{};
                for (size_t s = 0; s < pts.size(); s++) { cs.point[cs.n] = pts[s]; cs.value[cs.n] = q_zero(); cs.n++; }
                g.cols.push_back(cs); g.src.push_back({flat[k].tree, flat[k].idx});
            }
            qr.groups.push_back(std::move(g));
            i = j;
        }
        std::vector<QuotientArgs> launches;
        std::vector<QuotientBatch> batches; std::vector<QuotientEntry> entries;
        c.stage_checkpoint();
        auto stage_group = [&](QuotientGroup& g, Mailbox* m) {
            batches.clear(); entries.clear();
            build_quotient_batches_indexed(g.cols.data(), g.cols.size(), points, q_one(), batches, entries);
            quotient_entries_finish(batches.data(), batches.size(), entries.data(), g.descs.data());
            DSecure q; q.log_size = g.log; q.lc = 0;
            for (int w = 0; w < 4; w++) q.c[w] = c.alloc_u32(size_t(1) << g.log);
            QuotientArgs a{};
            a.batches = batches.empty() ? nullptr : c.stage(batches.data(), batches.size());
            a.entries = entries.empty() ? nullptr : c.stage(entries.data(), entries.size());
            g.n_batches = batches.size(); g.n_entries = entries.size();
            g.h_batches = a.batches ? m->host(a.batches) : nullptr; g.h_entries = a.entries ? m->host(a.entries) : nullptr;
            a.n_batches = (u32)batches.size(); a.log = g.log; a.tw = c.d_tw; a.tw_total = 1u << c.tw_root_log;
            for (int w = 0; w < 4; w++) a.out[w] = q.c[w];
            launches.push_back(a);
            qr.out.push_back(q);
        };
        // the largest group behind its own mailbox
        mb0->begin();
        stage_group(qr.groups[0], mb0);
        QuotientArgs first = launches[0];
        const u32 nblocks0 = quotient_groups_layout(&first, 1);
        const QuotientArgs* d_first = c.stage(&first, 1);
        mb0->arm();
        BF_HIP(hipEventRecord(c.ev[4], c.stream));       // the quotient phase's GPU time starts behind the mailbox, not in front of it
        accumulate_quotients(c.stream, d_first, 1, nblocks0);
        if (qr.groups.size() > 1) {
            mb1->begin();
            for (size_t g = 1; g < qr.groups.size(); g++) stage_group(qr.groups[g], mb1);
            const u32 nblocks = quotient_groups_layout(launches.data() + 1, (u32)launches.size() - 1);
            const QuotientArgs* d_groups = c.stage(launches.data() + 1, launches.size() - 1);
            mb1->arm();
            accumulate_quotients(c.stream, d_groups, (u32)launches.size() - 1, nblocks);
        }
        BF_HIP(hipGetLastError());
        return qr;
    }
    void quotients_fill(QuotientRun& qr, const std::vector<std::vector<std::vector<u32>>>& mask, const std::vector<PtQ>& points, const StarkProof& pf, Q31 random_coeff,
                        Mailbox* mb0, Mailbox* mb1) {
        std::vector<QuotientBatch> batches; std::vector<QuotientEntry> entries;
        for (size_t gi = 0; gi < qr.groups.size(); gi++) {
            QuotientGroup& g = qr.groups[gi];
            for (size_t k = 0; k < g.cols.size(); k++)
                for (u32 s = 0; s < g.cols[k].n; s++) g.cols[k].value[s] = pf.sampled_values[g.src[k].first][g.src[k].second][s];
            batches.clear(); entries.clear();
            build_quotient_batches_indexed(g.cols.data(), g.cols.size(), points, random_coeff, batches, entries);
            quotient_entries_finish(batches.data(), batches.size(), entries.data(), g.descs.data());
            if (batches.size() != g.n_batches || entries.size() != g.n_entries) throw HipError("quotients: the batch structure changed between enqueue and fill");
            if (g.n_batches) memcpy(g.h_batches, batches.data(), batches.size() * sizeof(QuotientBatch));
            if (g.n_entries) memcpy(g.h_entries, entries.data(), entries.size() * sizeof(QuotientEntry));
            if (gi == 0) mb0->post();
        }
        mb1->post();
    }

    // compute_fri_quotients: one secure column per distinct LDE size, descending.
    std::vector<DSecure> compute_quotients(std::vector<DTree>& trees, const std::vector<std::vector<std::vector<u32>>>& mask, const std::vector<PtQ>& points,
                                           const StarkProof& pf, Q31 random_coeff, std::vector<LevelWait>* q_waits = nullptr) {
        struct FlatCol { DCol col; size_t tree, idx; };
        std::vector<FlatCol> flat;
        for (size_t t = 0; t < trees.size(); t++) for (size_t i = 0; i < trees[t].evals.size(); i++) flat.push_back({trees[t].evals[i], t, i});
        std::stable_sort(flat.begin(), flat.end(), [](const FlatCol& a, const FlatCol& b) { return a.col.log_size > b.col.log_size; });
        mark("quotient columns sorted");
        std::vector<DSecure> out;
        std::vector<QuotientArgs> launches;
        // Launches: one per size group of >= 2^19 rows, largest first, each followed by an event (q_waits) — the FRI first-layer tree hashes
        // level L as soon as the quotient of size L exists, on the partner stream, while the smaller groups are still being computed — and one
        // launch for all the smaller groups together. (Shard group / host channel: one launch, no events.)
        const bool pipelined = (c.overlap & 2u) && q_waits && !sharded() && c.conv.merkle_channel == 0;
        // Otherwise the LARGEST group is launched as soon as its own constants exist (four composition columns: a handful of products) and the
        // host prepares the constants of the other groups — ~40 us of QM31 arithmetic, with the GPU idle behind the sampled values' round trip —
        // while that launch runs; the rest follows as the second launch.
        const bool early_first = !pipelined && !sharded();
        u32 launched = 0;
        c.stage_checkpoint();
        auto sb = std::make_unique<StageBatch>(c);
        std::vector<ColDesc> descs; std::vector<ColSamples> col_samples;        // reused by every size group
        std::vector<QuotientBatch> batches; std::vector<QuotientEntry> entries;
        descs.reserve(flat.size()); col_samples.reserve(flat.size()); entries.reserve(2 * flat.size()); batches.reserve(32);
        for (size_t i = 0; i < flat.size();) {
            size_t j = i; u32 log = flat[i].col.log_size;
            while (j < flat.size() && flat[j].col.log_size == log) j++;
            // ColumnSampleBatch::new_vec + quotient_constants (host/quotients.h)
            descs.clear(); col_samples.clear();
            for (size_t k = i; k < j; k++) {
                descs.push_back(flat[k].col.desc());
                const auto& pts = mask[flat[k].tree][flat[k].idx];
                ColSamples cs{};
                if (pts.size() > 2) throw HipError("quotients: more than two mask points on a column");
                for (size_t s = 0; s < pts.size(); s++) { cs.point[cs.n] = pts[s]; cs.value[cs.n] = pf.sampled_values[flat[k].tree][flat[k].idx][s]; cs.n++; }
                col_samples.push_back(cs);
            }
            batches.clear(); entries.clear();
            build_quotient_batches_indexed(col_samples.data(), col_samples.size(), points, random_coeff, batches, entries);
            quotient_entries_finish(batches.data(), batches.size(), entries.data(), descs.data());
            // shard group: the quotient of a row-sharded size is computed for this rank's row range only (every column of the group is
            // either complete or row-sharded over the same range)
            const bool sl = slice_log(log);
            DSecure q; q.log_size = log; q.lc = sl ? lc() : 0;
            for (int w = 0; w < 4; w++) q.c[w] = sl ? alloc_slice(log) : c.alloc_u32(size_t(1) << log);
            QuotientArgs a{};
            if (sl) { a.row0 = (u32)slice_first(log); a.n_rows = (u32)slice_cells(log); }
            // a full-size column of the group is row-sharded exactly when the group is; a replicated one may also be complete on every rank
            for (size_t k = i; k < j; k++) if (flat[k].col.sliced() != sl && (flat[k].col.shift == 0 || flat[k].col.sliced())) throw HipError("quotients: inconsistent row-sharding in a size group");
            a.batches = batches.empty() ? nullptr : c.stage(batches.data(), batches.size());
            a.entries = entries.empty() ? nullptr : c.stage(entries.data(), entries.size());
            a.n_batches = (u32)batches.size(); a.log = log; a.tw = c.d_tw; a.tw_total = 1u << c.tw_root_log;
            for (int w = 0; w < 4; w++) a.out[w] = q.c[w];
            launches.push_back(a);
            out.push_back(q);
            i = j;
            if (early_first && launches.size() == 1 && i < flat.size()) {
                QuotientArgs first = launches[0];
                const u32 nblocks = quotient_groups_layout(&first, 1);
                const QuotientArgs* d_first = c.stage(&first, 1);
                sb->end();
                accumulate_quotients(c.stream, d_first, 1, nblocks);
                mark("largest quotient group launched");
                sb = std::make_unique<StageBatch>(c);
                launched = 1;
            }
        }
        std::vector<std::pair<u32, u32>> ranges;      // [first group, count)
        {
            u32 g = launched;
            if (pipelined) while (g < launches.size() && launches[g].log >= 19) { ranges.push_back({g, 1u}); g++; }
            if (g < launches.size()) ranges.push_back({g, (u32)launches.size() - g});
        }
        std::vector<u32> blocks;
        for (auto& r : ranges) blocks.push_back(quotient_groups_layout(launches.data() + r.first, r.second));
        const QuotientArgs* d_groups = launches.empty() ? nullptr : c.stage(launches.data(), launches.size());
        sb->end();                                  // one copy for the parameter blocks of every (remaining) size group
        for (size_t k = 0; k < ranges.size(); k++) {
            accumulate_quotients(c.stream, d_groups + ranges[k].first, ranges[k].second, blocks[k]);
            if (pipelined) { hipEvent_t e = c.next_event(); BF_HIP(hipEventRecord(e, c.stream)); q_waits->push_back({(int)launches[ranges[k].first].log, e}); }
        }
        BF_HIP(hipGetLastError());
        return out;
    }

    static std::vector<size_t> fold_queries(const std::vector<size_t>& q, u32 n) {
        std::vector<size_t> o;
        for (size_t x : q) { size_t y = x >> n; if (o.empty() || o.back() != y) o.push_back(y); }
        return o;
    }
    // compute_decommitment_positions_and_witness_evals with fold_step = 1; witness values are gathered later.
    static void positions_and_witness(const std::vector<size_t>& queries, std::vector<size_t>& positions, std::vector<size_t>& witness_pos) {
        size_t i = 0;
        while (i < queries.size()) {
            size_t j = i;
            while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
            size_t start = (queries[i] >> 1) << 1, qi = i;
            for (size_t pos = start; pos < start + 2; pos++) {
                positions.push_back(pos);
                if (qi < j && queries[qi] == pos) { qi++; continue; }
                witness_pos.push_back(pos);
            }
            i = j;
        }
    }
    Finisher gather_secure_deferred(Gather& g, const DSecure& s, const std::vector<size_t>& pos, std::vector<Q31>* out) {
        size_t first = g.n_words;
        for (size_t p : pos) for (int w = 0; w < 4; w++) g.add(s.c[w], p, s.mine(p, c.shard.rank));
        size_t n = pos.size();
        return [first, n, out](const std::vector<u32>& d) { for (size_t k = 0; k < n; k++) out->push_back(q_make(d[first + 4 * k], d[first + 4 * k + 1], d[first + 4 * k + 2], d[first + 4 * k + 3])); };
    }
    std::vector<Q31> gather_secure(const DSecure& s, const std::vector<size_t>& pos) {
        Gather g;
        for (size_t p : pos) for (int w = 0; w < 4; w++) g.add(s.c[w], p, s.mine(p, c.shard.rank));
        auto d = g.run(c);
        std::vector<Q31> out;
        for (size_t k = 0; k < pos.size(); k++) out.push_back(q_make(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]));
        return out;
    }
    static std::vector<DCol> secure_cols(const DSecure& s) {
        std::vector<DCol> v(4);
        for (int w = 0; w < 4; w++) { v[w].ptr = s.c[w]; v[w].log_size = s.log_size; v[w].shift = 0; v[w].lc = s.lc; }
        return v;
    }

    void fri_and_decommit(std::vector<DTree>& trees, std::vector<DSecure>& quotients, StarkProof& pf, const std::vector<LevelWait>& q_waits,
                          const std::function<void()>& while_the_commit_phase_runs) {
        // FriProver::commit — first layer: one Merkle tree over the coordinate columns of every quotient.
        // The channel is stepped on the device through the whole commit phase (k_channel_mix_root_draw): per layer mix_root(root) and
        // draw_felt() run as a one-lane kernel and the folds read alpha from device memory, so the ~25 layers are enqueued back to back
        // with no host round trip. The roots arrive in pinned memory; the host channel replays the same steps afterwards and must end
        // in the same state.
        std::vector<DCol> first_cols;
        for (auto& q : quotients) for (auto& col : secure_cols(q)) first_cols.push_back(col);
        const size_t max_layers = 40;
        Hash32* pinned_roots = reinterpret_cast<Hash32*>(c.h_small + 64);                       // [0] first layer, [1 + i] inner layer i
        u32* pinned_chan = reinterpret_cast<u32*>(c.h_small + 64 + 32 * (max_layers + 1));      // digest[8] || n_sent
        u32* d_chan = c.alloc_u32(16);
        u32* d_alpha = c.alloc_u32(8 * (max_layers + 1));
        u32* d_roots = c.alloc_u32(8 * (max_layers + 1));                                        // root copies, read back once
        memcpy(pinned_chan, ch.digest.b, 32); pinned_chan[8] = ch.n_sent;
        // Pipelined with the quotient launches (q_waits): the channel state, the tree layouts and the first-layer tree go to the PARTNER stream —
        // on the main stream they would queue up behind every quotient kernel.
        hipStream_t main_stream = c.stream;
        const bool first_on_aux = c.conv.merkle_channel == 0 && !q_waits.empty();
        if (first_on_aux) c.stream = c.aux_of(main_stream);
        struct StreamRestore { Ctx& c; hipStream_t s; ~StreamRestore() { c.stream = s; } } restore_stream{c, main_stream};
        BF_HIP(hipMemcpyAsync(d_chan, pinned_chan, 36, hipMemcpyHostToDevice, c.stream));
        // Poseidon252Channel is stepped on the host (one root read-back per layer): two serial Hades permutations by a single lane would
        // cost more than the round trip. commit_step = Merkle tree of a layer + mix_root + draw alpha (alpha || alpha^2 -> d_alpha[idx]).
        const bool host_channel = c.conv.merkle_channel == 1;
        u32 line_log = quotients[0].log_size - 1;
        const u32 last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup;
        if (line_log > last_log + max_layers) throw HipError("FRI: too many layers");
        // Shard group: a layer with >= 2^14 rows per rank is row-sharded like the quotients (a fold maps the sibling pair (2i, 2i+1) to cell
        // i, so a rank's row range of the source folds into its row range of the destination). The first layer below that size is produced
        // range-wise into a complete buffer and finished by one all-gather; everything smaller is folded redundantly on every rank.
        auto new_layer = [&](u32 log) {
            DSecure l; l.log_size = log; l.lc = slice_log(log) ? lc() : 0;
            for (int w = 0; w < 4; w++) l.c[w] = l.lc ? alloc_slice(log) : c.alloc_u32(size_t(1) << log);
            return l;
        };
        // Every layer's storage and (device channel) every tree's layout exist before the first launch: the column descriptors and level
        // tables of all ~26 trees reach the device in ONE staging copy instead of one in front of every layer of the serial chain.
        const u32 n_inner = line_log > last_log ? line_log - last_log : 0;
        std::vector<DSecure> layers(n_inner + 1);
        for (u32 i = 0; i <= n_inner; i++) layers[i] = new_layer(line_log - i);
        std::vector<MerklePlan> plans;                  // [0] first layer, [1 + i] inner layer i
        if (!host_channel) {
            c.stage_checkpoint();
            StageBatch sb(c);
            plans.reserve(n_inner + 1);
            plans.push_back(merkle_plan(first_cols));
            for (u32 i = 0; i < n_inner; i++) plans.push_back(merkle_plan(secure_cols(layers[i])));
            sb.end();
        }
        auto commit_step = [&](size_t plan_idx, const std::vector<DCol>& cols, u32 alpha_idx, u32 root_idx) -> DevMerkle {
            if (!host_channel) { ChannelStep st{d_chan, d_alpha + 8 * alpha_idx, d_roots + 8 * root_idx}; return merkle_run(plans[plan_idx], nullptr, /*no_readback=*/true, &st); }
            DevMerkle t = merkle_commit(cols);
            ch.mix_root(t.root);
            const Q31 a = ch.draw_felt(), sq = q_mul(a, a);
            const u32 w[8] = {a.a.a, a.a.b, a.b.a, a.b.b, sq.a.a, sq.a.b, sq.b.a, sq.b.b};
            c.stage_checkpoint();
            const u32* st = c.stage(w, 8);
            BF_HIP(hipMemcpyAsync(d_alpha + 8 * alpha_idx, st, 32, hipMemcpyDeviceToDevice, c.stream));
            return t;
        };
        DevMerkle first_tree;
        if (first_on_aux) {
            // level L of the first-layer tree is hashed as soon as the quotient of size L exists; joined before the first fold
            ChannelStep st{d_chan, d_alpha, d_roots};
            first_tree = merkle_run(plans[0], nullptr, /*no_readback=*/true, &st, &q_waits);
            hipEvent_t e3 = c.next_event();
            BF_HIP(hipEventRecord(e3, c.stream));
            c.stream = main_stream;
            BF_HIP(hipStreamWaitEvent(main_stream, e3, 0));
        } else first_tree = commit_step(0, first_cols, 0, 0);
        struct Inner { DSecure ev; DevMerkle tree; };
        std::vector<Inner> inner;
        // destination range of a fold whose SOURCE has 2^src_log rows: the image of this rank's source range when the source is sharded
        auto fold_range = [&](u32 src_log, bool src_sliced, u32& first, u32& count) {
            if (src_sliced) { first = (u32)(slice_first(src_log) >> 1); count = (u32)(slice_cells(src_log) >> 1); } else { first = 0; count = 0; }
        };
        // a complete (unsharded) buffer of which every rank has filled only its range is finished by one all-gather per coordinate
        auto complete = [&](const DSecure& l) { for (int w = 0; w < 4; w++) c.shard.comm->all_gather(c.stream, l.c[w], sizeof(u32) << (l.log_size - lc())); };
        // circle -> line: the largest quotient opens layer 0 (nothing folded into it yet: no zero fill)
        size_t qi = 0;
        if (quotients[0].log_size - 1 != line_log) throw HipError("FRI: first layer size");
        {
            const DSecure& q = quotients[qi++];
            const u32* src[4] = {q.c[0], q.c[1], q.c[2], q.c[3]};
            u32 first, count; fold_range(q.log_size, q.lc != 0, first, count);
            fold_circle_into_line(c.stream, layers[0].c, src, d_alpha, c.d_itw, c.tw_root_log, q.log_size, /*fresh=*/true, first, count);
            if (q.lc != 0 && layers[0].lc == 0) complete(layers[0]);
        }
        // inner layers: commit layer k (-> alpha_{k+1}), then ONE launch folds it into layer k + 1 together with the quotient of layer k's
        // size (fold_line, then dst * alpha^2 + fold_circle: both with alpha_{k+1}). Below 2^10 rows the rest of the phase is one launch.
        const u32 TAIL_LOG = 10;
        // layers of 2^11 .. 2^16 rows: fold + tree + channel step in ONE launch (merkle.hip: k_fri_layer) — device channel, one process
        // (in a shard group: layers every rank holds whole, folded from a layer every rank holds whole)
        auto fused = [&](u32 k) { const u32 lg = line_log - k; return !host_channel && k >= 1 && k < n_inner && lg >= 11 && lg <= 16 && layers[k].lc == 0 && layers[k - 1].lc == 0; };
        auto take_quotient = [&](u32 size) -> const DSecure* {
            const DSecure* q = (qi < quotients.size() && quotients[qi].log_size == size) ? &quotients[qi++] : nullptr;
            if (qi < quotients.size() && quotients[qi].log_size == size) throw HipError("FRI: two quotient columns of one size");
            return q;
        };
        u32* d_counter = nullptr;
        u32 li = 0;
        for (; li < n_inner; li++) {
            const u32 log = line_log - li;
            if (!host_channel && log <= TAIL_LOG) break;
            Inner in; in.ev = layers[li];
            if (fused(li)) {
                d_counter = c.merkle_counter();
                const DSecure* q = take_quotient(log + 1);        // the circle evaluation that folds into this layer
                const DevMerkle& mk = plans[1 + li].mk;
                if (mk.max_log != log) throw HipError("FRI layer: tree layout");
                FriLayerArgs fa{};
                for (int w = 0; w < 4; w++) { fa.src[w] = layers[li - 1].c[w]; fa.quot[w] = q ? q->c[w] : nullptr; fa.dst[w] = layers[li].c[w]; }
                for (u32 lg = 0; lg <= log; lg++) { if (mk.shifts[lg] != 0) throw HipError("FRI layer: replicated level"); fa.tree[lg] = (uint4*)mk.layers[lg]; }
                fa.alpha8 = d_alpha + 8 * li; fa.itw = c.d_itw; fa.tw_total = 1u << c.tw_root_log; fa.log = log; fa.rfc = c.conv.merkle_node_hash ? 0xFFFFFFFFu : 0u;
                fa.counter = d_counter; fa.chan = d_chan; fa.alpha_out = d_alpha + 8 * (li + 1); fa.root_out = d_roots + 8 * (1 + li);
                fri_layer(c.stream, fa);
                in.tree = mk;
            } else in.tree = commit_step(1 + li, secure_cols(layers[li]), li + 1, 1 + li);
            inner.push_back(in);
            if (fused(li + 1)) continue;                           // the next layer folds this one itself
            const DSecure* q = take_quotient(log);
            DSecure& next = layers[li + 1];
            const u32* src[4] = {layers[li].c[0], layers[li].c[1], layers[li].c[2], layers[li].c[3]};
            const u32* qs[4] = {q ? q->c[0] : nullptr, q ? q->c[1] : nullptr, q ? q->c[2] : nullptr, q ? q->c[3] : nullptr};
            if (q && (q->lc != 0) != (layers[li].lc != 0)) throw HipError("FRI: a layer and the quotient of its size are sharded differently");
            u32 first, count; fold_range(log, layers[li].lc != 0, first, count);
            fold_line_circle(c.stream, next.c, src, q ? qs : nullptr, d_alpha + 8 * (li + 1), c.d_itw, c.tw_root_log, log, first, count);
            if (layers[li].lc != 0 && next.lc == 0) complete(next);
        }
        if (li < n_inner) {
            // k_fri_tail: layers li .. n_inner - 1 (2^TAIL_LOG rows and below): trees, channel steps and folds by one workgroup
            FriTailArgs ta{};
            ta.n_layers = n_inner - li; ta.top_log = line_log - li; ta.alpha_idx = li + 1; ta.root_idx = 1 + li;
            ta.chan = d_chan; ta.alpha = d_alpha; ta.roots = d_roots; ta.itw = c.d_itw; ta.tw_total = 1u << c.tw_root_log; ta.rfc = c.conv.merkle_node_hash ? 0xFFFFFFFFu : 0u;
            if (ta.n_layers > 10 || ta.top_log > TAIL_LOG) throw HipError("FRI tail: too many layers");
            for (u32 k = 0; k < ta.n_layers; k++) {
                const u32 log = ta.top_log - k;
                FriTailLayer& L = ta.layer[k];
                for (int w = 0; w < 4; w++) L.ev[w] = layers[li + k].c[w];
                if (qi < quotients.size() && quotients[qi].log_size == log) { for (int w = 0; w < 4; w++) L.quot[w] = quotients[qi].c[w]; qi++; }
                const DevMerkle& mk = plans[1 + li + k].mk;
                if (mk.max_log != log) throw HipError("FRI tail: tree layout");
                for (u32 lg = 0; lg <= log; lg++) { if (mk.shifts[lg] != 0) throw HipError("FRI tail: replicated level"); L.tree[lg] = (uint4*)mk.layers[lg]; }
                Inner in; in.ev = layers[li + k]; in.tree = mk;
                inner.push_back(in);
            }
            for (int w = 0; w < 4; w++) ta.ev_last[w] = layers[n_inner].c[w];
            double tail_nodes = 0;
            for (u32 k = 0; k < ta.n_layers; k++) tail_nodes += (double)((2u << (ta.top_log - k)) - 1);
            c.stage_checkpoint();
            fri_tail(c.stream, c.stage(&ta, 1), 48.0 * tail_nodes, tail_nodes);
        }
        DSecure layer = layers[n_inner];
        line_log = last_log;
        if (qi != quotients.size()) throw HipError("FRI: not all columns consumed");
        BF_HIP(hipGetLastError());
        if (!host_channel) {
            BF_HIP(hipMemcpyAsync(pinned_chan, d_chan, 36, hipMemcpyDeviceToHost, c.stream));
            BF_HIP(hipMemcpyAsync(pinned_roots, d_roots, 32 * (1 + inner.size()), hipMemcpyDeviceToHost, c.stream));
        }
        // last layer: 2^last_log evaluations -> line polynomial (host; LineEvaluation::interpolate on <= 2 values for the default config)
        {
            if (last_log != 1 || cfg.log_last_layer_degree_bound != 0) throw HipError("only the default FRI last-layer configuration is supported");
            std::vector<size_t> pos = {0, 1};
            mark("FRI commit phase enqueued");
            Gather gl;
            for (size_t p : pos) for (int w = 0; w < 4; w++) gl.add(layer.c[w], p, layer.mine(p, c.shard.rank));
            // the commit phase (~100 launches) is in flight: the host does its own checks now; a failed check waits for the stream before it
            // unwinds (the arena must not be handed out again under running kernels)
            try { while_the_commit_phase_runs(); } catch (...) { (void)hipStreamSynchronize(c.stream); throw; }
            mark("sanity check done");
            static const int last_layer_slot = [] { const char* v = getenv("BFHIP_MB_TAIL"); return v && v[0] == '2' ? -1 : 6; }();
            auto dl = gl.run(c, last_layer_slot);
            std::vector<Q31> v;
            for (size_t k = 0; k < pos.size(); k++) v.push_back(q_make(dl[4 * k], dl[4 * k + 1], dl[4 * k + 2], dl[4 * k + 3]));          // synchronises: roots and the device channel state are on the host now
            mark("FRI last layer arrived");
            if (!host_channel) {
                first_tree.root = pinned_roots[0];
                ch.mix_root(first_tree.root); (void)ch.draw_felt();
                for (size_t li = 0; li < inner.size(); li++) { inner[li].tree.root = pinned_roots[1 + li]; ch.mix_root(inner[li].tree.root); (void)ch.draw_felt(); }
                if (memcmp(pinned_chan, ch.digest.b, 32) != 0 || pinned_chan[8] != ch.n_sent) throw HipError("FRI: device channel diverged from the host channel");
            }
            // line_ifft on 2 values over LineDomain(half_odds(1)): c0 = (v0 + v1) / 2, c1 = (v0 - v1) / (2 x0) must vanish
            u32 inv2 = m_inv(2);
            Q31 c0 = q_mulm(q_add(v[0], v[1]), inv2);
            if (!q_eq(v[0], v[1])) throw HipError("invalid degree");
            pf.fri_proof.last_layer_coeffs = {c0};
            pf.fri_proof.last_layer_log_size = 0;
            ch.mix_felts(&c0, 1);
        }
        tap("fri_commit");

        // proof of work (GrindOps): GPU search in spans, smallest nonce wins. Poseidon252Channel: one Hades permutation per nonce and a
        // 1-in-16 hit rate at pow_bits = 5 (the test reads the top byte of the big-endian digest) — searched on the host.
        if (host_channel) {
            u64 nonce = 0;
            for (;; nonce++) { Channel t = ch; t.mix_u64(nonce); if (t.trailing_zeros() >= cfg.pow_bits) break; if (nonce > (u64(1) << 32)) throw HipError("grind: no nonce found"); }
            pf.proof_of_work = nonce;
            ch.mix_u64(nonce);
        } else if (cfg.pow_bits <= 10 && [&]() {
            // A few bits of work are found faster by the host than by a launch and a read-back (~35 us): 2^pow_bits tries of one compression
            // each on average. GrindOps asks for the SMALLEST nonce: a linear scan from zero finds it. Larger work goes to the GPU search.
            for (u64 nonce = 0; nonce < (u64(64) << cfg.pow_bits); nonce++) {
                Channel t = ch; t.mix_u64(nonce);
                if (t.trailing_zeros() >= cfg.pow_bits) { pf.proof_of_work = nonce; ch.mix_u64(nonce); return true; }
            }
            return false; }()) {
        } else {
            c.stage_checkpoint();
            u32* d_digest = (u32*)c.stage(ch.digest.b, 32);
            unsigned long long init = ~0ull;
            unsigned long long* d_best = (unsigned long long*)c.stage(&init, 1);
            unsigned long long best = ~0ull;
            const u32 span = 1u << 16;
            for (u64 base = 0; best == ~0ull; base += span) {
                grind_span(c.stream, d_digest, base, span, cfg.pow_bits, d_best, c.conv.mix_u64);
                c.read_back(&best, d_best, 8);
            }
            pf.proof_of_work = best;
            ch.mix_u64(best);
        }
        mark("nonce found");

        // FRI decommit
        double t0 = now();
        u32 max_log = quotients[0].log_size;
        std::vector<size_t> queries;
        {
            std::set<size_t> qs; u32 cnt = 0; u32 maskq = (u32)((u64(1) << max_log) - 1);
            while (cnt < cfg.n_queries) {   // Queries::generate: chunks_exact(4) of the drawn bytes (32 per draw for Blake2s, 31 for Poseidon252)
                std::vector<u8> r = ch.draw_random_bytes();
                for (size_t k = 0; 4 * k + 4 <= r.size() && cnt < cfg.n_queries; k++) { u32 w; memcpy(&w, r.data() + 4 * k, 4); qs.insert(w & maskq); cnt++; }
            }
            queries.assign(qs.begin(), qs.end());
        }
        std::map<u32, std::vector<size_t>> positions_by_log;
        for (auto& q : quotients) positions_by_log[q.log_size] = fold_queries(queries, max_log - q.log_size);
        // All decommitment reads (FRI witnesses, Merkle witnesses, queried values) are planned first and fetched by ONE gather launch:
        // the control flow depends only on the query positions.
        Gather g;
        g.reqs.reserve(4096);
        std::vector<Finisher> fin;
        fin.reserve(64);
        {
            std::map<u32, std::vector<size_t>> dpos;
            for (auto& q : quotients) {
                std::vector<size_t> pos, wpos;
                positions_and_witness(fold_queries(queries, max_log - q.log_size), pos, wpos);
                dpos[q.log_size] = pos;
                fin.push_back(gather_secure_deferred(g, q, wpos, &pf.fri_proof.first_layer.fri_witness));
            }
            fin.push_back(decommit(g, first_tree, first_cols, dpos, nullptr, &pf.fri_proof.first_layer.decommitment));
            pf.fri_proof.first_layer.commitment = first_tree.root;
        }
        auto lq = fold_queries(queries, 1);
        pf.fri_proof.inner_layers.resize(inner.size());
        for (size_t li = 0; li < inner.size(); li++) {
            auto& in = inner[li];
            FriLayerProof& lp = pf.fri_proof.inner_layers[li];
            std::vector<size_t> pos, wpos;
            positions_and_witness(lq, pos, wpos);
            fin.push_back(gather_secure_deferred(g, in.ev, wpos, &lp.fri_witness));
            std::map<u32, std::vector<size_t>> dpos; dpos[in.ev.log_size] = pos;
            fin.push_back(decommit(g, in.tree, secure_cols(in.ev), dpos, nullptr, &lp.decommitment));
            lp.commitment = in.tree.root;
            lq = fold_queries(lq, 1);
        }
        pf.queried_values.resize(trees.size());
        pf.decommitments.resize(trees.size());
        for (size_t ti = 0; ti < trees.size(); ti++) {
            fin.push_back(decommit(g, trees[ti].mk, trees[ti].evals, positions_by_log, &pf.queried_values[ti], &pf.decommitments[ti]));
            pf.commitments.push_back(trees[ti].mk.root);
        }
        mark("decommitment planned");
        // The proof's LAST wait is an event wait on purpose: with stamps the host never asks the runtime about the stream, and the runtime keeps
        // the bookkeeping of every launch until somebody does — one real synchronisation per proof, at the point where the stream is about to
        // drain anyway, releases it (without it the next proofs' launches slow down: fib19 +0.3 ms in the mean, r04)
        static const int tail_slot = [] { const char* v = getenv("BFHIP_MB_TAIL"); return v && v[0] == '1' ? 7 : -1; }();
        std::vector<u32> data = g.run(c, tail_slot);
        mark("decommitment data arrived");
        for (auto& f : fin) f(data);
        tm.decommit = now() - t0;
    }
};

// ---- what pool.hip needs of the prover (declared in ctx.h) ----------------------------------------------------------------------------
SharedPreprocessed* shared_preprocessed_create(Ctx& builder) {
    auto* sp = new SharedPreprocessed();
    try { builder.bind(); BF_HIP(hipEventCreateWithFlags(&sp->ready, hipEventDisableTiming)); } catch (...) { delete sp; throw; }
    return sp;
}
void shared_preprocessed_destroy(SharedPreprocessed* sp) { if (sp) { if (sp->ready) (void)hipEventDestroy(sp->ready); delete sp; } }
bool shared_preprocessed_matches(const SharedPreprocessed* sp, const Ctx& c, u32 log_max_rows) { return sp && sp->matches(c, log_max_rows); }
void shared_preprocessed_build(SharedPreprocessed* sp, Ctx& builder, u32 log_max_rows) {
    builder.bind();
    HipProver pv(builder, log_max_rows);
    try { pv.build_shared_preprocessed(*sp); } catch (...) { sp->valid = false; (void)hipStreamSynchronize(builder.stream); throw; }
}
void shared_preprocessed_invalidate(SharedPreprocessed* sp) { if (sp) sp->valid = false; }

}  // namespace bf

using namespace bf;

struct bfhip_trace { TraceInput in; };
// Selects where bfhip_trace_create / bfhip_prove_brainfuck of THIS context build the 13 component tables: 1 = gfx950 kernels (default), 0 = host builders.
extern "C" int32_t bfhip_ctx_set_table_builder(bfhip_ctx* ctx, int32_t on_gpu) { if (!ctx) { bfhip_set_error("null context"); return -1; } ctx->c.tables_on_gpu = on_gpu != 0; return 0; }
// Downloads one row-granular column of a resident trace (tests: GPU tables == host tables).
extern "C" int32_t bfhip_trace_column(bfhip_ctx* ctx, const bfhip_trace* t, uint32_t component, uint32_t column, uint32_t* out_h, size_t cap, size_t* n_rows) {
    try {
        if (!ctx || !t || !n_rows) { bfhip_set_error("null argument"); return -1; }
        if (component >= N_COMPONENTS || column >= t->in.rows[component].size()) { bfhip_set_error("bad component/column"); return -1; }
        const DCol& col = t->in.rows[component][column];
        *n_rows = col.stored();
        if (out_h) {
            if (cap < col.stored()) { bfhip_set_error("capacity"); return -2; }
            ctx->c.bind();
            BF_HIP(hipMemcpyAsync(out_h, col.ptr, col.stored() * sizeof(u32), hipMemcpyDeviceToHost, ctx->c.stream));
            ctx->c.sync();
        }
        return 0;
    } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}

// A proof that fails on one rank of a shard group must not leave the others waiting for its next collective: the transport is told to give
// up (in-process: the rendezvous object is marked failed and every waiting rank throws; RCCL: ncclCommAbort). The group is unusable afterwards.
static void release_group_after_failure(bfhip_ctx* ctx) {
    if (ctx && ctx->c.shard.count > 1 && ctx->c.shard.comm) {
        try { ctx->c.shard.comm->abort(); } catch (...) {}
        // every stream of the context may still carry work of the failed proof (the main stream's handle, the side stream, both partners)
        Ctx& c = ctx->c;
        for (hipStream_t st : {c.stream, c.id_main, c.stream2, c.aux[0], c.aux[1]}) if (st) (void)hipStreamSynchronize(st);
    }
}

static void fill_outputs(HipProver& pv, const BrainfuckProof& bp, char** proof_json, size_t* proof_len, char** transcript, double* phase_seconds) {
    if (proof_json) {
        std::string js = proof_to_json(bp, pv.c.conv.merkle_channel == 1);
        *proof_json = (char*)malloc(js.size() + 1);
        memcpy(*proof_json, js.c_str(), js.size() + 1);
        if (proof_len) *proof_len = js.size();
    }
    if (transcript) { *transcript = (char*)malloc(pv.transcript.size() + 1); memcpy(*transcript, pv.transcript.c_str(), pv.transcript.size() + 1); }
    if (phase_seconds) {
        const PhaseTimes& t = pv.tm;
        double v[10] = {t.preprocessed, t.tables, t.main_trace, t.interaction, t.composition, t.oods, t.quotients, t.fri, t.decommit, t.total};
        memcpy(phase_seconds, v, sizeof v);
    }
}

static int32_t trace_create_common(bfhip_ctx* ctx, const std::vector<Registers>& vm_trace, const std::vector<u32>& ins, bfhip_trace** out,
                                   uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells) {
    if (vm_trace.empty()) throw HipError("EmptyTrace");   // TraceError::EmptyTrace (memory/table.rs:83-86)
    auto* t = new bfhip_trace();
    try { HipProver::upload_trace(ctx->c, vm_trace, ins, t->in, /*use_arena=*/false, /*on_gpu=*/ctx->c.tables_on_gpu); } catch (...) { t->in.release(); delete t; throw; }
    if (log_sizes) memcpy(log_sizes, t->in.log_sizes, sizeof(u32) * N_COMPONENTS);
    if (n_steps) *n_steps = t->in.n_steps;
    if (main_cells) *main_cells = t->in.main_cells;
    if (interaction_cells) *interaction_cells = t->in.interaction_cells;
    *out = t;
    return 0;
}
extern "C" int32_t bfhip_trace_create_ram(bfhip_ctx* ctx, const char* code, const uint8_t* input, size_t n_input, size_t ram_size, bfhip_trace** out,
                                           uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells) {
    try {
        if (!ctx) throw HipError("null context");
        if (!code || !out || (!input && n_input)) throw HipError("null argument");
        ctx->c.bind();
        std::vector<u32> ins = compile(code);
        Machine m(ins, input ? std::vector<u8>(input, input + n_input) : std::vector<u8>(), ram_size ? ram_size : Machine::DEFAULT_RAM_SIZE);
        m.execute();
        return trace_create_common(ctx, m.trace, ins, out, log_sizes, n_steps, main_cells, interaction_cells);
    } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; } catch (...) { bfhip_set_error("unknown error"); return -1; }
}
extern "C" int32_t bfhip_trace_create(bfhip_ctx* ctx, const char* code, const uint8_t* input, size_t n_input, bfhip_trace** out,
                                       uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells) {
    return bfhip_trace_create_ram(ctx, code, input, n_input, 0, out, log_sizes, n_steps, main_cells, interaction_cells);
}
// prove_brainfuck(&Machine) receives an executed machine (mod.rs:471-473): its register trace (mod.rs:508) and its program.
extern "C" int32_t bfhip_trace_create_from_registers(bfhip_ctx* ctx, const uint32_t* trace7, size_t n_rows, const uint32_t* code_words, size_t n_code,
                                                      bfhip_trace** out, uint32_t log_sizes[13], uint64_t* main_cells, uint64_t* interaction_cells) {
    try {
        if (!ctx) throw HipError("null context");
        ctx->c.bind();
        if (!out) throw HipError("null argument");
        if (!trace7 || n_rows == 0) throw HipError("EmptyTrace");
        if (!code_words || n_code == 0) throw HipError("empty program");
        std::vector<Registers> tr(n_rows);
        for (size_t i = 0; i < n_rows; i++) {
            const u32* v = trace7 + 7 * i;
            for (int k = 0; k < 7; k++) if (v[k] >= P31) throw HipError("register value is not a canonical M31");
            tr[i] = Registers{v[0], v[1], v[2], v[3], v[4], v[5], v[6]};
        }
        std::vector<u32> ins(code_words, code_words + n_code);
        for (u32 w : ins) if (w >= P31) throw HipError("program word is not a canonical M31");
        return trace_create_common(ctx, tr, ins, out, log_sizes, nullptr, main_cells, interaction_cells);
    } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; } catch (...) { bfhip_set_error("unknown error"); return -1; }
}
extern "C" int32_t bfhip_trace_destroy(bfhip_ctx* ctx, bfhip_trace* t) { (void)ctx; if (t) { t->in.release(); delete t; } return 0; }

extern "C" int32_t bfhip_prove_trace(bfhip_ctx* ctx, const bfhip_trace* trace, uint32_t log_max_rows, char** proof_json, size_t* proof_len,
                                      char** transcript, double* phase_seconds) {
    try {
        if (!ctx) throw HipError("null context");
        if (!trace) throw HipError("null trace");
        ctx->c.bind();
        HipProver pv(ctx->c, log_max_rows);
        pv.want_transcript = transcript != nullptr;
        BrainfuckProof bp = pv.prove(trace->in);
        fill_outputs(pv, bp, proof_json, proof_len, transcript, phase_seconds);
        return 0;
    } catch (const std::exception& e) { release_group_after_failure(ctx); bfhip_set_error(e.what()); return -1; } catch (...) { release_group_after_failure(ctx); bfhip_set_error("unknown error"); return -1; }
}

extern "C" int32_t bfhip_prove_brainfuck(bfhip_ctx* ctx, const char* code, const uint8_t* input, size_t n_input, uint32_t log_max_rows,
                                          char** proof_json, size_t* proof_len, char** transcript, double* phase_seconds) {
    TraceInput in;
    try {
        if (!ctx) throw HipError("null context");
        if (!code) throw HipError("null program text");
        ctx->c.bind();
        HipProver pv(ctx->c, log_max_rows);
        pv.want_transcript = transcript != nullptr;
        // VM run + table build + upload happen while the GPU already works on the preprocessed commitment
        BrainfuckProof bp = pv.prove([&]() -> const TraceInput& {
            std::vector<u32> ins = compile(code);
            Machine m(ins, std::vector<u8>(input, input + n_input));
            m.execute();
            HipProver::upload_trace(ctx->c, m.trace, ins, in, /*use_arena=*/true, /*on_gpu=*/ctx->c.tables_on_gpu);
            return in;
        });
        fill_outputs(pv, bp, proof_json, proof_len, transcript, phase_seconds);
        in.release();
        return 0;
    } catch (const std::exception& e) { release_group_after_failure(ctx); in.release(); bfhip_set_error(e.what()); return -1; }
    catch (...) { release_group_after_failure(ctx); in.release(); bfhip_set_error("unknown error"); return -1; }
}
extern "C" void bfhip_free_host(void* p) { free(p); }
extern "C" int32_t bfhip_ctx_last_proof_flags(bfhip_ctx* ctx, uint32_t* flags) {
    if (!ctx || !flags) { bfhip_set_error("null argument"); return -1; }
    *flags = ctx->c.last_proof_flags;
    return 0;
}
extern "C" int32_t bfhip_ctx_reuse_preprocessed(bfhip_ctx* ctx, int32_t on) {
    if (on) { preprocessed_cache_of(&ctx->c).enabled = true; return 0; }
    std::lock_guard<std::mutex> g(g_cache_mutex);
    auto& caches = preprocessed_caches();
    auto it = caches.find(&ctx->c);
    if (it != caches.end()) { if (ctx->c.stream) (void)hipStreamSynchronize(ctx->c.stream); it->second.keep.release(); caches.erase(it); }
    return 0;
}

// verify_brainfuck (mod.rs:738-797). Host only. 0 = accepted, 1 = rejected (reason in err), -1 = internal error.
extern "C" int32_t bfhip_verify_brainfuck_conv(const char* proof_json, size_t proof_len, uint32_t log_max_rows, const bfhip_conventions* conv, char* err, size_t err_cap) {
    try {
        Conventions cv;
        if (conv) {
            if (conv->merkle_node_hash > 1 || conv->mix_u64 > 1 || conv->logup_mask_order > 1 || conv->merkle_channel > 1) { bfhip_set_error("unknown convention value"); return -1; }
            cv.merkle_node_hash = conv->merkle_node_hash; cv.mix_u64 = conv->mix_u64; cv.logup_mask_order = conv->logup_mask_order; cv.merkle_channel = conv->merkle_channel;
        }
        std::string reason;
        try { BrainfuckProof bp = proof_from_json(proof_json, proof_len, cv.merkle_channel == 1); reason = verify_brainfuck(bp, log_max_rows, cv); }
        catch (const std::exception& e) { reason = std::string("InvalidStructure: ") + e.what(); }
        if (err && err_cap) snprintf(err, err_cap, "%s", reason.c_str());
        return reason.empty() ? 0 : 1;
    } catch (...) { bfhip_set_error("unknown error"); return -1; }
}

extern "C" int32_t bfhip_verify_brainfuck(const char* proof_json, size_t proof_len, uint32_t log_max_rows, char* err, size_t err_cap) {
    return bfhip_verify_brainfuck_conv(proof_json, proof_len, log_max_rows, nullptr, err, err_cap);
}

// ---- host-only entry points (no GPU needed): compiler, VM and table builders of the drop-in's host side ---------------------------
extern "C" int32_t bfhip_host_compile(const char* code, uint32_t* out, size_t cap, size_t* n) {
    try { auto ins = compile(code); *n = ins.size(); if (ins.size() > cap) { bfhip_set_error("capacity"); return -2; } memcpy(out, ins.data(), 4 * ins.size()); return 0; }
    catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
extern "C" int32_t bfhip_host_run_ram(const char* code, const uint8_t* input, size_t n_input, size_t ram_size, uint8_t* out, size_t out_cap, size_t* n_out,
                                       uint32_t* trace7, size_t trace_cap_rows, size_t* n_rows) {
    try {
        Machine m(compile(code), std::vector<u8>(input, input + n_input), ram_size ? ram_size : Machine::DEFAULT_RAM_SIZE);
        m.execute();
        if (n_out) *n_out = m.output.size();
        if (out && m.output.size() <= out_cap) memcpy(out, m.output.data(), m.output.size());
        if (n_rows) *n_rows = m.trace.size();
        if (trace7 && m.trace.size() <= trace_cap_rows)
            for (size_t i = 0; i < m.trace.size(); i++) { const Registers& r = m.trace[i]; u32 v[7] = {r.clk, r.ip, r.ci, r.ni, r.mp, r.mv, r.mvi}; memcpy(trace7 + 7 * i, v, 28); }
        return 0;
    } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
extern "C" int32_t bfhip_host_run(const char* code, const uint8_t* input, size_t n_input, uint8_t* out, size_t out_cap, size_t* n_out,
                                   uint32_t* trace7, size_t trace_cap_rows, size_t* n_rows) {
    return bfhip_host_run_ram(code, input, n_input, 0, out, out_cap, n_out, trace7, trace_cap_rows, n_rows);
}
// Table of `component` (0..12, claim order of mod.rs:85-99) built from an explicit register trace (7 u32 per row) and compiled program.
extern "C" int32_t bfhip_host_table(const uint32_t* trace7, size_t n_trace, const uint32_t* code, size_t n_code, int32_t component,
                                     uint32_t* out_row_major, size_t cap, size_t* n_rows, size_t* n_cols) {
    try {
        std::vector<Registers> tr(n_trace);
        for (size_t i = 0; i < n_trace; i++) { const u32* v = trace7 + 7 * i; tr[i] = Registers{v[0], v[1], v[2], v[3], v[4], v[5], v[6]}; }
        std::vector<u32> ins(code, code + n_code);
        Table t;
        switch (component) {
            case C_MEMORY: t = memory_table(tr); break;
            case C_INSTRUCTION: t = instruction_table(tr, ins); break;
            case C_PROGRAM: t = program_table(ins); break;
            case C_PROCESSOR: t = processor_table(tr); break;
            case C_JNZ: t = jump_table(tr, OP_JNZ); break;
            case C_JZ: t = jump_table(tr, OP_JZ); break;
            case C_INPUT: t = instruction_sub_table(tr, OP_READCHAR); break;
            case C_LEFT: t = instruction_sub_table(tr, OP_LEFT); break;
            case C_MINUS: t = instruction_sub_table(tr, OP_MINUS); break;
            case C_OUTPUT: t = instruction_sub_table(tr, OP_PUTCHAR); break;
            case C_PLUS: t = instruction_sub_table(tr, OP_PLUS); break;
            case C_RIGHT: t = instruction_sub_table(tr, OP_RIGHT); break;
            case C_EOE: t = eoe_table(tr); break;
            default: bfhip_set_error("bad component"); return -1;
        }
        if (t.n_rows == 0) throw HipError("EmptyTrace");      // TraceError::EmptyTrace (memory/table.rs:83-86 and the six analogues)
        *n_rows = t.n_rows; *n_cols = t.cols.size();
        if (out_row_major) {
            if (t.n_rows * t.cols.size() > cap) { bfhip_set_error("capacity"); return -2; }
            for (size_t r = 0; r < t.n_rows; r++) for (size_t c = 0; c < t.cols.size(); c++) out_row_major[r * t.cols.size() + c] = t.cols[c][r];
        }
        return 0;
    } catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
// Profiler state is per stream, i.e. per context: contexts on other threads are not affected (prof.hip).
static void sync_both(Ctx& c) { c.sync(); if (c.stream2) BF_HIP(hipStreamSynchronize(c.stream2)); for (auto a : c.aux) if (a) BF_HIP(hipStreamSynchronize(a)); }
extern "C" int32_t bfhip_profile_enable(bfhip_ctx* ctx, int32_t mode) {
    try { if (!ctx) throw HipError("null context"); if (mode < 0 || mode > 2) throw HipError("bad profile mode"); ctx->c.bind(); ctx->c.ensure_side(); sync_both(ctx->c); prof_enable(ctx->c.stream, mode); prof_enable(ctx->c.stream2, mode); for (auto a : ctx->c.aux) if (a) prof_enable(a, mode); return 0; }
    catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
extern "C" int32_t bfhip_profile_reset(bfhip_ctx* ctx) {
    try { if (!ctx) throw HipError("null context"); ctx->c.bind(); sync_both(ctx->c); prof_reset(ctx->c.stream); if (ctx->c.stream2) prof_reset(ctx->c.stream2); for (auto a : ctx->c.aux) if (a) prof_reset(a); return 0; }
    catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
extern "C" int32_t bfhip_profile_report(bfhip_ctx* ctx, char** json) {
    try { if (!ctx) throw HipError("null context"); ctx->c.bind(); sync_both(ctx->c); hipStream_t ss[4] = {ctx->c.stream, ctx->c.stream2, ctx->c.aux[0], ctx->c.aux[1]}; std::string s = prof_report_json(ss, 4); *json = (char*)malloc(s.size() + 1); memcpy(*json, s.c_str(), s.size() + 1); return 0; }
    catch (const std::exception& e) { bfhip_set_error(e.what()); return -1; }
}
