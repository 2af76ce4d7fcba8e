// Internal launcher declarations shared by the .hip translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include "m31.h"
#include "air.h"
#include <string>
#include <vector>

namespace bf {

// A column in HBM. shift = 0: one u32 per domain cell. shift = 4: one u32 per table row, standing for a column whose values
// are broadcast into 16 consecutive cells (reference: memory/table.rs:95-104) — cell i reads ptr[i >> 4].
struct ColDesc { const u32* ptr; u32 shift; u32 pad_; };

// prof.hip — optional per-kernel HIP-event timing (bench.py roofline); state is per stream
int prof_mode(hipStream_t s);
void prof_enable(hipStream_t s, int mode);
void prof_run_begin(hipStream_t s, const char* name);   // mode 2: one event pair for a run of back-to-back launches of one kernel
void prof_run_end(hipStream_t s);
void prof_begin(hipStream_t s, const char* name, double bytes, double units, double aux = 0.0);
void prof_end(hipStream_t s);
void prof_reset(hipStream_t s);
std::string prof_report_json(const hipStream_t* streams, int n);
void prof_forget(hipStream_t s);
// bytes = algorithmic bytes of the launch, units = work units (Blake2s compressions / Hades permutations for the Merkle kernels, else 0).
// dominant: the kernel that mode 2 (lowest overhead) instruments — the Merkle layer kernel.
struct ProfScope {
    hipStream_t s; bool on;
    // aux: a second work count of the launch (circle-FFT kernels: butterflies, so that a VALU-bound pass can be priced against the VALU peak)
    ProfScope(hipStream_t s_, const char* name, double bytes, double units = 0.0, bool dominant = false, double aux = 0.0) : s(s_) {
        int m = prof_mode(s);
        on = m == 1 || (m == 2 && dominant);
        if (on) prof_begin(s, name, bytes, units, aux);
    }
    ~ProfScope() { if (on) prof_end(s); }
};

// ---- global address space --------------------------------------------------------------------------------------------------------
// A pointer read out of a descriptor / pointer array in HBM is only known to the compiler as a generic pointer, for which it emits
// FLAT loads and stores: 64-bit VGPR addressing, and they count on lgkmcnt as well as vmcnt, so every wait for an LDS operation
// also waits for the outstanding column traffic. All such pointers are HBM pointers; casting them to address space 1 gives
// global_load / global_store (SGPR base + 32-bit VGPR offset where the base is uniform).
#if defined(__HIPCC__)
#define BF_GLOBAL __attribute__((address_space(1)))
typedef u32 bf_u32x4 __attribute__((ext_vector_type(4)));
typedef BF_GLOBAL u32* g_u32p;
typedef const BF_GLOBAL u32* g_cu32p;
__device__ __forceinline__ g_cu32p as_global(const u32* p) { return (g_cu32p)(unsigned long long)p; }
__device__ __forceinline__ g_u32p as_global(u32* p) { return (g_u32p)(unsigned long long)p; }
__device__ __forceinline__ uint4 ld16(g_cu32p p) { bf_u32x4 v = *(const BF_GLOBAL bf_u32x4*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 ld16_stream(g_cu32p p) { bf_u32x4 v = __builtin_nontemporal_load((const BF_GLOBAL bf_u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st16(g_u32p p, uint4 v) { bf_u32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (BF_GLOBAL bf_u32x4*)p); }   // streamed: no reuse before eviction
// Parameter tables written by the host before the launch and only read by the kernel (staged batches, entries, descriptors): address space 4
// turns a load at a wave-uniform address into s_load (SGPR result, lgkmcnt) instead of a 64-lane flat_load that competes with the column
// traffic for vmcnt and VGPRs.
#define BF_CONSTANT __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ T ld_constant(const T* p) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "word-sized parameter blocks only");
    T out;
    const BF_CONSTANT u32* s = (const BF_CONSTANT u32*)(unsigned long long)p;
#pragma unroll
    for (u32 i = 0; i < sizeof(T) / 4; i++) reinterpret_cast<u32*>(&out)[i] = s[i];
    return out;
}
// value of cell i of a column: 32-bit byte offset (columns hold < 2^30 cells) on the descriptor's base pointer
__device__ __forceinline__ u32 ld_col(const ColDesc& d, u32 i) {
    const u32 byte_off = (i >> d.shift) << 2;
    return *(g_cu32p)((const BF_GLOBAL char*)(unsigned long long)d.ptr + byte_off);
}
#endif

// fft.hip
void gen_twiddles(hipStream_t stream, u32* d_tw, u32* d_itw, u32 R, const uint2* d_tlo, const uint2* d_thi);
// Transforms of columns of 2^log cells, batched over JOBS (one job = the columns of one size and storage).
// inverse: evaluations (bit-reversed) -> coefficients, scaled by 2^-log.
// forward: coefficients of 2^src_log (zero-extended to 2^log) -> evaluations on the canonic domain of size 2^log.
// circle = false selects "line mode" (no circle layer) used for 16x-replicated columns stored row-granular.
// d_src / d_dst: DEVICE arrays of column pointers.
struct FftJob { const u32* const* d_src; u32* const* d_dst; u32 ncols, log, src_log; bool circle; };
// One pass of one job (fft.hip). A launch covers several of them: workgroups [block0, block0 + grid_x * grid_y) belong to this one.
struct PassArgs {
    u32* const* dst;          // device array of column pointers (output / in-place)
    const u32* const* src;    // device array of column pointers (input of this pass)
    u32 ncols, cols_per_block;
    u32 log;                  // transform size
    u32 lo, k;                // layers [lo, lo + k)
    u32 tile_log;             // contiguous pass (lo == 0): tile = 2^tile_log >= 2^k cells; strided: tile = 2^(k + CHUNK_LOG)
    u32 src_mask;             // input index mask (2^src_log - 1): forward zero-extension = wrap-around load
    u32 circle;               // 1: layer 0 is the circle layer; 0: line mode
    u32 scale;                // inverse: multiply outputs by this (1 = none)
    u32 tw_total;             // 2^R
    const u32* tw;            // twiddle (forward) or inverse-twiddle (inverse) layered buffer
    u32 block0, grid_x;       // first workgroup of this group within its launch; tiles per column block
    u32 kind;                 // 0: the launch's own kernel; else the kind (fft.hip: K_PASS, K_TINY) of a small transform that rides in a contiguous-tile launch
};
struct FftLaunch { int kind; u32 first_group, ngroups, total_blocks; double bytes, alg, bfly; };
// fft_plan: host-side layout of every pass of every job; the caller copies plan.groups to device memory the stream can read (the
// staging ring: one copy together with the pointer arrays) and sets d_groups; fft_run issues the launches.
struct FftPlan { bool inverse = false; std::vector<PassArgs> groups; std::vector<FftLaunch> launches; const PassArgs* d_groups = nullptr; };
void fft_plan(FftPlan& plan, bool inverse, const FftJob* jobs, size_t njobs, const u32* tw, const u32* itw, u32 tw_root_log);
void fft_run(hipStream_t stream, const FftPlan& plan);

// Coefficients of IsFirst(n) (indicator of cell 0 of the 2^n-cell canonic domain), n = log_min..log_max, written in closed form by one
// launch: ptr[n - log_min] receives 2^n words (nullptr: skipped).
struct IsFirstCols { u32* ptr[28]; u32 log_min, log_max; };
void is_first_coeffs(hipStream_t stream, const IsFirstCols& a, const u32* itw, u32 tw_root_log);

// merkle.hip
// out_shift / prev_shift: replication of this layer / of the child layer (nodes are stored at index node >> shift)
// first/count: range of stored nodes to compute (count == 0: the whole layer) — a rank of a shard group computes its share only
// node_conv = Conventions::merkle_node_hash
void merkle_layer(hipStream_t stream, void* out, const void* prev, const ColDesc* d_cols, u32 ncols, u32 log, double col_bytes, u32 out_shift, u32 prev_shift,
                  u32 node_conv, u32 first = 0, u32 count = 0);
// A tree's layout, passed BY VALUE to the kernels that walk several levels (merkle_subtree, merkle_top): layer pointers and replication shifts by
// level, the column descriptors of all levels in one array (level `lg` owns cols[col_off[lg] .. col_off[lg - 1]), levels descending,
// col_off[-1] := n_cols).
struct MerkleTreeDesc { uint4* layers[32]; u32 shifts[32]; u32 col_off[32]; const ColDesc* cols; u32 n_cols, max_log; };
// levels [hi .. MERKLE_SUBTREE_ROOT_LEVEL] in one launch, 11 <= hi <= 17, all of them un-replicated; bytes / compressions: profiler accounting
static constexpr u32 MERKLE_SUBTREE_ROOT_LEVEL = 9;
void merkle_subtree(hipStream_t stream, const MerkleTreeDesc& tree, u32 hi, u32 node_conv, double bytes, double compressions);
void merkle_subtree_share(hipStream_t stream, const MerkleTreeDesc& tree, u32 hi, u32 stop, u32 lo, u32 wg0, u32 n_wg, u32 node_conv, double bytes, double compressions);
// levels [top_hi .. 0] by one workgroup, top_hi <= 9 (children of level top_hi from level top_hi + 1 in HBM unless top_hi == max_log);
// d_chan != nullptr: the kernel also performs channel_mix_root_draw on the root it has just computed
void merkle_top(hipStream_t stream, const MerkleTreeDesc& tree, u32 top_hi, u32 node_conv, u32* d_chan, u32* d_alpha8, u32* d_root_copy, double bytes, double compressions,
                u32* d_stamp = nullptr, u32 stamp_value = 0);
// The FRI commit phase below 2^10 rows as ONE single-workgroup launch (merkle.hip: k_fri_tail): per layer the Merkle tree of its 4 coordinate
// columns, the channel step (mix_root, draw alpha), the fold into the next layer (+ fold-in of the quotient of that size), all through LDS;
// evaluations, hashes, roots and alphas also go to HBM for the decommitment and the host's channel replay.
struct FriTailLayer {
    u32* ev[4];             // this layer's evaluations (2^log rows), already in HBM for the first layer, written by the kernel for the others
    const u32* quot[4];     // circle evaluation of 2^log rows folded into the NEXT layer, or nullptr
    uint4* tree[11];        // tree[lg] = level lg of this layer's Merkle tree, lg <= log
};
struct FriTailArgs {
    u32 n_layers, top_log;  // layers of 2^top_log, 2^(top_log - 1), ... rows are committed; top_log <= 10
    u32 alpha_idx, root_idx;// layer k draws alpha[8 * (alpha_idx + k)] and copies its root to roots[8 * (root_idx + k)]
    u32* ev_last[4];        // the layer after the last committed one (2^(top_log - n_layers) rows)
    u32* chan; u32* alpha; u32* roots;
    const u32* itw; u32 tw_total, rfc;
    FriTailLayer layer[10];
};
void fri_tail(hipStream_t stream, const FriTailArgs* d_args, double bytes, double compressions);   // bytes / compressions: profiler accounting (tree nodes only)
// One FRI inner layer of 2^log rows (11 <= log <= 16) in ONE launch: fold of the previous layer (src, 2^(log + 1) rows) with alpha8 (+ fold-in of
// `quot`, a circle evaluation of 2^(log + 1) rows, or nullptr), the layer's evaluations (dst), its Merkle tree (tree[lg] = level lg) and the
// channel step (mix_root, draw -> alpha_out[8]; root copy -> root_out). counter: one zero-initialised u32 in HBM (reset by the kernel).
struct FriLayerArgs {
    const u32* src[4]; const u32* quot[4]; u32* dst[4]; uint4* tree[18];
    const u32* alpha8; const u32* itw; u32 tw_total, log, rfc, pad_;
    u32* counter; u32* chan; u32* alpha_out; u32* root_out;
};
void fri_layer(hipStream_t stream, const FriLayerArgs& a);
void grind_span(hipStream_t stream, const u32* d_digest, u64 base, u32 span, u32 pow_bits, unsigned long long* d_best, u32 mix_u64_conv);
// diagnostic (bfhip_clock_probe): register-only Blake2s loop, per-workgroup {d s_memtime, d s_memrealtime} stamps
void clock_probe_launch(hipStream_t stream, uint4* d_stamps, u32* d_sink, u32 blocks, u32 iters);
// diagnostic (bfhip_clock_probe_mix): a one-wave clock sampler that runs beside the real Merkle kernel; pseudo-random fill of its input layer
void clock_sampler_launch(hipStream_t stream, uint4* d_out, const u32* d_stop_alias, unsigned long long max_ticks);
void fill_mix(hipStream_t stream, u32* p, size_t n);
// Blake2sChannel::mix_root(root) + draw_felt() on the device. d_chan = digest[8] || n_sent; d_alpha8 receives alpha[4] || alpha^2[4],
// d_root_copy a copy of the root.
void channel_mix_root_draw(hipStream_t stream, u32* d_chan, const u32* d_root, u32* d_alpha8, u32* d_root_copy);

// poseidon.hip — Poseidon252 Merkle variant (BASELINE config 5; not used by the reference)
void merkle_layer_poseidon(hipStream_t stream, void* out, const void* prev, const ColDesc* d_cols, u32 ncols, u32 log, u32 out_shift = 0, u32 prev_shift = 0,
                           u32 first = 0, u32 count = 0);
void hades_once(hipStream_t stream, const u32* d_in24, u32* d_out24);

// air.hip
struct ConstraintLaunch {
    const u32* is_first; ColDesc trace[13]; ColDesc inter[12]; u32* acc[4]; Q31 coeff[12]; Lookups el; Q31 total_sum; u32 denom_inv[2]; u32 log_size;
    u32 overwrite;   // 1: acc = value (first component of an accumulator: no zero fill, no read); 0: acc += value
    // Row range of the 2^(log_size+1)-row evaluation domain this launch covers (n_rows == 0: all rows) — a rank of a shard group evaluates
    // its contiguous share only. Column / accumulator pointers are then "virtual bases": valid for the rows of the range.
    u32 row0, n_rows;
    // Previous-row copy of the component's LAST logUp column (4 coordinates): prev[k][row] = inter[last + k][row at offset -1]. nullptr:
    // the kernel reads the neighbour itself (it generally lives in another rank's row range — SURVEY.md section 7 vii).
    const u32* inter_prev[4];
};
// `d_args` points to a ConstraintLaunch staged in device memory.
// n_rows: the launch's row count when d_args->n_rows != 0 (host copy of the same value), else 0
// group_rows: constraint_group_rows() of the host copy of *d_args (0: one lane per row; 32: as 16 with two 4096-row blocks per workgroup; 16: one AIR evaluation per group of 16 rows whose
// replicated columns share a stored cell)
void eval_constraints(hipStream_t stream, int comp, const ConstraintLaunch* d_args, u32 log_size, u32 n_rows = 0, u32 group_rows = 0);
u32 constraint_group_rows(const ConstraintLaunch& L, int comp);
// All components of a proof in one launch: classes = components of equal log_size (one accumulator each). The first component of a class
// must carry overwrite = 1 and the others 0 (row-group mode updates the accumulator per component; per-row mode writes the class sum once).
struct ConstraintClass { u32 n_comps, block0, group_rows, pad_; u32 comp[13]; u32 pad2_[3]; };
struct ConstraintBatch { u32 n_classes, total_blocks, pad_[2]; ConstraintClass cls[13]; };
void constraint_batch_init(ConstraintBatch& b, const ConstraintLaunch* launches, u32 n);
void eval_constraints_batch(hipStream_t stream, const ConstraintBatch* d_batch, const ConstraintBatch& h_batch, const ConstraintLaunch* d_args);
struct LogupLaunch {
    const u32* cols[13];   // row-granular main columns
    u32* out_rep[8];       // row-granular coordinate columns of the non-last logUp columns
    u32* out_last[4];      // full-size coordinate columns of the last logUp column (16 * 2^log_rows cells)
    void* vrow; void* wloc; void* totals; void* claimed;   // scratch: uint4[M], uint4[M], uint4[M/1024 + 2], uint4[1]
    Lookups el; u32 log_rows; int comp;
};
// The logUp generation of up to 13 components as ONE batch of four launches. LogupBatch is the device-side table (the caller copies it to
// memory the stream can read — the staging ring — and passes both the device address and the host copy).
struct LogupItem { const u32* cols[13]; u32* out_rep[8]; u32* out_last[4]; uint4* vrow; uint4* wloc; uint4* totals; uint4* claimed; u32 log_rows; int comp; u32 nb; u32 pad_; };
struct LogupBatch { Lookups el; u32 n; u32 rows_blk0[14], scan_blk0[14], last_blk0[14]; u32 pad_; LogupItem item[13]; };
void logup_batch_init(LogupBatch& b, const Lookups& el, const LogupLaunch* L, u32 n);
void logup_batch_run(hipStream_t stream, const LogupBatch* d_batch, const LogupBatch& h_batch);
void broadcast16(hipStream_t stream, const u32* d_rows, u32* d_out, u32 n_cells);

// quotient.hip
struct EvalJob { const u32* coeffs; u32 log_n; u32 point; u32 factor_shift; u32 partial_off; u32 out_idx; u32 pad_; };   // result -> out[out_idx]
void eval_at_points(hipStream_t stream, const EvalJob* d_jobs, u32 n_jobs, u32 total_partials, const void* d_factors, void* d_partials, void* d_out);
struct QuotientBatch { C31 prx, pry, pix, piy; Q31 a_sum, b_sum, batch_coeff; u32 n_cols; C31 kden; u32 n_full; };   // kden = prx * piy - pry * pix; n_full: see QuotientEntry
// One sampled column of a batch: its constant and where its cells are (cell of row i = ptr[i >> shift], shift 0 or >= 2). Within a batch the
// entries with shift == 0 come first (QuotientBatch::n_full of them): quotient_entries_finish orders them and fills ptr / shift from `col`.
struct QuotientEntry { Q31 c; const u32* ptr; u32 shift; u32 col; };
void quotient_entries_finish(QuotientBatch* batches, size_t n_batches, QuotientEntry* entries, const ColDesc* cols);
// row0 / n_rows: range of rows to compute (n_rows == 0: all 2^log rows; both multiples of 4); out pointers may be virtual bases
// block0: first workgroup of this size group within the one launch that covers all of them (set by quotient_groups_layout)
struct QuotientArgs { const QuotientBatch* batches; const QuotientEntry* entries; u32 n_batches; u32 log; const u32* tw; u32 tw_total; u32* out[4]; u32 row0, n_rows; u32 block0, pad_; };
// every size group of a proof in one launch: fill h_groups, call quotient_groups_layout (sets block0, returns the grid size), copy the table
// to device memory the stream can read, launch
u32 quotient_groups_layout(QuotientArgs* h_groups, u32 n_groups);
void accumulate_quotients(hipStream_t stream, const QuotientArgs* d_groups, u32 n_groups, u32 total_blocks);
// d_alpha8: device pointer to alpha[4] || alpha^2[4]
// fresh: dst holds nothing yet (treated as zero, not read)
// first / count: range of DESTINATION cells to compute (count == 0: all 2^(log-1)); pointers may be virtual bases
void fold_circle_into_line(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log, bool fresh = false,
                           u32 first = 0, u32 count = 0);
void fold_line(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log, u32 first = 0, u32 count = 0);
// dst = fold_line(src of 2^log rows, alpha) and, when quot != nullptr (a circle evaluation of 2^log rows), dst = dst * alpha^2 + fold_circle(quot, alpha)
void fold_line_circle(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* const quot[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log,
                      u32 first = 0, u32 count = 0);
// dst[i] = src[index of the row at trace-coset offset -1 of LDE row i] over a whole column of 2^(log_size+1) cells (blowup 2) — the
// "previous-row copy" a column owner materialises before the column is cut into row ranges
void prev_row_copy(hipStream_t stream, u32* dst, const u32* src, u32 log_size);
// out[out_off + w] = base[index + w] for w < n_words (n_words = 1: a column cell, 8: a hash); base == nullptr reads as zeros
struct GatherReq { const u32* base; u64 index; u32 out_off; u32 n_words; };
void gather_u32(hipStream_t stream, const GatherReq* d_req, u32 n, u32* d_out);
void accumulate(hipStream_t stream, u32* dst, const u32* src, u32 n);
struct AccumulateSizes { u32* dst[4]; const u32* src[12][4]; u32 log[12]; u32 n; };   // <= 12 sources (13 components: at most 13 distinct sizes), sorted by descending size, all <= 2^log of dst
void accumulate_sizes(hipStream_t stream, const AccumulateSizes& a);
void batch_inverse_m31(hipStream_t stream, const u32* src, u32* dst, u32 n);
void batch_inverse_qm31(hipStream_t stream, const u32* const src[4], u32* const dst[4], u32 n);
void bit_reverse(hipStream_t stream, const u32* src, u32* dst, u32 log);
void one_hot(hipStream_t stream, u32* dst, u32 n);

}  // namespace bf
