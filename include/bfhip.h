/* bfhip — C ABI of the MI355X-native Circle-STARK prover backend for the Brainfuck zkVM.
 *
 * This is the drop-in boundary: every entry point replaces one operation of stwo's backend trait surface
 * (`Backend = ColumnOps + FieldOps + PolyOps + QuotientOps + FriOps + AccumulationOps` + `MerkleOps<H>` + `GrindOps<C>`)
 * which the reference fixes to `SimdBackend` at crates/brainfuck_prover/src/brainfuck_air/mod.rs:56,399,480,486-487,497,732 and
 * crates/brainfuck_prover/src/components/mod.rs:42. A Rust `HipBackend` binds these with `extern "C"`.
 *
 * Conventions
 *  - every function returns int32 status: 0 = ok, <0 = error (bfhip_last_error() gives the text); no exceptions cross the boundary.
 *  - `*_d` pointers are device pointers obtained from bfhip_malloc; `*_h` are host pointers borrowed for the call only.
 *  - M31 values are canonical u32 in [0, 2^31-1); QM31 values are 4 x u32; secure columns are 4 separate u32 columns (SoA).
 *  - columns are in bit-reversed circle-domain order, exactly like stwo's CircleEvaluation<_, _, BitReversedOrder>.
 *  - one ctx = one GPU + one HIP stream; calls on a ctx are serial; distinct ctxs may be driven from distinct threads.
 */
#ifndef BFHIP_H
#define BFHIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfhip_ctx bfhip_ctx;
typedef struct bfhip_trace bfhip_trace;

/* Byte-level conventions of the un-vendored stwo-prover @ 31e8dbc (Cargo.toml:41) that cannot be confirmed offline (SURVEY.md Appendix B.2;
 * DESIGN.md section 6 lists every such point). Each is one named switch so that a comparison against a real stwo proof can flip one at a
 * time. The zero value of every field is the default = the published stwo code of that period to the best of our reconstruction.
 *  merkle_node_hash  Blake2sMerkleHasher::hash_node (core/vcs/blake2_merkle.rs), used by Blake2sMerkleChannel (mod.rs:56,486-487):
 *      BFHIP_MERKLE_STWO_COMPRESS  state = 0^32; if children: state = compress(state, left || right, 0,0,0,0); then for the column values
 *                                  in zero-padded chunks of 16 words: state = compress(state, chunk, 0,0,0,0). No parameter block, byte
 *                                  counter or final flag (blake2s_ref::compress is the bare RFC 7693 F function).
 *      BFHIP_MERKLE_RFC7693        node = Blake2s-256(left || right || LE u32 values), the standard hash of the same byte string.
 *  mix_u64           Blake2sChannel::mix_u64 (core/channel/blake2s.rs), used for the claim (components/mod.rs:133) and the proof of work:
 *      BFHIP_MIX_U64_COMPRESS      digest = compress(digest words, [n_lo, n_hi, 0 x 14], 0,0,0,0) — what SimdBackend's grind searches.
 *      BFHIP_MIX_U64_HASH          digest = Blake2s-256(digest || LE64(n) zero padded to 32 bytes) (form of older revisions).
 *  logup_mask_order  mask offsets of each component's last logUp column in LogupAtRow::finalize (reached from finalize_logup(),
 *                    e.g. components/memory/component.rs:133): BFHIP_LOGUP_MASK_CUR_PREV = [0, -1], BFHIP_LOGUP_MASK_PREV_CUR = [-1, 0].
 *                    Changes the order of those columns' two sampled values (and with it the transcript). */
enum { BFHIP_MERKLE_STWO_COMPRESS = 0, BFHIP_MERKLE_RFC7693 = 1 };
enum { BFHIP_MIX_U64_COMPRESS = 0, BFHIP_MIX_U64_HASH = 1 };
enum { BFHIP_LOGUP_MASK_CUR_PREV = 0, BFHIP_LOGUP_MASK_PREV_CUR = 1 };
/* merkle_channel is not a convention but the protocol variant, carried in the same struct so that it reaches prover and verifier alike:
 *      BFHIP_CHANNEL_BLAKE2S      Blake2sMerkleChannel — what the reference instantiates (mod.rs:56,486-487). Default.
 *      BFHIP_CHANNEL_POSEIDON252  Poseidon252MerkleChannel (BASELINE.json config 5; an upstream stwo capability over starknet-crypto 0.6.2,
 *                                 Cargo.lock:821-864, which the reference never instantiates): Poseidon252MerkleHasher nodes, Poseidon252Channel
 *                                 transcript, felt252 hashes serialised as "0x.." hex strings. merkle_node_hash and mix_u64 do not apply. */
enum { BFHIP_CHANNEL_BLAKE2S = 0, BFHIP_CHANNEL_POSEIDON252 = 1 };
typedef struct bfhip_conventions {
    uint32_t merkle_node_hash;
    uint32_t mix_u64;
    uint32_t logup_mask_order;
    uint32_t merkle_channel;
    uint32_t reserved[4];   /* must be zero */
} bfhip_conventions;

const char* bfhip_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU). */
int32_t bfhip_device_count(void);
/* hipMemGetInfo of one device: bytes free right now and in total (either pointer may be NULL). What a caller sizes a pool with, and what
 * the leak test of a failed bfhip_ctx_create compares before / after. */
int32_t bfhip_device_memory(int32_t device_id, uint64_t* free_bytes, uint64_t* total_bytes);

/* Context: owns the stream, the twiddle tree and scratch memory.
 * max_log_domain = log2 of the largest evaluation domain that will be used (reference: LOG_MAX_ROWS + log_blowup + 1 = 26, since
 * SimdBackend::precompute_twiddles(CanonicCoset::new(LOG_MAX_ROWS + log_blowup + 2).circle_domain().half_coset), mod.rs:480-484).
 * Range [6, 29]: columns of up to 2^29 cells (LOG_MAX_ROWS <= 27). */
/* A context owns a main HIP stream, creates its side stream at its first proof (the preprocessed commitment runs there) and two partner streams on
 * demand (bfhip_ctx_set_overlap, shard groups). HIP gives every stream one of GPU_MAX_HW_QUEUES (default 4) hardware queues at creation; with two
 * streams per context up to four proofs in flight per process need no setting (profiles/r05_inflight_history.txt). Several proofs in flight on one GPU:
 * a pool (bfhip_pool_create below) behind one caller thread, or one context and one host thread each.
 * A creation that fails (no device, out of memory at the twiddle tree, ...) releases everything it had allocated. */
int32_t bfhip_ctx_create(int32_t device_id, uint32_t max_log_domain, bfhip_ctx** out);
int32_t bfhip_ctx_destroy(bfhip_ctx* ctx);
int32_t bfhip_ctx_sync(bfhip_ctx* ctx);
/* Intra-proof overlap on the context's partner streams (events only, bytes unchanged); default 0 = off. bit 1: the FRI first-layer tree is
 * hashed level by level behind the quotient launches (compute_fri_quotients inside mod.rs:732); bit 0: the Merkle layers of a tree's largest
 * columns are hashed while its smaller columns are still being transformed (tree_builder.commit, mod.rs:500,583,723). Both pairs of kernels
 * are VALU-limited on gfx950 and stretch each other when they co-run: measured gain 0-0.3 ms of 31 (profiles/r03_overlap_ab*.txt).
 * bit 2 (shard groups only): the column -> row send-receive of a tree's largest size class is issued on the partner stream and overlaps the
 * transforms of the tree's smaller columns; the rest follows in a second send-receive on the same stream (every rank of the group must use
 * the same mask: it changes the number of collectives). A context that never called this function (and was not created under BFHIP_OVERLAP)
 * behaves as if bit 2 were set while it is a member of a group whose ranks sit on DIFFERENT GPUs (RCCL groups; in-process groups over several
 * devices from their second collective on): there the exchange is an xGMI transfer. Calling it with bit 2 clear switches that default off.
 * Unmeasured on multi-GPU hardware. */
int32_t bfhip_ctx_set_overlap(bfhip_ctx* ctx, uint32_t mask);
/* Device memory of this context in bytes: out[0] = reserved by its per-proof arena (1 GiB chunks, kept until the context is destroyed),
 * out[1] = the arena's high-water mark over all proofs so far, out[2] = twiddle tree + inverse, out[3] = arena bytes in use now. */
int32_t bfhip_ctx_memory(bfhip_ctx* ctx, uint64_t out[4]);
/* Host waits of this context: 0 (default) = poll briefly, then yield / block; 1 = hipStreamSynchronize at once (hosts with more waiting
 * contexts than cores). Waits inside a shard group are always bounded polls (BFHIP_COMM_TIMEOUT_S, default 300 s).
 * Mailboxes (small proofs, LOG_MAX_ROWS <= 21: the launches behind a Fiat-Shamir point are enqueued before the host knows the challenge and a
 * one-workgroup kernel waits on the GPU for the host's post) are OFF by default under policy 1 — the kernel would spin at the head of a
 * hardware queue while the host thread sleeps. Where they are on (policy 0, or forced with BFHIP_MAILBOX=1), a host thread that is stalled for
 * longer than BFHIP_MAILBOX_TIMEOUT_MS (default 10 000 ms: SIGSTOP, a debugger, heavy oversubscription) makes that proof FAIL with a
 * "mailbox kernel gave up waiting for the host" error instead of merely being slow; the context stays usable and the next proof starts
 * clean. At most one proof per GPU of a process runs in the mailbox order at a time (two could block each other through a shared hardware queue): with several
 * proofs in flight the others keep the plain order. BFHIP_MAILBOX=0 switches them off altogether. All members of a group must use the same overlap mask (bfhip_ctx_set_overlap). */
int32_t bfhip_ctx_set_sync_policy(bfhip_ctx* ctx, int32_t blocking);
/* Mailbox settings of a live context (what BFHIP_MAILBOX / BFHIP_MAILBOX_TIMEOUT_MS set at creation):
 * mode -2 = keep the current mode, -1 = automatic (see above), 0 = off, 1 = on; timeout_ms 0 = keep the current timeout; test_delay_ms: the
 * host sleeps that long before every post — a test hook that exists only in the test-hooks build of the library (libbfhip_testhooks.so,
 * -DBFHIP_TEST_HOOKS: there BFHIP_MAILBOX_TEST_DELAY_MS presets it); the default build accepts <= 0 (no delay) and rejects anything else.
 * Takes effect at the next proof. */
int32_t bfhip_ctx_set_mailbox(bfhip_ctx* ctx, int32_t mode, uint32_t timeout_ms, int32_t test_delay_ms);

/* Conventions used by every operation of this context (prover, bfhip_merkle_commit_layer, bfhip_grind). conv == NULL restores the defaults. */
int32_t bfhip_ctx_set_conventions(bfhip_ctx* ctx, const bfhip_conventions* conv);
int32_t bfhip_ctx_get_conventions(bfhip_ctx* ctx, bfhip_conventions* out);

/* Device buffers (ColumnOps storage: BaseColumn / SecureColumnByCoords live in HBM behind these). */
int32_t bfhip_malloc(bfhip_ctx* ctx, size_t bytes, void** out_d);
int32_t bfhip_free(bfhip_ctx* ctx, void* ptr_d);
int32_t bfhip_upload(bfhip_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
int32_t bfhip_download(bfhip_ctx* ctx, void* dst_h, const void* src_d, size_t bytes);
int32_t bfhip_memset_zero(bfhip_ctx* ctx, void* dst_d, size_t bytes);

/* PolyOps::precompute_twiddles — done once in bfhip_ctx_create; this returns the device buffers (layered like stwo's TwiddleTree
 * rooted at Coset::half_odds(max_log_domain - 1)): 2^(max_log_domain-1) u32 each. */
int32_t bfhip_twiddles(bfhip_ctx* ctx, const uint32_t** tw_d, const uint32_t** itw_d, uint32_t* root_log);

/* PolyOps::interpolate_columns (tree_builder.extend_evals, mod.rs:497,550-562,690-702): n_cols evaluations of 2^log_size cells on
 * CanonicCoset(log_size).circle_domain() -> coefficients of the same size. cols_h = host array of device pointers. In place when
 * dst == src. replicated != 0: the columns hold one value per *table row* (2^(log_size-4) cells) standing for a column whose
 * values are broadcast 16x (memory/table.rs:95-104); the output then holds the 2^(log_size-4) coefficients of index = 0 mod 16
 * (all other coefficients of such a column are zero). */
int32_t bfhip_interpolate(bfhip_ctx* ctx, uint32_t* const* src_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, int32_t replicated);
/* PolyOps::evaluate_polynomials (tree_builder.commit -> LDE, mod.rs:500,583,723): coefficients of 2^log_size -> evaluations on
 * CanonicCoset(log_eval).circle_domain(), log_eval >= log_size. replicated as above (output is row-granular, 2^(log_eval-4) cells). */
int32_t bfhip_evaluate(bfhip_ctx* ctx, uint32_t* const* coeff_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, uint32_t log_eval, int32_t replicated);

/* gen_is_first::<B>(log_size) followed by interpolate (mod.rs:497 `tree_builder.extend_evals(gen_is_first(..))`, one call per
 * log_size in log_min..=log_max): the coefficients of the indicator of cell 0, written in closed form by ONE launch (no transform).
 * dst_cols_h[n - log_min] = device pointer to 2^n words, or NULL to skip that size. 4 <= log_min <= log_max < log_min + 28. */
int32_t bfhip_is_first_coeffs(bfhip_ctx* ctx, uint32_t log_min, uint32_t log_max, uint32_t* const* dst_cols_h);

/* ---- single backend operations (each is what one stwo trait method would call; all reached from mod.rs:732 prover::prove unless noted) ----
 * A secure (QM31) column is passed as 4 coordinate pointers. Values/points passed from the host are u32[4] per QM31. */

/* The 16-lane broadcast of `trace_evaluation` (memory/table.rs:95-104: trace[col].data[row] = value.into()): dst[16 r + l] = rows[r].
 * Only for callers that want the full-size column; the prover itself keeps such columns row-granular. */
int32_t bfhip_broadcast16(bfhip_ctx* ctx, const uint32_t* rows_d, uint32_t* dst_d, size_t n_rows);
/* ColumnOps::bit_reverse_column: dst[bit_reverse(i)] = src[i] (out of place), 2^log_size cells. */
int32_t bfhip_bit_reverse(bfhip_ctx* ctx, const uint32_t* src_d, uint32_t* dst_d, uint32_t log_size);
/* FieldOps::batch_inverse over M31: dst[i] = src[i]^-1, src[i] != 0. */
int32_t bfhip_batch_inverse_m31(bfhip_ctx* ctx, const uint32_t* src_d, uint32_t* dst_d, size_t n);
/* FieldOps::batch_inverse over QM31 (the PackedSecureField denominators of LogupTraceGenerator::finalize_col, memory/table.rs:513):
 * 4 coordinate columns in, 4 out (may alias), n elements, none zero. */
int32_t bfhip_batch_inverse_qm31(bfhip_ctx* ctx, const uint32_t* const src_d[4], uint32_t* const dst_d[4], size_t n);
/* AccumulationOps::accumulate: dst[i] += src[i] in M31. */
int32_t bfhip_accumulate(bfhip_ctx* ctx, uint32_t* dst_d, const uint32_t* src_d, size_t n);
/* PolyOps::eval_at_point (CommitmentSchemeProver::prove_values): f(P) for P = (x, y) in QM31^2 given as u32[8] = x[4] || y[4];
 * coeffs_d holds 2^log_size coefficients (replicated != 0: the 2^(log_size-4) coefficients of index 0 mod 16). out_h = u32[4]. */
int32_t bfhip_eval_at_point(bfhip_ctx* ctx, const uint32_t* coeffs_d, uint32_t log_size, int32_t replicated, const uint32_t point_h[8], uint32_t out_h[4]);
/* MerkleOps<Blake2sMerkleHasher>::commit_on_layer (tree_builder.commit, mod.rs:500,583,723): 2^log_size nodes,
 * node i = hash_node(prev[2i], prev[2i+1], [cols[k][i >> col_shift[k]] for k < n_cols]) under the context's merkle_node_hash convention;
 * prev_layer_d may be NULL (deepest layer). col_shifts_h may be NULL (all 0). Hashes are 32-byte records. */
int32_t bfhip_merkle_commit_layer(bfhip_ctx* ctx, uint32_t log_size, const void* prev_layer_d, const uint32_t* const* cols_h, const uint32_t* col_shifts_h,
                                  uint32_t n_cols, void* out_hashes_d);
/* MerkleOps<Poseidon252MerkleHasher>::commit_on_layer (upstream stwo capability named by BASELINE.json config 5; the reference itself
 * only uses Blake2s — SURVEY.md F9): node i = poseidon_hash_many([prev[2i], prev[2i+1]]? ++ blocks), block = 8 M31 column values packed
 * as w = w * 2^31 + v (zero padded). Hashes are felt252 values stored as 8 little-endian u32 limbs (32 bytes), canonical form. */
int32_t bfhip_merkle_commit_layer_poseidon252(bfhip_ctx* ctx, uint32_t log_size, const void* prev_layer_d, const uint32_t* const* cols_h,
                                             const uint32_t* col_shifts_h, uint32_t n_cols, void* out_hashes_d);
/* One Hades permutation (Starknet Poseidon, width 3) of three canonical felt252 values, 8 LE u32 limbs each — the known-answer-test hook. */
int32_t bfhip_hades_permutation(bfhip_ctx* ctx, const uint32_t in_h[24], uint32_t out_h[24]);
/* FriOps::fold_line: 2^log_size evaluations over LineDomain(Coset::half_odds(log_size)) -> 2^(log_size-1); alpha_h = u32[4]. */
int32_t bfhip_fold_line(bfhip_ctx* ctx, const uint32_t* const src_d[4], uint32_t* const dst_d[4], uint32_t log_size, const uint32_t alpha_h[4]);
/* FriOps::fold_circle_into_line: dst = dst * alpha^2 + fold(src); src has 2^log_size cells on CanonicCoset(log_size).circle_domain(). */
int32_t bfhip_fold_circle_into_line(bfhip_ctx* ctx, uint32_t* const dst_d[4], const uint32_t* const src_d[4], uint32_t log_size, const uint32_t alpha_h[4]);
/* GrindOps::grind for Blake2sChannel: smallest nonce such that mix_u64(nonce) on `digest_h` (32 bytes) leaves >= pow_bits trailing zeros. */
int32_t bfhip_grind(bfhip_ctx* ctx, const uint8_t digest_h[32], uint32_t pow_bits, uint64_t* nonce);
/* Decommitment reads: out_h[j] = col_d[idx_h[j]] for n positions of one column. */
int32_t bfhip_gather(bfhip_ctx* ctx, const uint32_t* col_d, const uint64_t* idx_h, size_t n, uint32_t* out_h);

/* ---- per-component AIR operations (what `ComponentProver` / `LogupTraceGenerator` / `QuotientOps` of a HipBackend would call) ----
 * component: 0 memory, 1 instruction, 2 program, 3 processor, 4 jump-if-not-zero, 5 jump-if-zero, 6 input, 7 left, 8 minus, 9 output,
 * 10 plus, 11 right, 12 end_of_execution (the numbering of `bfhip_trace_column`). lookup_h = u32[24]: (z[4], alpha[4]) of the Memory, Instruction and Processor relations, in that order
 * (mod.rs:589-597 draws them in this order). */

/* Shape of one component: main-trace columns, logUp (secure) columns, constraints. Returns -1 for an unknown component. */
int32_t bfhip_component_shape(int32_t component, uint32_t* n_main_cols, uint32_t* n_logup_cols, uint32_t* n_constraints);
/* `interaction_trace_evaluation` of one component (e.g. memory/table.rs:485-518, processor/table.rs:456-529): LogupTraceGenerator
 * new_col / write_frac / finalize_col / finalize_last. main_rows_h: n_main device pointers to row-granular columns (2^(log_size-4)
 * rows). out_cols_h: 4 * n_logup device pointers to coordinate columns in bit-reversed circle-domain order; the last 4 (the prefix-
 * summed column) hold 2^log_size cells, earlier ones are 16-lane replicated and are written row-granular (2^(log_size-4) cells).
 * claimed_sum_h = u32[4] receives the component's claimed sum. */
int32_t bfhip_logup_generate(bfhip_ctx* ctx, int32_t component, uint32_t log_size, const uint32_t* const* main_rows_h, const uint32_t lookup_h[24],
                             uint32_t* const* out_cols_h, uint32_t claimed_sum_h[4]);
/* `ComponentProver::evaluate_constraint_quotients_on_domain` of one component (FrameworkComponent<XEval>, components/<name>/component.rs):
 * every constraint of the component on CanonicCoset(log_size + 1).circle_domain(), each multiplied by its random-coefficient power
 * and by the inverse of the trace-domain vanishing polynomial, added into acc_d (4 coordinate columns of 2^(log_size+1) cells).
 * is_first_d: the preprocessed IsFirst LDE (2^(log_size+1) cells). main_lde_h / inter_lde_h: device pointers to the n_main and
 * 4 * n_logup LDE columns; *_shifts_h (may be NULL = 0) give the replication shift of each (cell i is read at index i >> shift).
 * coeffs_h = u32[4 * n_constraints]: the coefficient of constraint j in evaluation order (the caller reverses the accumulator's
 * power slice the way `accum.columns()` does). claimed_sum_h = the component's logUp total. */
int32_t bfhip_eval_constraints(bfhip_ctx* ctx, int32_t component, uint32_t log_size, const uint32_t* is_first_d, const uint32_t* const* main_lde_h,
                               const uint32_t* main_shifts_h, const uint32_t* const* inter_lde_h, const uint32_t* inter_shifts_h, const uint32_t lookup_h[24],
                               const uint32_t claimed_sum_h[4], const uint32_t* coeffs_h, uint32_t* const acc_d[4]);
/* `QuotientOps::accumulate_quotients` (compute_fri_quotients, reached from mod.rs:732) for the columns of one LDE size: n_cols
 * columns of 2^log_size cells on CanonicCoset(log_size).circle_domain() (bit-reversed; col_shifts_h as above: 0 or >= 2, may be NULL).
 * Column k has n_samples_h[k] samples; sample_points_h (u32[8] each: x[4] || y[4]) and sample_values_h (u32[4] each) list them column
 * by column. Batches are formed per distinct point in BTreeMap order like ColumnSampleBatch::new_vec. out_d: 4 coordinate columns. */
int32_t bfhip_accumulate_quotients(bfhip_ctx* ctx, uint32_t log_size, const uint32_t* const* cols_h, const uint32_t* col_shifts_h, uint32_t n_cols,
                                   const uint32_t* n_samples_h, const uint32_t* sample_points_h, const uint32_t* sample_values_h,
                                   const uint32_t random_coeff_h[4], uint32_t* const out_d[4]);

/* prove_brainfuck (crates/brainfuck_prover/src/brainfuck_air/mod.rs:471-735), device resident: compiles and runs `code` on the
 * host VM (crates/brainfuck_vm), builds the 13 component tables, then commits, evaluates constraints, samples, builds the FRI
 * quotients, runs FRI, grinds and decommits on the GPU. *proof_json receives the serde_json form of BrainfuckProof (mod.rs:71-76),
 * malloc'd — release with bfhip_free_host. log_max_rows = LOG_MAX_ROWS (mod.rs:428: 24; 20 under cfg(test), mod.rs:433); the
 * context must have been created with max_log_domain >= log_max_rows + 2.
 * transcript (optional): channel digests after each stage, "name:hex\n". phase_seconds (optional): 10 doubles
 * {preprocessed, tables(host), main_trace, interaction, composition, oods, quotients, fri(+pow+decommit), decommit, total}. */
int32_t bfhip_prove_brainfuck(bfhip_ctx* ctx, const char* code, const uint8_t* input_h, size_t n_input, uint32_t log_max_rows,
                              char** proof_json, size_t* proof_len, char** transcript, double* phase_seconds);
void bfhip_free_host(void* p);
/* What the last COMPLETED proof of this context actually did (0 before the first): bit 0 = it ran in the mailbox order (forcing mode 1 does not
 * guarantee it: one proof per GPU holds that order at a time, see bfhip_ctx_set_sync_policy), bit 1 = it took the context's kept preprocessed tree
 * (bfhip_ctx_reuse_preprocessed), bit 2 = it took a pool's shared preprocessed tree (bfhip_pool_set_preprocessed), bit 3 = shard group: the transforms were replicated
 * (bfhip_ctx_set_shard_policy). Tests and tools read it so that
 * a setting that silently did not apply is visible. */
int32_t bfhip_ctx_last_proof_flags(bfhip_ctx* ctx, uint32_t* flags);
/* ---- one proof over several GPUs (shard group) ------------------------------------------------------------------------------------------
 * north_star: "trace columns shard naturally across the 8 GPUs of one node, with the Merkle root and FRI fold reduced via RCCL over xGMI"
 * (SURVEY.md section 8(e)). The ranks of a group prove ONE trace together and all return the byte-identical proof of the single-GPU path:
 *   - the full-size columns of the interaction and composition trees are COLUMN-sharded for interpolation / LDE (greedy by size), then one
 *     grouped send-receive cuts every LDE column into contiguous bit-reversed ROW ranges (contiguous ranges of a bit-reversed circle domain
 *     are sub-cosets), together with a previous-row copy of each component's last logUp column (its mask offset -1 is not a halo);
 *   - Merkle subtrees, constraint evaluation, FRI quotients and the FRI folds down to 2^14 rows per rank are row-local; one all-gather per
 *     tree completes the layer of 256 nodes per rank, the top is hashed redundantly so that every rank feeds the same root to its channel;
 *   - out-of-domain samples and decommitment words are each produced by one rank and completed by a max-reduce (exact: zero elsewhere).
 * Every exchange is issued by the library on the context's own stream with device buffers on both ends; the host program supplies no
 * callbacks. Two transports:
 *   RCCL  — one process per GPU: rank 0 calls bfhip_rccl_unique_id, the host program hands the 128 bytes to the other ranks (any control
 *           channel: torch.distributed, MPI, a file), every rank calls bfhip_ctx_join_rccl_group. librccl is loaded on first use.
 *   local — the N contexts belong to one process and are driven by N host threads; they may share a GPU (how the tests run N ranks on a
 *           one-GPU box) or own one each (peer copies between the GPUs, ordered by HIP events).
 * count must be a power of two in [2, 64]. Every rank must issue the same sequence of prove calls on the same trace. */
typedef struct bfhip_local_group bfhip_local_group;
int32_t bfhip_local_group_create(uint32_t count, bfhip_local_group** out);
int32_t bfhip_local_group_destroy(bfhip_local_group* group);
int32_t bfhip_ctx_join_local_group(bfhip_ctx* ctx, bfhip_local_group* group, uint32_t rank);
int32_t bfhip_rccl_unique_id(uint8_t id[128]);
int32_t bfhip_ctx_join_rccl_group(bfhip_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t count);
int32_t bfhip_ctx_leave_group(bfhip_ctx* ctx);
/* How a shard group divides a proof (every rank of a group the same value; takes effect at the next proof; byte-identical proofs either way):
 *   0  exchange: the transforms are column-sharded and one grouped send-receive per tree cuts the LDE columns into row ranges (the default of rounds 2-5;
 *      1.03 / 0.84 / 0.72 GB per fib19 proof and rank at 2 / 4 / 8 ranks, over N - 1 xGMI links per GPU);
 *   1  replicate the transforms: every rank interpolates and extends every column and evaluates the constraints on every row itself (the transforms are
 *      20 % of a proof), and ONLY the Merkle hashing, the quotient rows and the FRI folds are divided by row range — no column -> row exchange at all,
 *      what travels is the per-tree all-gather of 256 nodes per rank and the max-reduces;
 *  -1  automatic (default): 1 for a group of two ranks on different GPUs (their exchange would cross ONE xGMI link: 13.5 ms at 76 GB/s against 5.6 ms of
 *      transforms), 0 otherwise. Unmeasured on multi-GPU hardware: DESIGN.md section 7 has the arithmetic. */
int32_t bfhip_ctx_set_shard_policy(bfhip_ctx* ctx, int32_t policy);
/* Since the group was joined: out = {all-gathers, max-reduces, grouped send-receives, payload bytes this rank sent to other ranks}. */
int32_t bfhip_ctx_group_stats(bfhip_ctx* ctx, uint64_t out[4]);
/* GPU-side milliseconds this rank's stream spent inside {all-gathers, max-reduces, grouped send-receives} since the group was joined (one
 * HIP-event pair per collective): the communication share of a proof over several GPUs. Synchronises the context's stream. */
int32_t bfhip_ctx_group_times(bfhip_ctx* ctx, double out_ms[3]);
/* Per-collective latency of this rank since the group was joined (or since the last call with reset != 0), for {all-gathers, max-reduces, grouped
 * send-receives} in that order, 7 numbers each: {count, GPU-side p50, p90, max, host-side p50, p90, max} in microseconds. GPU side = the HIP-event pair
 * around the collective on its stream (includes waiting for the slowest peer); host side = the time the calling thread spent inside the transport's
 * call (RCCL: the enqueue). A proof over N GPUs issues ~31 collectives of mostly small payloads: on hardware their latency decides, not their bytes.
 * Synchronises the context's stream. */
int32_t bfhip_ctx_group_latency(bfhip_ctx* ctx, int32_t reset, double out_us[21]);
/* Test entry: joins an RCCL group (unique id, rank, count), runs ONE grouped send-receive on the given blocks and leaves. With the real library
 * the blocks must be device memory. In the test-hooks build of the library (libbfhip_testhooks.so, -DBFHIP_TEST_HOOKS) the environment variable
 * BFHIP_RCCL_LIBRARY names a test double of the RCCL entry points (tests/mock_rccl.c: host-memory blocks, no GPU needed); the default build
 * ignores that variable and loads librccl only. stats_out (optional) = the counters of bfhip_ctx_group_stats. */
int32_t bfhip_rccl_exchange_raw(const uint8_t id[128], uint32_t rank, uint32_t count, uint32_t n_sends, const uint32_t* send_peer, void* const* send_ptr,
                                const size_t* send_bytes, uint32_t n_recvs, const uint32_t* recv_peer, void* const* recv_ptr, const size_t* recv_bytes,
                                uint64_t stats_out[4]);
/* Exercises the RCCL transport with a one-rank communicator on this context's GPU (library load, communicator, all-gather, max-reduce,
 * grouped exchange): what a single-GPU box can check of the multi-process path. */
int32_t bfhip_rccl_selftest(bfhip_ctx* ctx);
/* rank / count (0 / 1 outside a group) and a static description of the transport. Any pointer may be NULL. */
int32_t bfhip_ctx_group_info(bfhip_ctx* ctx, uint32_t* rank, uint32_t* count, const char** transport);

/* Optional (off by default): keep the preprocessed tree (IsFirst(LOG_MAX_ROWS..=4): polynomials, LDE columns, Merkle layers, root) of
 * the first proof in the context and reuse it for later proofs with the same LOG_MAX_ROWS. The reference recommits it in every
 * prove_brainfuck call (mod.rs:495-500); proof bytes are identical either way. Call with on = 0 before bfhip_ctx_destroy to release it.
 * Joining or leaving a shard group drops the cached tree (it is rebuilt by the next proof): the ranks of a group must all reuse or all rebuild,
 * which holds when every rank sets this option the same way and starts its membership with an empty cache. */
int32_t bfhip_ctx_reuse_preprocessed(bfhip_ctx* ctx, int32_t on);

/* The two halves of prove_brainfuck, so that a caller (and the benchmark) can keep the prover input resident in HBM:
 * bfhip_trace_create = VM run + the 13 `XTable::from(&vm_trace)` builders (mod.rs:508-547) + upload of the row-granular columns;
 * bfhip_prove_trace  = everything from the preprocessed commitment (mod.rs:493) to the finished proof (mod.rs:734). */
int32_t bfhip_trace_create(bfhip_ctx* ctx, const char* code, const uint8_t* input_h, size_t n_input, bfhip_trace** out,
                           uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells);
/* The same with the VM's RAM size given (MachineBuilder::with_ram_size, machine.rs:56-60; `--ram-size` of bin/brainfuck_prover.rs);
 * ram_size = 0 selects Machine::DEFAULT_RAM_SIZE = 30000 (machine.rs:114). */
int32_t bfhip_trace_create_ram(bfhip_ctx* ctx, const char* code, const uint8_t* input_h, size_t n_input, size_t ram_size, bfhip_trace** out,
                               uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells);
/* What prove_brainfuck(&Machine) actually receives (mod.rs:471-473): an EXECUTED machine — its register trace (`inputs.trace()`, mod.rs:508;
 * n_rows rows of 7 u32: clk, ip, ci, ni, mp, mv, mvi, the final ci = ni = 0 row included) and its compiled program (`inputs.program()`,
 * n_code words incl. the jump-target words). No re-execution: the rows are uploaded as they are and the 13 tables are built from them.
 * Values must be canonical M31 (< 2^31 - 1); the trace must be non-empty. */
int32_t bfhip_trace_create_from_registers(bfhip_ctx* ctx, const uint32_t* trace7_h, size_t n_rows, const uint32_t* code_words_h, size_t n_code,
                                          bfhip_trace** out, uint32_t log_sizes[13], uint64_t* main_cells, uint64_t* interaction_cells);
int32_t bfhip_trace_destroy(bfhip_ctx* ctx, bfhip_trace* trace);
/* Where the 13 `XTable::from(&vm_trace)` builders (mod.rs:511-547) of this context run: 1 = on the GPU (default; sorts, clk-gap fill, padding,
 * pairing and per-opcode selection as gfx950 kernels), 0 = host builders + upload. Results are identical. */
int32_t bfhip_ctx_set_table_builder(bfhip_ctx* ctx, int32_t on_gpu);
/* Row-granular main-trace column `column` of component `component` (claim order) of a resident trace -> host. out_h may be NULL (size query). */
int32_t bfhip_trace_column(bfhip_ctx* ctx, const bfhip_trace* trace, uint32_t component, uint32_t column, uint32_t* out_h, size_t cap, size_t* n_rows);
int32_t bfhip_prove_trace(bfhip_ctx* ctx, const bfhip_trace* trace, uint32_t log_max_rows, char** proof_json, size_t* proof_len,
                          char** transcript, double* phase_seconds);

/* ---- proofs in flight: a pool of sub-contexts on one GPU behind ONE caller thread -------------------------------------------------------------
 * The reference's caller is a single thread of control (prove_brainfuck, mod.rs:471-735); a single proof leaves the GPU partly idle in its
 * single-workgroup chains (tree tops, small FRI layers) and at its Fiat-Shamir round trips. A pool proves the proofs of a batch n_in_flight at a
 * time on internal worker threads (one sub-context each) and returns when all are done: +19 % (2 in flight) / +23 % (3) throughput at 2^22 rows,
 * +36..57 % at 2^20 rows, +10 % for fib19. Proof bytes are those of bfhip_prove_trace, proof by proof.
 * The sub-contexts share the twiddle tree and point tables of the first one, and — by default — ONE preprocessed commitment per batch:
 *   bfhip_pool_set_preprocessed(pool, mode): 0 = every proof recommits IsFirst(LOG_MAX_ROWS..=4) as the reference does in every prove_brainfuck
 *   call (mod.rs:495-500); 1 (default) = once per batch, committed by a builder context beside the first proofs' main-trace phase; 2 = kept
 *   across batches while LOG_MAX_ROWS and the hasher stay the same. Byte-neutral: the tree depends on LOG_MAX_ROWS and the hasher only.
 * Traces of a batch must be resident on the pool's device: create them with bfhip_trace_create*(bfhip_pool_ctx(pool, i), ...) — any i — between
 * batches (a sub-context must not be used by the caller while a batch runs). n_in_flight in [1, 16]; 2-3 is where the gain saturates. */
typedef struct bfhip_pool bfhip_pool;
int32_t bfhip_pool_create(int32_t device_id, uint32_t n_in_flight, uint32_t max_log_domain, bfhip_pool** out);
int32_t bfhip_pool_destroy(bfhip_pool* pool);
int32_t bfhip_pool_size(bfhip_pool* pool, uint32_t* n_in_flight);
/* Sub-context i (borrowed: owned by the pool, never pass it to bfhip_ctx_destroy): for bfhip_trace_create*, per-context settings, memory queries. */
int32_t bfhip_pool_ctx(bfhip_pool* pool, uint32_t i, bfhip_ctx** out);
/* bfhip_ctx_set_conventions on every sub-context and on the builder of the shared preprocessed tree (conv == NULL: the defaults). */
int32_t bfhip_pool_set_conventions(bfhip_pool* pool, const bfhip_conventions* conv);
int32_t bfhip_pool_set_preprocessed(bfhip_pool* pool, int32_t mode);
/* n x bfhip_prove_trace. Outputs are arrays of n entries, each optional (NULL): proofs_json[i] (malloc'd, bfhip_free_host; NULL when proof i failed),
 * proof_lens[i], statuses[i] (0 = ok, < 0 = that proof's error). seconds (optional) has n + 1 entries: each proof's own wall time from its start on
 * its worker, then the wall time of the whole batch. Returns 0 when every proof succeeded, -1 otherwise (bfhip_last_error: the first failed
 * proof's message); the other proofs of the batch are still delivered. */
int32_t bfhip_prove_batch(bfhip_pool* pool, const bfhip_trace* const* traces, uint32_t n, uint32_t log_max_rows, char** proofs_json, size_t* proof_lens,
                          int32_t* statuses, double* seconds);
/* n x bfhip_prove_brainfuck: VM run, table build and upload of proof i happen inside its worker, beside the other workers' GPU work.
 * inputs_h may be NULL (no program reads input); n_inputs[i] bytes at inputs_h[i] otherwise. Outputs as above. */
int32_t bfhip_prove_batch_brainfuck(bfhip_pool* pool, const char* const* codes, const uint8_t* const* inputs_h, const size_t* n_inputs, uint32_t n,
                                    uint32_t log_max_rows, char** proofs_json, size_t* proof_lens, int32_t* statuses, double* seconds);

/* verify_brainfuck (mod.rs:738-797): replays the channel, checks the logUp sum (mod.rs:207-227), the OODS consistency, the proof of work,
 * every Merkle decommitment and FRI. Host only (the reference verifies on the CPU as well). Returns 0 = accepted, 1 = rejected with the
 * reason written to err (VerificationError name), -1 = internal error. */
int32_t bfhip_verify_brainfuck(const char* proof_json, size_t proof_len, uint32_t log_max_rows, char* err, size_t err_cap);
/* The same under explicit conventions (NULL = defaults): a proof verifies only under the conventions it was produced with. */
int32_t bfhip_verify_brainfuck_conv(const char* proof_json, size_t proof_len, uint32_t log_max_rows, const bfhip_conventions* conv, char* err, size_t err_cap);

/* Host-only pieces of the drop-in (usable without a GPU): the Brainfuck compiler (crates/brainfuck_vm/src/compiler.rs:17-37), the VM
 * (crates/brainfuck_vm/src/machine.rs:141-238; trace rows are 7 u32: clk, ip, ci, ni, mp, mv, mvi) and the 13 table builders
 * (`XTable::from`, the table.rs files under crates/brainfuck_prover/src/components; component index = claim order of mod.rs:85-99), row-major out. */
int32_t bfhip_host_compile(const char* code, uint32_t* out, size_t cap, size_t* n);
int32_t bfhip_host_run(const char* code, const uint8_t* input_h, size_t n_input, uint8_t* out, size_t out_cap, size_t* n_out,
                       uint32_t* trace7, size_t trace_cap_rows, size_t* n_rows);
/* bfhip_host_run with the RAM size given (Machine::new_with_config, machine.rs:116-131); ram_size = 0: the default 30000 cells. */
int32_t bfhip_host_run_ram(const char* code, const uint8_t* input_h, size_t n_input, size_t ram_size, uint8_t* out, size_t out_cap, size_t* n_out,
                           uint32_t* trace7, size_t trace_cap_rows, size_t* n_rows);
int32_t bfhip_host_table(const uint32_t* trace7, size_t n_trace, const uint32_t* code, size_t n_code, int32_t component,
                         uint32_t* out_row_major, size_t cap, size_t* n_rows, size_t* n_cols);

/* Optional per-kernel timing with HIP events on the context's stream (used by bench.py for the roofline object).
 * Report: JSON {"kernel": {"calls": n, "total_ms": t, "bytes": algorithmic_bytes}, ...}, malloc'd (bfhip_free_host). */
/* mode: 0 off, 1 every instrumented kernel, 2 only k_merkle_layer (the dominant kernel; lowest overhead). */
int32_t bfhip_profile_enable(bfhip_ctx* ctx, int32_t mode);
int32_t bfhip_profile_reset(bfhip_ctx* ctx);
int32_t bfhip_profile_report(bfhip_ctx* ctx, char** json);
/* Diagnostic: the shader clock this device sustains under the dominant kernel's instruction mix — a register-only loop of the Blake2s compression
 * (no memory traffic) launched back to back for `seconds` (>= 0.5 for a settled clock), every workgroup stamping the shader-cycle counter against
 * the constant 100 MHz counter around its loop. out = {median GHz over the workgroups of the last launch, min, max, 10^9 compressions/s of the last
 * launches, launches issued, ms per launch}. bench.py prices the Merkle kernel's VALU fraction against the nominal 2.4 GHz AND against this clock:
 * devices of one model differ by up to 12 % on compute-bound loops, and a line that only knows the nominal clock cannot tell a slow device from a
 * slow kernel. Never part of a proof. */
int32_t bfhip_clock_probe(bfhip_ctx* ctx, double seconds, double out[6]);
/* The same question under the REAL dominant kernel: k_merkle_layer itself (an inner layer of 2^log_nodes nodes, 16 <= log_nodes <= 26, over pseudo-random hashes: VALU plus its memory traffic) is
 * launched back to back for `seconds` while a one-wave sampler on the context's other stream stamps the shader-cycle counter against the 100 MHz counter from before the
 * first launch until the host stops it — no stamp executes in the kernel itself. out = {GHz the device held under that mix, 10^9 compressions/s of the kernel on this shape,
 * launches, us per launch, 1 if the sampler spanned the window (0: it ran into its own bound and the clock is not to be trusted), seconds the sampler covered}. A device can
 * hold its clock under the register-only loop of bfhip_clock_probe and still be slow in a proof (measured, r06): probing the proof's own largest layer shape (log_nodes 25:
 * 3 GiB of hashes in flight) is what tells. Uses 96 B x 2^log_nodes of device memory for the duration of the call. Never part of a proof. */
int32_t bfhip_clock_probe_mix(bfhip_ctx* ctx, double seconds, uint32_t log_nodes, double out[6]);

#ifdef __cplusplus
}
#endif
#endif /* BFHIP_H */
