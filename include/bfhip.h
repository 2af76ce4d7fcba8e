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

const char* bfhip_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU). */
int32_t bfhip_device_count(void);

/* Context: owns the stream, the twiddle tree and scratch memory.
 * max_log_domain = log2 of the largest evaluation domain that will be used (reference: LOG_MAX_ROWS + log_blowup + 1 = 26, since
 * SimdBackend::precompute_twiddles(CanonicCoset::new(LOG_MAX_ROWS + log_blowup + 2).circle_domain().half_coset), mod.rs:480-484). */
int32_t bfhip_ctx_create(int32_t device_id, uint32_t max_log_domain, bfhip_ctx** out);
int32_t bfhip_ctx_destroy(bfhip_ctx* ctx);
int32_t bfhip_ctx_sync(bfhip_ctx* ctx);

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

/* The two halves of prove_brainfuck, so that a caller (and the benchmark) can keep the prover input resident in HBM:
 * bfhip_trace_create = VM run + the 13 `XTable::from(&vm_trace)` builders (mod.rs:508-547) + upload of the row-granular columns;
 * bfhip_prove_trace  = everything from the preprocessed commitment (mod.rs:493) to the finished proof (mod.rs:734). */
int32_t bfhip_trace_create(bfhip_ctx* ctx, const char* code, const uint8_t* input_h, size_t n_input, bfhip_trace** out,
                           uint32_t log_sizes[13], uint64_t* n_steps, uint64_t* main_cells, uint64_t* interaction_cells);
int32_t bfhip_trace_destroy(bfhip_ctx* ctx, bfhip_trace* trace);
int32_t bfhip_prove_trace(bfhip_ctx* ctx, const bfhip_trace* trace, uint32_t log_max_rows, char** proof_json, size_t* proof_len,
                          char** transcript, double* phase_seconds);

/* Host-only pieces of the drop-in (usable without a GPU): the Brainfuck compiler (crates/brainfuck_vm/src/compiler.rs:17-37), the VM
 * (crates/brainfuck_vm/src/machine.rs:141-238; trace rows are 7 u32: clk, ip, ci, ni, mp, mv, mvi) and the 13 table builders
 * (`XTable::from`, the table.rs files under crates/brainfuck_prover/src/components; component index = claim order of mod.rs:85-99), row-major out. */
int32_t bfhip_host_compile(const char* code, uint32_t* out, size_t cap, size_t* n);
int32_t bfhip_host_run(const char* code, const uint8_t* input_h, size_t n_input, uint8_t* out, size_t out_cap, size_t* n_out,
                       uint32_t* trace7, size_t trace_cap_rows, size_t* n_rows);
int32_t bfhip_host_table(const uint32_t* trace7, size_t n_trace, const uint32_t* code, size_t n_code, int32_t component,
                         uint32_t* out_row_major, size_t cap, size_t* n_rows, size_t* n_cols);

/* Optional per-kernel timing with HIP events on the context's stream (used by bench.py for the roofline object).
 * Report: JSON {"kernel": {"calls": n, "total_ms": t, "bytes": algorithmic_bytes}, ...}, malloc'd (bfhip_free_host). */
int32_t bfhip_profile_enable(bfhip_ctx* ctx, int32_t on);
int32_t bfhip_profile_reset(bfhip_ctx* ctx);
int32_t bfhip_profile_report(bfhip_ctx* ctx, char** json);

#ifdef __cplusplus
}
#endif
#endif /* BFHIP_H */
