// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
// C ABI over the CPU restatement so that tests/ (ctypes), __graft_entry__.smoke() and bench.py's cpu_baseline leg can call it.
// Nothing in the shipped product links or loads this library.
#include "json.h"
#include <chrono>
#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown"; return -1; }

static std::vector<Registers> run_program(const char* code, const u8* input, size_t n_in, std::vector<u32>* code_out, std::vector<u8>* output) {
    std::vector<u32> ins = compile(code);
    Machine m(ins, std::vector<u8>(input, input + n_in));
    m.execute();
    if (code_out) *code_out = ins;
    if (output) *output = m.output;
    return m.trace;
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
void orc_free(void* p) { free(p); }
int orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n); return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}
// SIMD mode of the port (oracle/simd_port.cpp): 1 = the Merkle layer loop and the circle transforms run on AVX-512 (bench.py's cpu_baseline,
// the stand-in for SimdBackend + rayon); 0 (default) = the scalar checker code. Returns the mode in force (0 when the host lacks AVX-512).
int orc_set_simd(int on) { orc::simd::set_enabled(on != 0); return orc::simd::enabled() ? 1 : 0; }
int orc_simd_available() { return orc::simd::available() ? 1 : 0; }

// Byte-level stwo conventions (field.h `Conventions`; same numbering as include/bfhip.h `bfhip_conventions`). Process-wide: set it before a call.
int orc_set_conventions(u32 merkle_node_hash, u32 mix_u64, u32 logup_mask_order, u32 merkle_channel) {
    if (merkle_node_hash > 1 || mix_u64 > 1 || logup_mask_order > 1 || merkle_channel > 1) { g_err = "bad convention value"; return -1; }
    conventions().merkle_node_hash = merkle_node_hash; conventions().mix_u64 = mix_u64; conventions().logup_mask_order = logup_mask_order;
    conventions().merkle_channel = merkle_channel;
    return 0;
}
// Hades permutation of three canonical felt252 values (4 LE u64 limbs each) and poseidon_hash_many of n of them — pinned against
// oracle/poseidon252.py (big-integer restatement with the public Hades known-answer vector) in tests/test_oracle_poseidon.py
int orc_hades(const u64 in[12], u64 out[12]) {
    Felt s[3]; for (int k = 0; k < 3; k++) s[k] = Felt::from_canonical(in + 4 * k);
    hades_permutation(s);
    for (int k = 0; k < 3; k++) s[k].to_canonical(out + 4 * k);
    return 0;
}
int orc_poseidon_hash_many(const u64* in, size_t n, u64 out[4]) {
    std::vector<Felt> v; for (size_t i = 0; i < n; i++) v.push_back(Felt::from_canonical(in + 4 * i));
    poseidon_hash_many(v).to_canonical(out);
    return 0;
}
// Blake2sMerkleHasher::hash_node under the current convention (left/right may be NULL)
int orc_hash_node(const u8* left, const u8* right, const u32* vals, size_t n, u8 out[32]) {
    Hash32 l, r; if (left) { memcpy(l.b, left, 32); memcpy(r.b, right, 32); }
    Hash32 h = hash_node(left ? &l : nullptr, left ? &r : nullptr, vals, n); memcpy(out, h.b, 32); return 0;
}

// ---- fields ------------------------------------------------------------------------------------------------------------
// op: 0 add, 1 sub, 2 mul, 3 inv(a)
int orc_m31_op(int op, const u32* a, const u32* b, u32* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        M31 x(a[i]), y(b ? b[i] : 0);
        out[i] = (op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : inv(x)).v;
    }
    return 0;
}
int orc_qm31_op(int op, const u32* a, const u32* b, u32* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        QM31 x = QM31::from_u32(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
        QM31 y = b ? QM31::from_u32(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]) : QM31::zero();
        QM31 r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : inv(x);
        auto o = r.to_u32();
        for (int k = 0; k < 4; k++) out[4 * i + k] = o[k];
    }
    return 0;
}

// ---- circle / FFT ------------------------------------------------------------------------------------------------------------
int orc_domain_point(u32 log_size, u32 index, u32 out_xy[2]) { PointM p = CanonicCoset{log_size}.circle_domain().at(index); out_xy[0] = p.x.v; out_xy[1] = p.y.v; return 0; }
// twiddle buffer of Coset::half_odds(log)  (length 2^log), and its inverse
int orc_twiddles(u32 root_log, u32* tw, u32* itw) {
    ORC_TRY
    TwiddleTree t = precompute_twiddles(Coset::half_odds(root_log));
    for (size_t i = 0; i < t.twiddles.size(); i++) { tw[i] = t.twiddles[i].v; if (itw) itw[i] = t.itwiddles[i].v; }
    return 0;
    ORC_CATCH
}
// in-place on n_cols contiguous columns of 2^log_size
int orc_circle_interpolate(u32* values, u32 log_size, size_t n_cols) {
    ORC_TRY
    TwiddleTree t = precompute_twiddles(CanonicCoset{log_size}.circle_domain().half_coset);
#pragma omp parallel for
    for (size_t c = 0; c < n_cols; c++) circle_interpolate(reinterpret_cast<M31*>(values + (c << log_size)), log_size, t);
    return 0;
    ORC_CATCH
}
// coeffs (2^log_size per column) -> evals (2^eval_log per column) on CanonicCoset(eval_log).circle_domain()
int orc_circle_evaluate(const u32* coeffs, u32 log_size, u32 eval_log, size_t n_cols, u32* out) {
    ORC_TRY
    TwiddleTree t = precompute_twiddles(CanonicCoset{eval_log}.circle_domain().half_coset);
#pragma omp parallel for
    for (size_t c = 0; c < n_cols; c++) {
        u32* o = out + (c << eval_log);
        memcpy(o, coeffs + (c << log_size), sizeof(u32) << log_size);
        memset(o + (size_t(1) << log_size), 0, (sizeof(u32) << eval_log) - (sizeof(u32) << log_size));
        circle_evaluate(reinterpret_cast<M31*>(o), eval_log, t);
    }
    return 0;
    ORC_CATCH
}
int orc_eval_at_point(const u32* coeffs, u32 log_size, const u32 point[8], u32 out[4]) {
    PointQ p(QM31::from_u32(point[0], point[1], point[2], point[3]), QM31::from_u32(point[4], point[5], point[6], point[7]));
    auto r = eval_at_point(reinterpret_cast<const M31*>(coeffs), log_size, p).to_u32();
    for (int k = 0; k < 4; k++) out[k] = r[k];
    return 0;
}

// FRI folds on secure columns given as 4 coordinate arrays (stwo backend/cpu/fri.rs).
int orc_fold_line(const u32* const src[4], u32 log, const u32 alpha[4], u32* const dst[4]) {
    ORC_TRY
    size_t n = size_t(1) << log;
    std::vector<QM31> ev(n);
    for (size_t i = 0; i < n; i++) ev[i] = QM31::from_u32(src[0][i], src[1][i], src[2][i], src[3][i]);
    auto out = fold_line(ev, log, QM31::from_u32(alpha[0], alpha[1], alpha[2], alpha[3]));
    for (size_t i = 0; i < out.size(); i++) { auto a = out[i].to_u32(); for (int k = 0; k < 4; k++) dst[k][i] = a[k]; }
    return 0;
    ORC_CATCH
}
int orc_fold_circle_into_line(u32* const dst[4], const u32* const src[4], u32 log, const u32 alpha[4]) {
    ORC_TRY
    size_t n = size_t(1) << log;
    SecureCol s; s.init(log);
    for (int k = 0; k < 4; k++) memcpy(s.c[k].data(), src[k], 4 * n);
    std::vector<QM31> d(n / 2);
    for (size_t i = 0; i < n / 2; i++) d[i] = QM31::from_u32(dst[0][i], dst[1][i], dst[2][i], dst[3][i]);
    fold_circle_into_line(d, s, QM31::from_u32(alpha[0], alpha[1], alpha[2], alpha[3]));
    for (size_t i = 0; i < n / 2; i++) { auto a = d[i].to_u32(); for (int k = 0; k < 4; k++) dst[k][i] = a[k]; }
    return 0;
    ORC_CATCH
}
// Nonce search on an explicit digest.
u64 orc_grind_digest(const u8 digest[32], u32 pow_bits) { Channel c; memcpy(c.digest.b, digest, 32); return grind(c, pow_bits); }

// ---- hashing / channel / Merkle ---------------------------------------------------------------------------------------------
int orc_blake2s(const u8* data, size_t len, u8 out[32]) { Hash32 h = Blake2s::hash(data, len); memcpy(out, h.b, 32); return 0; }
// cols: n pointers, logs: n log sizes. Writes every layer's hashes contiguously, deepest layer first, if layers_out != NULL.
int orc_merkle_commit(const u32* const* cols, const u32* logs, size_t n, u8 root[32], u8* layers_out) {
    ORC_TRY
    std::vector<ColRef> r;
    for (size_t i = 0; i < n; i++) r.push_back({cols[i], logs[i]});
    MerkleProver mp = MerkleProver::commit(r);
    memcpy(root, mp.root().b, 32);
    if (layers_out) { u8* o = layers_out; for (int l = (int)mp.layers.size() - 1; l >= 0; l--) { memcpy(o, mp.layers[l].data(), 32 * mp.layers[l].size()); o += 32 * mp.layers[l].size(); } }
    return 0;
    ORC_CATCH
}
// Channel scripting for tests: ops encoded as a byte stream is overkill; expose the primitives on an opaque handle.
void* orc_channel_new() { return new Channel(); }
void orc_channel_free(void* c) { delete (Channel*)c; }
void orc_channel_digest(void* c, u8 out[32]) { memcpy(out, ((Channel*)c)->digest.b, 32); }
void orc_channel_mix_root(void* c, const u8 root[32]) { Hash32 h; memcpy(h.b, root, 32); ((Channel*)c)->mix_root(h); }
void orc_channel_mix_u64(void* c, u64 v) { ((Channel*)c)->mix_u64(v); }
void orc_channel_mix_felts(void* c, const u32* f, size_t n) { std::vector<QM31> v; for (size_t i = 0; i < n; i++) v.push_back(QM31::from_u32(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3])); ((Channel*)c)->mix_felts(v.data(), n); }
void orc_channel_draw_felt(void* c, u32 out[4]) { auto a = ((Channel*)c)->draw_felt().to_u32(); for (int k = 0; k < 4; k++) out[k] = a[k]; }
u64 orc_channel_grind(void* c, u32 pow_bits) { return grind(*(Channel*)c, pow_bits); }
u32 orc_channel_trailing_zeros(void* c) { return ((Channel*)c)->trailing_zeros(); }

// ---- VM / tables ------------------------------------------------------------------------------------------------------------
int orc_compile(const char* code, u32* out, size_t cap, size_t* n) {
    ORC_TRY
    auto ins = compile(code);
    *n = ins.size();
    if (ins.size() > cap) { g_err = "cap"; return -2; }
    memcpy(out, ins.data(), 4 * ins.size());
    return 0;
    ORC_CATCH
}
// trace_out: 7 u32 per row (clk, ip, ci, ni, mp, mv, mvi)
int orc_run(const char* code, const u8* input, size_t n_in, u8* out, size_t out_cap, size_t* n_out, u32* trace_out, size_t trace_cap_rows, size_t* n_rows) {
    ORC_TRY
    std::vector<u8> output;
    auto tr = run_program(code, input, n_in, nullptr, &output);
    if (n_out) *n_out = output.size();
    if (out && output.size() <= out_cap) memcpy(out, output.data(), output.size());
    if (n_rows) *n_rows = tr.size();
    if (trace_out && tr.size() <= trace_cap_rows)
        for (size_t i = 0; i < tr.size(); i++) { u32 v[7] = {tr[i].clk, tr[i].ip, tr[i].ci, tr[i].ni, tr[i].mp, tr[i].mv, tr[i].mvi}; memcpy(trace_out + 7 * i, v, 28); }
    return 0;
    ORC_CATCH
}
// Component table in row-major form (n_rows x n_cols). Pass out == NULL to query the shape.
int orc_table(const char* code, const u8* input, size_t n_in, int component, u32* out, size_t cap, size_t* n_rows, size_t* n_cols) {
    ORC_TRY
    std::vector<u32> ins;
    auto tr = run_program(code, input, n_in, &ins, nullptr);
    auto tables = build_tables(tr, ins);
    const Table& t = tables.at(component);
    *n_rows = t.n_rows; *n_cols = t.cols.size();
    if (out) {
        if (t.n_rows * t.cols.size() > cap) { g_err = "cap"; return -2; }
        for (size_t r = 0; r < t.n_rows; r++) for (size_t c = 0; c < t.cols.size(); c++) out[r * t.cols.size() + c] = t.cols[c][r];
    }
    return 0;
    ORC_CATCH
}
// Table built from an explicit register trace + compiled program (mirrors the reference's table unit tests).
int orc_table_from_registers(const u32* trace7, size_t n_trace, const u32* code, size_t n_code, int component, u32* out, size_t cap, size_t* n_rows, size_t* n_cols) {
    ORC_TRY
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
        default: t = eoe_table(tr); break;
    }
    if (t.n_rows == 0) throw std::runtime_error("EmptyTrace");      // TraceError::EmptyTrace (e.g. memory/table.rs:83-86): what trace_evaluation answers
    *n_rows = t.n_rows; *n_cols = t.cols.size();
    if (out) {
        if (t.n_rows * t.cols.size() > cap) { g_err = "cap"; return -2; }
        for (size_t r = 0; r < t.n_rows; r++) for (size_t c = 0; c < t.cols.size(); c++) out[r * t.cols.size() + c] = t.cols[c][r];
    }
    return 0;
    ORC_CATCH
}
int orc_log_sizes(const char* code, const u8* input, size_t n_in, u32 out[13], u64* n_steps) {
    ORC_TRY
    std::vector<u32> ins;
    auto tr = run_program(code, input, n_in, &ins, nullptr);
    auto tables = build_tables(tr, ins);
    for (int c = 0; c < N_COMPONENTS; c++) out[c] = tables[c].log_size();
    if (n_steps) *n_steps = tr.size();
    return 0;
    ORC_CATCH
}

// AIR satisfiability on the trace domain (stwo assert_constraints analogue) with given lookup elements (z, alpha per relation:
// memory, instruction, processor; 8 u32 each). Returns 0 if all rows satisfy all constraints; else 1 and (*bad_row, *bad_constraint).
int orc_assert_constraints(const char* code, const u8* input, size_t n_in, int component, const u32* elems24, int corrupt_col, size_t corrupt_row, u32 corrupt_val,
                           size_t* bad_row, int* bad_constraint) {
    ORC_TRY
    std::vector<u32> ins;
    auto tr = run_program(code, input, n_in, &ins, nullptr);
    auto tables = build_tables(tr, ins);
    Table& t = tables.at(component);
    if (corrupt_col >= 0) t.cols.at(corrupt_col).at(corrupt_row) = corrupt_val;
    InteractionElements el;
    auto q = [&](int i) { return QM31::from_u32(elems24[4 * i], elems24[4 * i + 1], elems24[4 * i + 2], elems24[4 * i + 3]); };
    el.memory = LookupElements::make(q(0), q(1)); el.instruction = LookupElements::make(q(2), q(3)); el.processor = LookupElements::make(q(4), q(5));
    u32 log = t.log_size();
    std::vector<std::vector<u32>> main_cols;
    for (auto& c : t.cols) main_cols.push_back(broadcast16(c));
    QM31 claimed;
    auto inter = gen_interaction_trace(component, t, el, &claimed);
    std::vector<u32> isf(size_t(1) << log, 0); isf[0] = 1;
    std::vector<const u32*> tc, ic;
    for (auto& c : main_cols) tc.push_back(c.data());
    for (auto& c : inter) ic.push_back(c.data());
    for (size_t row = 0; row < (size_t(1) << log); row++) {
        AssertEvaluator ae;
        ae.is_first_col = isf.data(); ae.trace_cols = tc.data(); ae.inter_cols = ic.data(); ae.row = row; ae.log_size = log; ae.total_sum = claimed;
        eval_component(component, ae, el);
        if (ae.failed >= 0) { if (bad_row) *bad_row = row; if (bad_constraint) *bad_constraint = ae.failed; return 1; }
    }
    return 0;
    ORC_CATCH
}

// The same check on a caller-supplied table (rows: n_main row-granular columns of n_rows values, column-major) — the reference's negative AIR
// tests patch single cells of a built table (memory/component.rs:211-609). value4 receives the first failing constraint's value.
int orc_assert_constraints_table(int component, const u32* rows, size_t n_rows, const u32* elems24, size_t* bad_row, int* bad_constraint, u32 value4[4]) {
    ORC_TRY
    Table t; t.init(N_MAIN_COLS[component], n_rows);
    for (size_t c = 0; c < t.cols.size(); c++) memcpy(t.cols[c].data(), rows + c * n_rows, n_rows * sizeof(u32));
    InteractionElements el;
    auto q = [&](int i) { return QM31::from_u32(elems24[4 * i], elems24[4 * i + 1], elems24[4 * i + 2], elems24[4 * i + 3]); };
    el.memory = LookupElements::make(q(0), q(1)); el.instruction = LookupElements::make(q(2), q(3)); el.processor = LookupElements::make(q(4), q(5));
    u32 log = t.log_size();
    std::vector<std::vector<u32>> main_cols;
    for (auto& c : t.cols) main_cols.push_back(broadcast16(c));
    QM31 claimed;
    auto inter = gen_interaction_trace(component, t, el, &claimed);
    std::vector<u32> isf(size_t(1) << log, 0); isf[0] = 1;
    std::vector<const u32*> tc, ic;
    for (auto& c : main_cols) tc.push_back(c.data());
    for (auto& c : inter) ic.push_back(c.data());
    for (size_t row = 0; row < (size_t(1) << log); row++) {
        AssertEvaluator ae;
        ae.is_first_col = isf.data(); ae.trace_cols = tc.data(); ae.inter_cols = ic.data(); ae.row = row; ae.log_size = log; ae.total_sum = claimed;
        eval_component(component, ae, el);
        if (ae.failed >= 0) {
            if (bad_row) *bad_row = row;
            if (bad_constraint) *bad_constraint = ae.failed;
            if (value4) { auto v = ae.failed_value.to_u32(); for (int k = 0; k < 4; k++) value4[k] = v[k]; }
            return 1;
        }
    }
    return 0;
    ORC_CATCH
}

// ---- per-component operations on caller-supplied columns (checkers for the bfhip per-component C ABI) -------------------------
static InteractionElements elements_from(const u32* e) {
    auto q = [&](int i) { return QM31::from_u32(e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]); };
    InteractionElements el;
    el.memory = LookupElements::make(q(0), q(1)); el.instruction = LookupElements::make(q(2), q(3)); el.processor = LookupElements::make(q(4), q(5));
    return el;
}
// gen_interaction_trace of one component. rows: n_main row-granular columns of n_rows (power of two) values, column-major.
// out: 4 * n_logup full-size columns of 2^log_size cells (log_size = log2(n_rows) + 4), column-major. claimed = u32[4].
int orc_logup_generate(int component, const u32* rows, size_t n_rows, const u32* elems24, u32* out, u32 claimed[4]) {
    ORC_TRY
    Table t; t.init(N_MAIN_COLS[component], n_rows);
    for (size_t c = 0; c < t.cols.size(); c++) memcpy(t.cols[c].data(), rows + c * n_rows, n_rows * sizeof(u32));
    QM31 cs;
    auto cols = gen_interaction_trace(component, t, elements_from(elems24), &cs);
    size_t n = size_t(1) << t.log_size();
    for (size_t c = 0; c < cols.size(); c++) memcpy(out + c * n, cols[c].data(), n * sizeof(u32));
    auto a = cs.to_u32(); for (int k = 0; k < 4; k++) claimed[k] = a[k];
    return 0;
    ORC_CATCH
}
// One component's share of compute_composition: acc (4 columns of 2^(log_size+1), column-major) += sum_j coeff[j] * constraint_j * denom_inv.
// is_first, main (n_main columns) and inter (4 * n_logup columns) are full-size LDE columns of 2^(log_size+1) cells, column-major.
int orc_eval_constraints(int component, u32 log_size, const u32* is_first, const u32* main, const u32* inter, const u32* elems24, const u32 claimed[4],
                         const u32* coeffs, u32* acc) {
    ORC_TRY
    u32 eval_log = log_size + 1;
    size_t n = size_t(1) << eval_log;
    InteractionElements el = elements_from(elems24);
    int nc = component_info(component).n_constraints;
    std::vector<QM31> coeff(nc);
    for (int j = 0; j < nc; j++) coeff[j] = QM31::from_u32(coeffs[4 * j], coeffs[4 * j + 1], coeffs[4 * j + 2], coeffs[4 * j + 3]);
    CircleDomain ed = CanonicCoset{eval_log}.circle_domain();
    M31 denom_inv[2];
    for (int i = 0; i < 2; i++) denom_inv[i] = inv(coset_vanishing<M31>(CanonicCoset{log_size}.coset(), ed.at(i)));
    std::vector<const u32*> tc(N_MAIN_COLS[component]), ic(4 * N_LOGUP_COLS[component]);
    for (size_t k = 0; k < tc.size(); k++) tc[k] = main + k * n;
    for (size_t k = 0; k < ic.size(); k++) ic[k] = inter + k * n;
    QM31 total = QM31::from_u32(claimed[0], claimed[1], claimed[2], claimed[3]);
#pragma omp parallel for schedule(static)
    for (size_t row = 0; row < n; row++) {
        DomainEvaluator de;
        de.is_first_col = is_first; de.trace_cols = tc.data(); de.inter_cols = ic.data();
        de.row = row; de.log_size = log_size; de.eval_log = eval_log; de.coeff = coeff.data();
        de.total_sum = total;
        eval_component(component, de, el);
        QM31 v = QM31::from_u32(acc[row], acc[n + row], acc[2 * n + row], acc[3 * n + row]) + de.row_res * denom_inv[row >> log_size];
        auto a = v.to_u32(); for (int k = 0; k < 4; k++) acc[k * n + row] = a[k];
    }
    return 0;
    ORC_CATCH
}
// accumulate_quotients for n_cols full-size columns (2^log_size cells, column-major) with per-column samples listed column by column.
int orc_accumulate_quotients(u32 log_size, const u32* cols, size_t n_cols, const u32* n_samples, const u32* points8, const u32* values4, const u32 random_coeff[4], u32* out) {
    ORC_TRY
    size_t n = size_t(1) << log_size;
    std::vector<std::vector<PointSample>> samples(n_cols);
    size_t si = 0;
    auto q = [](const u32* p) { return QM31::from_u32(p[0], p[1], p[2], p[3]); };
    for (size_t c = 0; c < n_cols; c++)
        for (u32 s = 0; s < n_samples[c]; s++, si++) samples[c].push_back({PointQ(q(points8 + 8 * si), q(points8 + 8 * si + 4)), q(values4 + 4 * si)});
    std::vector<const std::vector<PointSample>*> refs;
    for (auto& v : samples) refs.push_back(&v);
    auto sb = sample_batches(refs);
    auto qc = quotient_constants(sb, q(random_coeff));
    auto pts = domain_points(log_size);
#pragma omp parallel for schedule(static)
    for (size_t row = 0; row < n; row++) {
        std::vector<u32> vals(n_cols);
        for (size_t c = 0; c < n_cols; c++) vals[c] = cols[c * n + row];
        auto a = accumulate_row_quotients(sb, vals.data(), qc, pts[bit_reverse_index((u32)row, log_size)]).to_u32();
        for (int k = 0; k < 4; k++) out[k * n + row] = a[k];
    }
    return 0;
    ORC_CATCH
}

// ---- prove / verify -----------------------------------------------------------------------------------------------------
// Returns a malloc'd JSON string (free with orc_free). transcript_out (optional, malloc'd): "name:hexdigest\n" per tap.
int orc_prove(const char* code, const u8* input, size_t n_in, u32 log_max_rows, char** json_out, size_t* json_len, char** transcript_out, double* seconds) {
    ORC_TRY
    std::vector<u32> ins;
    auto tr = run_program(code, input, n_in, &ins, nullptr);
    Prover pv; pv.log_max_rows = log_max_rows;
    std::string transcript;
    pv.trace_hook = [&](const char* name, const Channel& ch) {
        char buf[80]; transcript += name; transcript += ":";
        for (int i = 0; i < 32; i++) { snprintf(buf, sizeof buf, "%02x", ch.digest.b[i]); transcript += buf; }
        transcript += "\n";
    };
    auto t0 = std::chrono::steady_clock::now();
    BrainfuckProof bp = pv.prove(tr, ins);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    std::string js = proof_to_json(bp);
    *json_out = (char*)malloc(js.size() + 1); memcpy(*json_out, js.c_str(), js.size() + 1); *json_len = js.size();
    if (transcript_out) { *transcript_out = (char*)malloc(transcript.size() + 1); memcpy(*transcript_out, transcript.c_str(), transcript.size() + 1); }
    return 0;
    ORC_CATCH
}
// 0 = verified; 1 = rejected (reason in err); -1 = malformed input
int orc_verify(const char* json, size_t len, u32 log_max_rows, char* err, size_t errcap) {
    ORC_TRY
    BrainfuckProof bp;
    try { bp = proof_from_json(json, len); } catch (const std::exception& e) { if (err) snprintf(err, errcap, "InvalidStructure: %s", e.what()); return 1; }
    Verifier v; v.log_max_rows = log_max_rows;
    std::string e = v.verify(bp);
    if (err) snprintf(err, errcap, "%s", e.c_str());
    return e.empty() ? 0 : 1;
    ORC_CATCH
}

}  // extern "C"
