// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
// The 13 AIRs (FrameworkEval::evaluate) restated from crates/brainfuck_prover/src/components/**/component.rs, generic over
// an "EvalAtRow"-like evaluator E, plus the logUp bookkeeping of stwo@31e8dbc `constraint_framework/mod.rs` (`logup_proxy!`,
// is_first variant, no cumsum shift — PARITY UNPINNED) and LookupElements::combine (memory/table.rs:448-454).
#pragma once
#include "circle.h"
#include "tables.h"

namespace orc {

// stwo constraint_framework::logup::LookupElements<N>; drawn as [z, alpha] = draw_felts(2) (brainfuck_air/mod.rs:158-164).
struct LookupElements {
    QM31 z, alpha;
    QM31 alpha_powers[7];
    static LookupElements make(QM31 z, QM31 alpha) {
        LookupElements l; l.z = z; l.alpha = alpha;
        QM31 cur = QM31::one();
        for (int i = 0; i < 7; i++) { l.alpha_powers[i] = cur; cur = cur * alpha; }
        return l;
    }
    static LookupElements dummy() { LookupElements l; l.z = QM31::one(); l.alpha = QM31::one(); for (auto& p : l.alpha_powers) p = QM31::one(); return l; }
    // combine(values) = sum_i alpha^i * v_i - z
    template <class F>
    QM31 combine(const F* v, size_t n) const {
        QM31 acc = QM31::zero();
        for (size_t i = 0; i < n; i++) acc = acc + alpha_powers[i] * v[i];
        return acc - z;
    }
};
struct InteractionElements { LookupElements memory, instruction, processor; };

static inline QM31 to_ef(M31 x) { return QM31(x); }
static inline QM31 to_ef(QM31 x) { return x; }

// CRTP base carrying the logUp state (stwo LogupAtRow) on top of a concrete evaluator D providing:
//   F is_first_mask(); F next_trace_mask(); EF next_ext_mask0(); void next_ext_mask0m1(EF& cur, EF& prev);
//   void add_constraint(F / EF); F cst(u32);
template <class D, class F_>
struct EvalBase {
    using F = F_;
    using EF = QM31;
    struct Frac { EF num, den; };
    std::vector<Frac> fracs;
    F logup_is_first{};
    QM31 total_sum = QM31::zero();
    D& self() { return *static_cast<D*>(this); }

    // add_to_relation -> write_logup_frac: the first fraction of a row fetches IsFirst(log_size) (second preprocessed mask).
    void add_to_relation(const LookupElements& rel, EF multiplicity, std::initializer_list<F> values) {
        if (fracs.empty()) logup_is_first = self().is_first_mask();
        fracs.push_back({multiplicity, rel.combine(values.begin(), values.size())});
    }
    // finalize_logup (no batching): one interaction column per fraction; the last carries masks {0, -1}.
    void finalize_logup() {
        EF prev_col_cumsum = QM31::zero();
        size_t last = fracs.size() - 1;
        for (size_t k = 0; k < last; k++) {
            EF cur = self().next_ext_mask0();
            EF diff = cur - prev_col_cumsum;
            prev_col_cumsum = cur;
            self().add_constraint(diff * fracs[k].den - fracs[k].num);
        }
        EF cur, prev_row;
        self().next_ext_mask0m1(cur, prev_row);
        EF fixed_prev_row = prev_row - total_sum * logup_is_first;
        EF diff = cur - fixed_prev_row - prev_col_cumsum;
        self().add_constraint(diff * fracs[last].den - fracs[last].num);
        fracs.clear();
    }
};

// ---- the 13 evaluate() bodies (Appendix C of SURVEY.md; citations per function) ---------------------------------------

// memory/component.rs:62-137
template <class E> void eval_memory(E& e, const InteractionElements& el) {
    using F = typename E::F;
    F is_first = e.is_first_mask();
    F clk = e.next_trace_mask(), mp = e.next_trace_mask(), mv = e.next_trace_mask(), d = e.next_trace_mask();
    F next_clk = e.next_trace_mask(), next_mp = e.next_trace_mask(), next_mv = e.next_trace_mask(), next_d = e.next_trace_mask();
    F one = e.cst(1);
    e.add_constraint(is_first * clk);
    e.add_constraint(is_first * mp);
    e.add_constraint(is_first * mv);
    e.add_constraint(is_first * d);
    e.add_constraint(d * (d - one));
    e.add_constraint(next_d * (next_d - one));
    e.add_constraint((next_mp - mp) * (next_mp - mp - one));
    e.add_constraint((next_mp - mp - one) * (next_clk - clk - one));
    e.add_constraint((next_mp - mp) * next_mv);
    e.add_constraint(d * (next_mp - mp));
    e.add_constraint(d * (next_mv - mv));
    e.add_to_relation(el.memory, to_ef(d - one), {clk, mp, mv});
    e.finalize_logup();
}
// instruction/component.rs:65-142
template <class E> void eval_instruction(E& e, const InteractionElements& el) {
    using F = typename E::F;
    F is_first = e.is_first_mask();
    F ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), d = e.next_trace_mask();
    F next_ip = e.next_trace_mask(), next_ci = e.next_trace_mask(), next_ni = e.next_trace_mask(), next_d = e.next_trace_mask();
    F one = e.cst(1);
    e.add_constraint(is_first * ip);
    e.add_constraint(d * (d - one));
    e.add_constraint(next_d * (next_d - one));
    e.add_constraint(d * ci);
    e.add_constraint(d * ni);
    e.add_constraint(next_d * next_ci);
    e.add_constraint(next_d * next_ni);
    e.add_constraint((next_ip - ip) * (next_ip - ip - one));
    e.add_constraint((next_ip - ip - one) * (next_ci - ci));
    e.add_constraint((next_ip - ip - one) * (next_ni - ni));
    e.add_to_relation(el.instruction, to_ef(d - one), {ip, ci, ni});
    e.finalize_logup();
}
// program/component.rs:60-104
template <class E> void eval_program(E& e, const InteractionElements& el) {
    using F = typename E::F;
    F is_first = e.is_first_mask();
    F ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), d = e.next_trace_mask();
    F one = e.cst(1);
    e.add_constraint(is_first * ip);
    e.add_constraint(d * (d - one));
    e.add_constraint(d * ci);
    e.add_constraint(d * ni);
    e.add_to_relation(el.instruction, to_ef(one - d), {ip, ci, ni});
    e.finalize_logup();
}
// processor/component.rs:79-153
template <class E> void eval_processor(E& e, const InteractionElements& el) {
    using F = typename E::F;
    F is_first = e.is_first_mask();
    F clk = e.next_trace_mask(), ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), mp = e.next_trace_mask();
    F mv = e.next_trace_mask(), mvi = e.next_trace_mask(), d = e.next_trace_mask(), next_clk = e.next_trace_mask();
    F one = e.cst(1);
    e.add_constraint(is_first * clk);
    e.add_constraint(is_first * ip);
    e.add_constraint(is_first * mp);
    e.add_constraint(is_first * mv);
    e.add_constraint(mv * (mv * mvi - one));
    e.add_constraint(mvi * (mv * mvi - one));
    e.add_constraint(next_clk - clk - one);
    QM31 num = QM31::one() - to_ef(d);
    e.add_to_relation(el.processor, num, {clk, ip, ci, ni, mp, mv, mvi});
    e.add_to_relation(el.instruction, num, {ip, ci, ni});
    e.add_to_relation(el.memory, num, {clk, mp, mv});
    e.finalize_logup();
}
// jump/jump_if_not_zero_component.rs:61-130 and jump/jump_if_zero_component.rs:61-130
template <class E> void eval_jump(E& e, const InteractionElements& el, bool if_zero) {
    using F = typename E::F;
    F clk = e.next_trace_mask(), ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), mp = e.next_trace_mask();
    F mv = e.next_trace_mask(), mvi = e.next_trace_mask(), next_clk = e.next_trace_mask(), next_ip = e.next_trace_mask();
    F next_mp = e.next_trace_mask(), next_mv = e.next_trace_mask(), d = e.next_trace_mask(), is_mv_zero = e.next_trace_mask();
    F one = e.cst(1), two = e.cst(2);
    e.add_constraint(ci * (ci - e.cst(if_zero ? OP_JZ : OP_JNZ)));
    e.add_constraint(next_clk - clk - one);
    e.add_constraint(d * (d - one));
    e.add_constraint(d * mv);
    e.add_constraint(d * ci);
    if (if_zero) e.add_constraint((d - one) * (mv * (next_ip - ip - two) + is_mv_zero * (next_ip - (ni + one))));
    else e.add_constraint((d - one) * (is_mv_zero * (next_ip - ip - two) + mv * (next_ip - ni)));
    e.add_constraint(next_mp - mp);
    e.add_constraint(next_mv - mv);
    e.add_to_relation(el.processor, to_ef(d - one), {clk, ip, ci, ni, mp, mv, mvi});
    e.finalize_logup();
}
// processor/instructions/{input,left,minus,output,plus,right}_component.rs:62-122
template <class E> void eval_instr_sub(E& e, const InteractionElements& el, u32 opcode) {
    using F = typename E::F;
    F clk = e.next_trace_mask(), ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), mp = e.next_trace_mask();
    F mv = e.next_trace_mask(), mvi = e.next_trace_mask(), d = e.next_trace_mask(), next_ip = e.next_trace_mask();
    F next_mp = e.next_trace_mask(), next_mv = e.next_trace_mask();
    F one = e.cst(1);
    e.add_constraint(ci * (ci - e.cst(opcode)));
    e.add_constraint(d * (d - one));
    e.add_constraint(d * mv);
    e.add_constraint(d * ci);
    e.add_constraint((one - d) * (next_ip - ip - one));
    switch (opcode) {
        case OP_PLUS: e.add_constraint(next_mp - mp); e.add_constraint((one - d) * (next_mv - mv - one)); break;
        case OP_MINUS: e.add_constraint(next_mp - mp); e.add_constraint((one - d) * (next_mv - mv + one)); break;
        case OP_LEFT: e.add_constraint((one - d) * (next_mp - mp + one)); break;
        case OP_RIGHT: e.add_constraint((one - d) * (next_mp - mp - one)); break;
        case OP_READCHAR: e.add_constraint(next_mp - mp); break;
        case OP_PUTCHAR: e.add_constraint(next_mp - mp); e.add_constraint(next_mv - mv); break;
    }
    e.add_to_relation(el.processor, to_ef(d - one), {clk, ip, ci, ni, mp, mv, mvi});
    e.finalize_logup();
}
// end_of_execution/component.rs:61-90
template <class E> void eval_eoe(E& e, const InteractionElements& el) {
    using F = typename E::F;
    F clk = e.next_trace_mask(), ip = e.next_trace_mask(), ci = e.next_trace_mask(), ni = e.next_trace_mask(), mp = e.next_trace_mask();
    F mv = e.next_trace_mask(), mvi = e.next_trace_mask();
    e.add_constraint(ci);
    e.add_to_relation(el.processor, -QM31::one(), {clk, ip, ci, ni, mp, mv, mvi});
    e.finalize_logup();
}

template <class E> void eval_component(int comp, E& e, const InteractionElements& el) {
    switch (comp) {
        case C_MEMORY: eval_memory(e, el); break;
        case C_INSTRUCTION: eval_instruction(e, el); break;
        case C_PROGRAM: eval_program(e, el); break;
        case C_PROCESSOR: eval_processor(e, el); break;
        case C_JNZ: eval_jump(e, el, false); break;
        case C_JZ: eval_jump(e, el, true); break;
        case C_INPUT: eval_instr_sub(e, el, OP_READCHAR); break;
        case C_LEFT: eval_instr_sub(e, el, OP_LEFT); break;
        case C_MINUS: eval_instr_sub(e, el, OP_MINUS); break;
        case C_OUTPUT: eval_instr_sub(e, el, OP_PUTCHAR); break;
        case C_PLUS: eval_instr_sub(e, el, OP_PLUS); break;
        case C_RIGHT: eval_instr_sub(e, el, OP_RIGHT); break;
        case C_EOE: eval_eoe(e, el); break;
    }
}

// InfoEvaluator: counts constraints / masks (stwo constraint_framework/info.rs).
struct InfoEvaluator : EvalBase<InfoEvaluator, M31> {
    int n_constraints = 0, n_preprocessed = 0, n_trace = 0, n_interaction = 0;
    M31 is_first_mask() { n_preprocessed++; return M31(0); }
    M31 next_trace_mask() { n_trace++; return M31(0); }
    QM31 next_ext_mask0() { n_interaction += 4; return QM31::zero(); }
    void next_ext_mask0m1(QM31& c, QM31& p) { n_interaction += 4; c = p = QM31::zero(); }
    template <class G> void add_constraint(G) { n_constraints++; }
    M31 cst(u32 k) { return M31(k); }
};
static inline InfoEvaluator component_info(int comp) {
    InfoEvaluator ie;
    InteractionElements el{LookupElements::dummy(), LookupElements::dummy(), LookupElements::dummy()};
    eval_component(comp, ie, el);
    return ie;
}

// SimdDomainEvaluator equivalent: evaluates one row of the LDE domain (log = log_size + 1), bit-reversed storage.
struct DomainEvaluator : EvalBase<DomainEvaluator, M31> {
    const u32* is_first_col;              // IsFirst(log_size) LDE column
    const u32* const* trace_cols;         // main LDE columns of the component
    const u32* const* inter_cols;         // interaction LDE columns (4 per logUp column)
    size_t row; u32 log_size, eval_log;
    const QM31* coeff; int ci = 0;        // random_coeff_powers (already reversed): constraint j uses coeff[j]
    int ti = 0, ii = 0;
    QM31 row_res = QM31::zero();
    M31 is_first_mask() { return M31(is_first_col[row]); }
    M31 next_trace_mask() { return M31(trace_cols[ti++][row]); }
    QM31 next_ext_mask0() { QM31 v = QM31::from_u32(inter_cols[ii][row], inter_cols[ii + 1][row], inter_cols[ii + 2][row], inter_cols[ii + 3][row]); ii += 4; return v; }
    void next_ext_mask0m1(QM31& c, QM31& p) {
        size_t pr = offset_bit_reversed_circle_domain_index(row, log_size, eval_log, -1);
        c = QM31::from_u32(inter_cols[ii][row], inter_cols[ii + 1][row], inter_cols[ii + 2][row], inter_cols[ii + 3][row]);
        p = QM31::from_u32(inter_cols[ii][pr], inter_cols[ii + 1][pr], inter_cols[ii + 2][pr], inter_cols[ii + 3][pr]);
        ii += 4;
    }
    template <class G> void add_constraint(G c) { row_res = row_res + coeff[ci++] * c; }
    M31 cst(u32 k) { return M31(k); }
};

// PointEvaluator: evaluates the constraints at the OODS point from the sampled mask values (Horner in random_coeff).
struct PointEvaluator : EvalBase<PointEvaluator, QM31> {
    const QM31* preproc; int pi = 0;               // one value per is_first fetch
    const std::vector<QM31>* trace_vals; int ti = 0;     // per main column: [value at offset 0]
    const std::vector<QM31>* inter_vals; int ii = 0;     // per interaction column: [off 0] or [off 0, off -1]
    QM31 denom_inverse; QM31 random_coeff; QM31* accumulation;
    QM31 is_first_mask() { return preproc[pi++]; }
    QM31 next_trace_mask() { return trace_vals[ti++][0]; }
    static QM31 combine_ef(QM31 a, QM31 b, QM31 c, QM31 d) { QM31 e[4] = {a, b, c, d}; return from_partial_evals(e); }
    QM31 next_ext_mask0() { QM31 v = combine_ef(inter_vals[ii][0], inter_vals[ii + 1][0], inter_vals[ii + 2][0], inter_vals[ii + 3][0]); ii += 4; return v; }
    void next_ext_mask0m1(QM31& c, QM31& p) {
        const int ic = conventions().logup_mask_order == 1 ? 1 : 0, ip = 1 - ic;   // position of offset 0 / offset -1 in the sampled list
        c = combine_ef(inter_vals[ii][ic], inter_vals[ii + 1][ic], inter_vals[ii + 2][ic], inter_vals[ii + 3][ic]);
        p = combine_ef(inter_vals[ii][ip], inter_vals[ii + 1][ip], inter_vals[ii + 2][ip], inter_vals[ii + 3][ip]);
        ii += 4;
    }
    template <class G> void add_constraint(G c) { *accumulation = *accumulation * random_coeff + denom_inverse * c; }
    QM31 cst(u32 k) { return QM31(M31(k)); }
};

// AssertEvaluator: stwo `assert_constraints` analogue on the TRACE domain (used by tests to check a trace satisfies its AIR,
// mirroring memory/component.rs:163-209 and plus_component.rs:145-190).
struct AssertEvaluator : EvalBase<AssertEvaluator, M31> {
    const u32* is_first_col; const u32* const* trace_cols; const u32* const* inter_cols;
    size_t row; u32 log_size; int ti = 0, ii = 0, ci = 0; int failed = -1; QM31 failed_value = QM31::zero();
    M31 is_first_mask() { return M31(is_first_col[row]); }
    M31 next_trace_mask() { return M31(trace_cols[ti++][row]); }
    QM31 rd(size_t r) { return QM31::from_u32(inter_cols[ii][r], inter_cols[ii + 1][r], inter_cols[ii + 2][r], inter_cols[ii + 3][r]); }
    QM31 next_ext_mask0() { QM31 v = rd(row); ii += 4; return v; }
    void next_ext_mask0m1(QM31& c, QM31& p) {
        // previous row in coset order on the trace domain itself (eval_log == log_size: step 2^-1 is not integral, so walk coset order)
        size_t d = bit_reverse_index((u32)row, log_size);
        size_t cidx = circle_domain_index_to_coset_index(d, log_size);
        size_t n = size_t(1) << log_size;
        size_t pc = (cidx + n - 1) % n;
        size_t pr = bit_reverse_index((u32)coset_index_to_circle_domain_index(pc, log_size), log_size);
        c = rd(row); p = rd(pr); ii += 4;
    }
    void add_constraint(M31 c) { if (!c.is_zero() && failed < 0) { failed = ci; failed_value = QM31(c); } ci++; }
    void add_constraint(QM31 c) { if (!c.is_zero() && failed < 0) { failed = ci; failed_value = c; } ci++; }
    M31 cst(u32 k) { return M31(k); }
};

}  // namespace orc
