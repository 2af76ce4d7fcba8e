// The 13 Brainfuck AIRs as host/device templates over an "EvalAtRow"-shaped evaluator — SURVEY.md §8 row a6.
// Mirrors `FrameworkEval::evaluate` of crates/brainfuck_prover/src/components/**/component.rs (cited per function) and the
// logUp bookkeeping stwo's `EvalAtRow::{add_to_relation, finalize_logup}` performs for them (is_first variant: the first
// fraction of a row fetches IsFirst(log_size); the last logUp column carries mask offsets {0, -1}).
// Two evaluators use these bodies: the gfx950 constraint kernels (air.hip, F = M31 at an LDE row) and the host-side
// out-of-domain evaluator (host/prover.h, F = QM31 at the OODS point; stwo's PointEvaluator).
#pragma once
#include "m31.h"

namespace bf {

enum ComponentId { C_MEMORY = 0, C_INSTRUCTION, C_PROGRAM, C_PROCESSOR, C_JNZ, C_JZ, C_INPUT, C_LEFT, C_MINUS, C_OUTPUT, C_PLUS, C_RIGHT, C_EOE, N_COMPONENTS };
constexpr u32 LOG_N_LANES = 4;  // stwo simd::m31::LOG_N_LANES; every table row is broadcast to 16 cells (memory/table.rs:95-104)
enum : u32 { OP_RIGHT = '>', OP_LEFT = '<', OP_PLUS = '+', OP_MINUS = '-', OP_PUTCHAR = '.', OP_READCHAR = ',', OP_JZ = '[', OP_JNZ = ']' };

// (main columns, logUp columns) per component — TraceColumn::count() in each table.rs
BF_HD u32 n_main_cols(int c) { const u32 t[N_COMPONENTS] = {8, 8, 4, 9, 13, 13, 11, 11, 11, 11, 11, 11, 7}; return t[c]; }
BF_HD u32 n_logup_cols(int c) { return c == C_PROCESSOR ? 3 : 1; }
// own constraints + one per logUp column
BF_HD u32 n_constraints(int c) { const u32 t[N_COMPONENTS] = {12, 11, 5, 10, 9, 9, 7, 7, 8, 8, 8, 7, 2}; return t[c]; }
// components whose evaluate() fetches IsFirst itself (a second fetch always happens inside the logUp finalisation)
BF_HD bool uses_is_first(int c) { return c <= C_PROCESSOR; }

// ---- thin operator layer so the AIR bodies read like the reference -------------------------------------------------------
struct Fm { u32 v; };
BF_HD Fm operator+(Fm a, Fm b) { return {m_add(a.v, b.v)}; }
BF_HD Fm operator-(Fm a, Fm b) { return {m_sub(a.v, b.v)}; }
BF_HD Fm operator*(Fm a, Fm b) { return {m_mul(a.v, b.v)}; }
struct Fq { Q31 v; };
BF_HD Fq operator+(Fq a, Fq b) { return {q_add(a.v, b.v)}; }
BF_HD Fq operator-(Fq a, Fq b) { return {q_sub(a.v, b.v)}; }
BF_HD Fq operator*(Fq a, Fq b) { return {q_mul(a.v, b.v)}; }
BF_HD Fq operator*(Fq a, Fm b) { return {q_mulm(a.v, b.v)}; }
BF_HD Fq to_ef(Fm a) { return {q_from_m(a.v)}; }
BF_HD Fq to_ef(Fq a) { return a; }

// LookupElements<N>: z, alpha^0..alpha^6 (memory/table.rs:426-454, instruction/table.rs:391, processor/table.rs:393)
struct Lookup { Q31 z; Q31 alpha_pow[7]; };
struct Lookups { Lookup memory, instruction, processor; };
BF_HD Lookup make_lookup(Q31 z, Q31 alpha) { Lookup l; l.z = z; Q31 cur = q_one(); for (int i = 0; i < 7; i++) { l.alpha_pow[i] = cur; cur = q_mul(cur, alpha); } return l; }

template <class F> BF_HD Fq combine3(const Lookup& l, F a, F b, F c) {
    return Fq{l.alpha_pow[0]} * a + Fq{l.alpha_pow[1]} * b + Fq{l.alpha_pow[2]} * c - Fq{l.z};
}
template <class F> BF_HD Fq combine7(const Lookup& l, F a, F b, F c, F d, F e, F f, F g) {
    return Fq{l.alpha_pow[0]} * a + Fq{l.alpha_pow[1]} * b + Fq{l.alpha_pow[2]} * c + Fq{l.alpha_pow[3]} * d + Fq{l.alpha_pow[4]} * e +
           Fq{l.alpha_pow[5]} * f + Fq{l.alpha_pow[6]} * g - Fq{l.z};
}

// Base-field arguments (the constraint kernels: every LDE row): the sum alpha^i * v_i is a dot product of QM31 constants with M31
// values — four 64-bit accumulators with lazy reduction (m31.h: m_fold / m_canon) instead of a modular multiply-add per term and
// coordinate. Same canonical value as the generic form above (which the out-of-domain evaluator, F = QM31, keeps using).
template <int N> BF_HD Fq combine_base(const Lookup& l, const u32 (&v)[N]) {
    u64 acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (i && i % 3 == 0) { acc[0] = m_fold(acc[0]); acc[1] = m_fold(acc[1]); acc[2] = m_fold(acc[2]); acc[3] = m_fold(acc[3]); }
        const Q31 a = l.alpha_pow[i];
        acc[0] += (u64)a.a.a * v[i]; acc[1] += (u64)a.a.b * v[i]; acc[2] += (u64)a.b.a * v[i]; acc[3] += (u64)a.b.b * v[i];
    }
    return Fq{q_sub(q_make(m_canon(acc[0]), m_canon(acc[1]), m_canon(acc[2]), m_canon(acc[3])), l.z)};
}
BF_HD Fq combine3(const Lookup& l, Fm a, Fm b, Fm c) { const u32 v[3] = {a.v, b.v, c.v}; return combine_base<3>(l, v); }
BF_HD Fq combine7(const Lookup& l, Fm a, Fm b, Fm c, Fm d, Fm e, Fm f, Fm g) { const u32 v[7] = {a.v, b.v, c.v, d.v, e.v, f.v, g.v}; return combine_base<7>(l, v); }

// Evaluator concept E:
//   typename E::F;  is_first() -> F, or a tag T with T * F -> a type constraint() accepts;  F trace();  F cst(u32);
//   void constraint(F) / constraint(Fq);
//   void logup_mid(Fq numerator, Fq denominator) / logup_last(..) — one per add_to_relation, in order; the last closes the row
//                                                (add_to_relation + finalize_logup of the reference)

// memory/component.rs:62-137
template <class E> BF_HD void air_memory(E& e, const Lookups& el) {
    typedef typename E::F F;
    auto is_first = e.is_first();   // F for a per-point evaluator; a tag for the row-group evaluator (the AIRs are linear in IsFirst)
    F clk = e.trace(), mp = e.trace(), mv = e.trace(), d = e.trace(), next_clk = e.trace(), next_mp = e.trace(), next_mv = e.trace(), next_d = e.trace();
    F one = e.cst(1);
    e.constraint(is_first * clk);
    e.constraint(is_first * mp);
    e.constraint(is_first * mv);
    e.constraint(is_first * d);
    e.constraint(d * (d - one));
    e.constraint(next_d * (next_d - one));
    e.constraint((next_mp - mp) * (next_mp - mp - one));
    e.constraint((next_mp - mp - one) * (next_clk - clk - one));
    e.constraint((next_mp - mp) * next_mv);
    e.constraint(d * (next_mp - mp));
    e.constraint(d * (next_mv - mv));
    e.logup_last(to_ef(d - one), combine3(el.memory, clk, mp, mv));
}
// instruction/component.rs:65-142
template <class E> BF_HD void air_instruction(E& e, const Lookups& el) {
    typedef typename E::F F;
    auto is_first = e.is_first();   // F for a per-point evaluator; a tag for the row-group evaluator (the AIRs are linear in IsFirst)
    F ip = e.trace(), ci = e.trace(), ni = e.trace(), d = e.trace(), next_ip = e.trace(), next_ci = e.trace(), next_ni = e.trace(), next_d = e.trace();
    F one = e.cst(1);
    e.constraint(is_first * ip);
    e.constraint(d * (d - one));
    e.constraint(next_d * (next_d - one));
    e.constraint(d * ci);
    e.constraint(d * ni);
    e.constraint(next_d * next_ci);
    e.constraint(next_d * next_ni);
    e.constraint((next_ip - ip) * (next_ip - ip - one));
    e.constraint((next_ip - ip - one) * (next_ci - ci));
    e.constraint((next_ip - ip - one) * (next_ni - ni));
    e.logup_last(to_ef(d - one), combine3(el.instruction, ip, ci, ni));
}
// program/component.rs:60-104
template <class E> BF_HD void air_program(E& e, const Lookups& el) {
    typedef typename E::F F;
    auto is_first = e.is_first();   // F for a per-point evaluator; a tag for the row-group evaluator (the AIRs are linear in IsFirst)
    F ip = e.trace(), ci = e.trace(), ni = e.trace(), d = e.trace();
    F one = e.cst(1);
    e.constraint(is_first * ip);
    e.constraint(d * (d - one));
    e.constraint(d * ci);
    e.constraint(d * ni);
    e.logup_last(to_ef(one - d), combine3(el.instruction, ip, ci, ni));
}
// processor/component.rs:79-153 — three relation entries in the order Processor, Instruction, Memory
template <class E> BF_HD void air_processor(E& e, const Lookups& el) {
    typedef typename E::F F;
    auto is_first = e.is_first();   // F for a per-point evaluator; a tag for the row-group evaluator (the AIRs are linear in IsFirst)
    F clk = e.trace(), ip = e.trace(), ci = e.trace(), ni = e.trace(), mp = e.trace(), mv = e.trace(), mvi = e.trace(), d = e.trace(), next_clk = e.trace();
    F one = e.cst(1);
    e.constraint(is_first * clk);
    e.constraint(is_first * ip);
    e.constraint(is_first * mp);
    e.constraint(is_first * mv);
    e.constraint(mv * (mv * mvi - one));
    e.constraint(mvi * (mv * mvi - one));
    e.constraint(next_clk - clk - one);
    Fq num = Fq{q_one()} - to_ef(d);
    Fq den_p = combine7(el.processor, clk, ip, ci, ni, mp, mv, mvi), den_i = combine3(el.instruction, ip, ci, ni), den_m = combine3(el.memory, clk, mp, mv);
    e.logup_mid(num, den_p);
    e.logup_mid(num, den_i);
    e.logup_last(num, den_m);
}
// jump/jump_if_not_zero_component.rs:61-130, jump/jump_if_zero_component.rs:61-130
template <class E> BF_HD void air_jump(E& e, const Lookups& el, bool if_zero) {
    typedef typename E::F F;
    F clk = e.trace(), ip = e.trace(), ci = e.trace(), ni = e.trace(), mp = e.trace(), mv = e.trace(), mvi = e.trace();
    F next_clk = e.trace(), next_ip = e.trace(), next_mp = e.trace(), next_mv = e.trace(), d = e.trace(), is_mv_zero = e.trace();
    F one = e.cst(1), two = e.cst(2);
    e.constraint(ci * (ci - e.cst(if_zero ? OP_JZ : OP_JNZ)));
    e.constraint(next_clk - clk - one);
    e.constraint(d * (d - one));
    e.constraint(d * mv);
    e.constraint(d * ci);
    if (if_zero) e.constraint((d - one) * (mv * (next_ip - ip - two) + is_mv_zero * (next_ip - (ni + one))));
    else e.constraint((d - one) * (is_mv_zero * (next_ip - ip - two) + mv * (next_ip - ni)));
    e.constraint(next_mp - mp);
    e.constraint(next_mv - mv);
    e.logup_last(to_ef(d - one), combine7(el.processor, clk, ip, ci, ni, mp, mv, mvi));
}
// processor/instructions/{input,left,minus,output,plus,right}_component.rs:62-122
template <class E> BF_HD void air_instr(E& e, const Lookups& el, u32 opcode) {
    typedef typename E::F F;
    F clk = e.trace(), ip = e.trace(), ci = e.trace(), ni = e.trace(), mp = e.trace(), mv = e.trace(), mvi = e.trace(), d = e.trace();
    F next_ip = e.trace(), next_mp = e.trace(), next_mv = e.trace();
    F one = e.cst(1);
    e.constraint(ci * (ci - e.cst(opcode)));
    e.constraint(d * (d - one));
    e.constraint(d * mv);
    e.constraint(d * ci);
    e.constraint((one - d) * (next_ip - ip - one));
    if (opcode == OP_PLUS) { e.constraint(next_mp - mp); e.constraint((one - d) * (next_mv - mv - one)); }
    else if (opcode == OP_MINUS) { e.constraint(next_mp - mp); e.constraint((one - d) * (next_mv - mv + one)); }
    else if (opcode == OP_LEFT) e.constraint((one - d) * (next_mp - mp + one));
    else if (opcode == OP_RIGHT) e.constraint((one - d) * (next_mp - mp - one));
    else if (opcode == OP_READCHAR) e.constraint(next_mp - mp);
    else { e.constraint(next_mp - mp); e.constraint(next_mv - mv); }   // OP_PUTCHAR
    e.logup_last(to_ef(d - one), combine7(el.processor, clk, ip, ci, ni, mp, mv, mvi));
}
// end_of_execution/component.rs:61-90
template <class E> BF_HD void air_eoe(E& e, const Lookups& el) {
    typedef typename E::F F;
    F clk = e.trace(), ip = e.trace(), ci = e.trace(), ni = e.trace(), mp = e.trace(), mv = e.trace(), mvi = e.trace();
    e.constraint(ci);
    e.logup_last(Fq{q_neg(q_one())}, combine7(el.processor, clk, ip, ci, ni, mp, mv, mvi));
}

template <int COMP, class E> BF_HD void air_eval(E& e, const Lookups& el) {
    if (COMP == C_MEMORY) air_memory(e, el);
    else if (COMP == C_INSTRUCTION) air_instruction(e, el);
    else if (COMP == C_PROGRAM) air_program(e, el);
    else if (COMP == C_PROCESSOR) air_processor(e, el);
    else if (COMP == C_JNZ) air_jump(e, el, false);
    else if (COMP == C_JZ) air_jump(e, el, true);
    else if (COMP == C_INPUT) air_instr(e, el, OP_READCHAR);
    else if (COMP == C_LEFT) air_instr(e, el, OP_LEFT);
    else if (COMP == C_MINUS) air_instr(e, el, OP_MINUS);
    else if (COMP == C_OUTPUT) air_instr(e, el, OP_PUTCHAR);
    else if (COMP == C_PLUS) air_instr(e, el, OP_PLUS);
    else if (COMP == C_RIGHT) air_instr(e, el, OP_RIGHT);
    else air_eoe(e, el);
}

// logUp row machinery shared by evaluators (stwo LogupAtRow, finalize_logup without batching): every fraction but the last becomes
// a `(cur - prev_col) * den - num` constraint on a single-offset interaction column; the last one uses the {0,-1} masks and the
// IsFirst-corrected previous row. The first fraction of a row fetches IsFirst(log_size) (second preprocessed mask of the row).
// No arrays, so the evaluator state stays in registers on the GPU.
template <class D, class F_>
struct LogupState {
    Fq prev_col; F_ lu_is_first; Q31 total_sum; bool lu_started = false;
    BF_HD D& self() { return *static_cast<D*>(this); }
    BF_HD void lu_begin() { if (!lu_started) { lu_is_first = self().is_first(); prev_col = Fq{q_zero()}; lu_started = true; } }
    BF_HD void logup_mid(Fq n, Fq d) {
        lu_begin();
        Fq cur = self().inter_cur();
        Fq diff = cur - prev_col;
        prev_col = cur;
        self().constraint(diff * d - n);
    }
    BF_HD void logup_last(Fq n, Fq d) {
        lu_begin();
        Fq cur, prev_row;
        self().inter_cur_prev(cur, prev_row);
        Fq fixed_prev = prev_row - Fq{total_sum} * lu_is_first;
        Fq diff = cur - fixed_prev - prev_col;
        self().constraint(diff * d - n);
    }
};

}  // namespace bf
