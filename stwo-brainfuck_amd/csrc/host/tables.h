// Host side of the drop-in: witness/trace generation for the 13 components (the `XTable::from(&vm_trace)` calls of
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:511-547). Output is ROW-GRANULAR: one u32 per table row and column; the
// 16x lane broadcast the reference performs in `trace_evaluation` (memory/table.rs:95-104) is never materialised — the GPU
// kernels index such columns with `cell >> 4`. SURVEY.md §8(f)1 lists moving the sorts to the GPU as the next widening step.
#pragma once
#include "vm.h"
#include <algorithm>
#include <stdexcept>
#include <thread>
#include <exception>

namespace bf {


static const char* const COMPONENT_NAMES[N_COMPONENTS] = {"memory", "instruction", "program", "processor", "jump_if_not_zero", "jump_if_zero",
    "input_instruction", "left_instruction", "minus_instruction", "output_instruction", "plus_instruction", "right_instruction", "end_of_execution"};

// A component table in row granularity: cols[c][r], r < n_rows (power of two). log_size = log2(n_rows) + LOG_N_LANES.
struct Table {
    std::vector<std::vector<u32>> cols;
    size_t n_rows = 0;
    u32 log_size() const { u32 l = 0; while ((size_t(1) << l) < n_rows) l++; return l + LOG_N_LANES; }
    void init(size_t ncols, size_t rows) { n_rows = rows; cols.assign(ncols, std::vector<u32>(rows, 0)); }
};

static inline size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }  // usize::next_power_of_two (0 -> 1)
static inline u32 madd(u32 a, u32 b) { return m_add(a, b % P31); }

// --- Memory: memory/table.rs:249-303 (sort by (mp, clk), clk-gap fill, pad), :121-151 (pairing) -------------------------------------
// The VM trace is already ordered by clk, so the stable sort by (mp, clk) is a stable bucketing by mp (counting sort); rows are
// written straight into the 8 output columns.
static inline Table memory_table(const std::vector<Registers>& trace) {
    Table t;
    if (trace.empty()) return t;
    bool clk_sorted = true;
    for (size_t i = 1; i < trace.size(); i++) if (trace[i].clk <= trace[i - 1].clk) { clk_sorted = false; break; }
    std::vector<u32> order(trace.size());
    u32 max_mp = 0;
    for (auto& r : trace) max_mp = std::max(max_mp, r.mp);
    if (clk_sorted && max_mp < (1u << 24)) {
        std::vector<u32> cnt(max_mp + 2, 0);
        for (auto& r : trace) cnt[r.mp + 1]++;
        for (size_t m = 1; m < cnt.size(); m++) cnt[m] += cnt[m - 1];
        for (size_t i = 0; i < trace.size(); i++) order[cnt[trace[i].mp]++] = (u32)i;
    } else {
        for (size_t i = 0; i < trace.size(); i++) order[i] = (u32)i;
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return trace[a].mp != trace[b].mp ? trace[a].mp < trace[b].mp : trace[a].clk < trace[b].clk; });
    }
    // size after gap fill
    size_t n = 0;
    for (size_t k = 0; k < order.size(); k++) {
        const Registers& e = trace[order[k]];
        if (k > 0) { const Registers& p = trace[order[k - 1]]; u32 nc = madd(p.clk, 1); if (e.mp == p.mp && e.clk > nc) n += e.clk - nc; }
        n++;
    }
    if (n > (size_t(1) << 28)) throw std::runtime_error("the Memory table would have more than 2^28 rows (2^32 domain rows): not a provable trace");
    size_t rows = next_pow2(n);
    t.init(8, rows);
    u32 *clk = t.cols[0].data(), *mp = t.cols[1].data(), *mv = t.cols[2].data(), *d = t.cols[3].data();
    size_t w = 0;
    auto put = [&](u32 c, u32 p, u32 v, u32 dd) { clk[w] = c; mp[w] = p; mv[w] = v; d[w] = dd; w++; };
    for (size_t k = 0; k < order.size(); k++) {
        const Registers& e = trace[order[k]];
        if (k > 0) {
            const Registers& p = trace[order[k - 1]];
            u32 nc = madd(p.clk, 1);
            if (e.mp == p.mp && e.clk > nc) for (u32 c = nc; c < e.clk; c++) put(c, p.mp, p.mv, 1);
        }
        put(e.clk, e.mp, e.mv, 0);
    }
    u32 lc = clk[w - 1], lp = mp[w - 1], lv = mv[w - 1];
    for (size_t i = 1, n_pad = rows - w; i <= n_pad; i++) put(madd(lc, (u32)i), lp, lv, 1);
    // pairing with the next entry; the last row pairs with one more dummy (last.clk + 1, last.mp, last.mv)
    u32 *nclk = t.cols[4].data(), *nmp = t.cols[5].data(), *nmv = t.cols[6].data(), *nd = t.cols[7].data();
    for (size_t r = 0; r + 1 < rows; r++) { nclk[r] = clk[r + 1]; nmp[r] = mp[r + 1]; nmv[r] = mv[r + 1]; nd[r] = d[r + 1]; }
    nclk[rows - 1] = madd(clk[rows - 1], 1); nmp[rows - 1] = mp[rows - 1]; nmv[rows - 1] = mv[rows - 1]; nd[rows - 1] = 1;
    return t;
}

// --- Instruction: instruction/table.rs:250-284 (program ∪ trace, stable sort by (ip, clk)), :239-248 (pad), :116-145 --
struct InsEntry { u32 ip, ci, ni, d; };
static inline std::vector<Registers> program_registers(const std::vector<u32>& code) {
    std::vector<Registers> p(code.size());
    for (size_t i = 0; i < code.size(); i++) { p[i].ip = (u32)i; p[i].ci = code[i]; p[i].ni = (i + 1 == code.size()) ? 0 : code[i + 1]; }
    return p;
}
static inline Table instruction_table(const std::vector<Registers>& trace, const std::vector<u32>& code) {
    std::vector<Registers> all = program_registers(code);
    all.insert(all.end(), trace.begin(), trace.end());
    std::stable_sort(all.begin(), all.end(), [](const Registers& a, const Registers& b) { return a.ip != b.ip ? a.ip < b.ip : a.clk < b.clk; });
    std::vector<InsEntry> e;
    for (auto& r : all) e.push_back({r.ip, r.ci, r.ni, 0});
    Table t;
    if (e.empty()) return t;
    u32 last_ip = e.back().ip;
    size_t pad = next_pow2(e.size()) - e.size();
    for (size_t i = 0; i < pad; i++) e.push_back({last_ip, 0, 0, 1});
    e.push_back({last_ip, 0, 0, 1});
    t.init(8, e.size() - 1);
    for (size_t r = 0; r + 1 < e.size(); r++) {
        t.cols[0][r] = e[r].ip; t.cols[1][r] = e[r].ci; t.cols[2][r] = e[r].ni; t.cols[3][r] = e[r].d;
        t.cols[4][r] = e[r + 1].ip; t.cols[5][r] = e[r + 1].ci; t.cols[6][r] = e[r + 1].ni; t.cols[7][r] = e[r + 1].d;
    }
    return t;
}

// --- Program: program/table.rs:111-141, pad :62-71 ------------------------------------------------------------------
static inline Table program_table(const std::vector<u32>& code) {
    auto p = program_registers(code);
    Table t;
    if (p.empty()) return t;
    size_t n = next_pow2(p.size());
    t.init(4, n);
    for (size_t r = 0; r < n; r++) {
        if (r < p.size()) { t.cols[0][r] = p[r].ip; t.cols[1][r] = p[r].ci; t.cols[2][r] = p[r].ni; t.cols[3][r] = 0; }
        else { t.cols[0][r] = p.back().ip; t.cols[3][r] = 1; }
    }
    return t;
}

// --- Processor: processor/table.rs:255-265 (entries), :241-253 (pad), :117-145 (pairing) ------------------------------
static inline Table processor_table(const std::vector<Registers>& trace) {
    Table t;
    if (trace.empty()) return t;
    size_t n = next_pow2(trace.size());
    t.init(9, n);
    Registers last = trace.back();
    auto clk_at = [&](size_t r) { return r < trace.size() ? trace[r].clk : madd(last.clk, (u32)(r - trace.size() + 1)); };
    for (size_t r = 0; r < n; r++) {
        if (r < trace.size()) {
            const Registers& g = trace[r];
            t.cols[0][r] = g.clk; t.cols[1][r] = g.ip; t.cols[2][r] = g.ci; t.cols[3][r] = g.ni;
            t.cols[4][r] = g.mp; t.cols[5][r] = g.mv; t.cols[6][r] = g.mvi; t.cols[7][r] = 0;
        } else {
            t.cols[0][r] = clk_at(r); t.cols[1][r] = last.ip; t.cols[7][r] = 1;
        }
        // next entry: r+1 within the padded list, else the extra dummy (last_padded.clk + 1, last_padded.ip)
        t.cols[8][r] = (r + 1 < n) ? clk_at(r + 1) : madd(clk_at(n - 1), 1);
    }
    return t;
}

// --- `< > + - , .` sub-tables: instructions/table.rs:310-328 (selection), :293-307 (pad), :134-161,202-219 (chunks(2)) ---
// --- `[ ]` jump tables: jump/table.rs:280-297, :264-277, :122-146, :191-208 ---------------------------------------------
struct SubEntry { u32 clk, ip, ci, ni, mp, mv, mvi, d; };
static inline std::vector<SubEntry> sub_intermediate(const std::vector<Registers>& trace, u32 opcode) {
    std::vector<SubEntry> e;
    for (size_t k = 0; k + 1 < trace.size(); k++)
        if (trace[k].ci == opcode) {
            const Registers& a = trace[k]; const Registers& b = trace[k + 1];
            e.push_back({a.clk, a.ip, a.ci, a.ni, a.mp, a.mv, a.mvi, 0});
            e.push_back({b.clk, b.ip, b.ci, b.ni, b.mp, b.mv, b.mvi, 0});
        }
    u32 last_clk = e.empty() ? 0 : e.back().clk, last_ip = e.empty() ? 0 : e.back().ip;
    size_t pad = next_pow2(e.size()) - e.size();
    for (size_t i = 0; i < pad; i++) e.push_back({madd(last_clk, (u32)i), last_ip, 0, 0, 0, 0, 0, 1});
    return e;
}
static inline Table instruction_sub_table(const std::vector<Registers>& trace, u32 opcode) {
    auto e = sub_intermediate(trace, opcode);
    Table t;
    size_t n = (e.size() + 1) / 2;
    t.init(11, n);
    for (size_t r = 0; r < n; r++) {
        const SubEntry& a = e[2 * r];
        SubEntry dummy{madd(a.clk, 1), a.ip, 0, 0, 0, 0, 0, 1};
        const SubEntry& b = (2 * r + 1 < e.size()) ? e[2 * r + 1] : dummy;
        t.cols[0][r] = a.clk; t.cols[1][r] = a.ip; t.cols[2][r] = a.ci; t.cols[3][r] = a.ni; t.cols[4][r] = a.mp;
        t.cols[5][r] = a.mv; t.cols[6][r] = a.mvi; t.cols[7][r] = a.d; t.cols[8][r] = b.ip; t.cols[9][r] = b.mp; t.cols[10][r] = b.mv;
    }
    return t;
}
static inline Table jump_table(const std::vector<Registers>& trace, u32 opcode) {
    auto e = sub_intermediate(trace, opcode);
    Table t;
    size_t n = (e.size() + 1) / 2;
    t.init(13, n);
    for (size_t r = 0; r < n; r++) {
        const SubEntry& a = e[2 * r];
        SubEntry dummy{madd(a.clk, 1), a.ip, 0, 0, 0, 0, 0, 1};
        const SubEntry& b = (2 * r + 1 < e.size()) ? e[2 * r + 1] : dummy;
        if (a.d != b.d) throw std::runtime_error("Both entries should be either real or dummy.");  // jump/table.rs:192
        t.cols[0][r] = a.clk; t.cols[1][r] = a.ip; t.cols[2][r] = a.ci; t.cols[3][r] = a.ni; t.cols[4][r] = a.mp;
        t.cols[5][r] = a.mv; t.cols[6][r] = a.mvi; t.cols[7][r] = b.clk; t.cols[8][r] = b.ip; t.cols[9][r] = b.mp; t.cols[10][r] = b.mv;
        t.cols[11][r] = a.d; t.cols[12][r] = m_sub(1, m_mul(a.mv, a.mvi));
    }
    return t;
}

// --- End of execution: end_of_execution/table.rs:100-111, :71-98 (exactly one row; fixed log_size = LOG_N_LANES) ---------
static inline Table eoe_table(const std::vector<Registers>& trace) {
    std::vector<const Registers*> rows;
    for (auto& r : trace) if (r.ci == 0) rows.push_back(&r);
    if (rows.size() != 1) throw std::runtime_error("InvalidEndOfExecution");
    Table t;
    t.init(7, 1);
    const Registers& g = *rows[0];
    u32 v[7] = {g.clk, g.ip, g.ci, g.ni, g.mp, g.mv, g.mvi};
    for (int c = 0; c < 7; c++) t.cols[c][0] = v[c];
    return t;
}

// All 13 tables in claim/commit order (brainfuck_air/mod.rs:550-562).
static inline Table build_one_table(int comp, const std::vector<Registers>& trace, const std::vector<u32>& code) {
    switch (comp) {
        case C_MEMORY: return memory_table(trace);
        case C_INSTRUCTION: return instruction_table(trace, code);
        case C_PROGRAM: return program_table(code);
        case C_PROCESSOR: return processor_table(trace);
        case C_JNZ: return jump_table(trace, OP_JNZ);
        case C_JZ: return jump_table(trace, OP_JZ);
        case C_INPUT: return instruction_sub_table(trace, OP_READCHAR);
        case C_LEFT: return instruction_sub_table(trace, OP_LEFT);
        case C_MINUS: return instruction_sub_table(trace, OP_MINUS);
        case C_OUTPUT: return instruction_sub_table(trace, OP_PUTCHAR);
        case C_PLUS: return instruction_sub_table(trace, OP_PLUS);
        case C_RIGHT: return instruction_sub_table(trace, OP_RIGHT);
        default: return eoe_table(trace);
    }
}
// The 13 builders are independent: one host thread each (the reference builds them one after the other, mod.rs:511-547).
static inline std::vector<Table> build_tables(const std::vector<Registers>& trace, const std::vector<u32>& code) {
    std::vector<Table> t(N_COMPONENTS);
    std::vector<std::exception_ptr> err(N_COMPONENTS);
    std::vector<std::thread> th;
    for (int c = 0; c < N_COMPONENTS; c++)
        th.emplace_back([&, c]() { try { t[c] = build_one_table(c, trace, code); } catch (...) { err[c] = std::current_exception(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
    for (auto& x : t) if (x.n_rows == 0) throw std::runtime_error("EmptyTrace");
    return t;
}

}  // namespace bf
