// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). PARITY UNPINNED at the stwo boundary.
// CPU restatement of the whole hot path: prove_brainfuck (crates/brainfuck_prover/src/brainfuck_air/mod.rs:471-735) and
// verify_brainfuck (mod.rs:738-797) together with the stwo@31e8dbc pieces they call: TreeBuilder/CommitmentSchemeProver
// (core/pcs/prover.rs), LogupTraceGenerator (constraint_framework/logup.rs), DomainEvaluationAccumulator (core/air/accumulation.rs),
// FrameworkComponent (constraint_framework/component.rs), compute_fri_quotients (core/pcs/quotients.rs, backend/cpu/quotients.rs),
// FriProver/FriVerifier (core/fri.rs), Queries (core/queries.rs), prover::prove/verify (core/prover/mod.rs).
#pragma once
#include "air.h"
#include "merkle.h"
#include <map>
#include <set>
#include <string>
#include <sstream>
#include <stdexcept>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace orc {

struct PcsConfig { u32 pow_bits = 5, log_blowup = 1, log_last_layer_degree_bound = 0, n_queries = 3; };  // PcsConfig::default()

struct FriLayerProof { std::vector<QM31> fri_witness; MerkleDecommitment decommitment; Hash32 commitment; };
struct FriProof { FriLayerProof first_layer; std::vector<FriLayerProof> inner_layers; std::vector<QM31> last_layer_coeffs; u32 last_layer_log_size = 0; };
struct StarkProof {
    std::vector<Hash32> commitments;
    std::vector<std::vector<std::vector<QM31>>> sampled_values;  // tree -> column -> point
    std::vector<MerkleDecommitment> decommitments;
    std::vector<std::vector<u32>> queried_values;
    u64 proof_of_work = 0;
    FriProof fri_proof;
};
// BrainfuckProof { claim, interaction_claim, proof } (mod.rs:71-76)
struct BrainfuckProof { u32 log_sizes[N_COMPONENTS]; QM31 claimed_sums[N_COMPONENTS]; StarkProof proof; };

// ------------------------------------------------------------------------------------------------------------------
struct PolyCol { u32 log_size; std::vector<u32> coeffs; };
struct EvalCol { u32 log_size; std::vector<u32> values; };

static inline PolyCol interpolate_col(const std::vector<u32>& values, u32 log_size, const TwiddleTree& tw) {
    PolyCol p{log_size, values};
    circle_interpolate(reinterpret_cast<M31*>(p.coeffs.data()), log_size, tw);
    return p;
}
static inline EvalCol evaluate_poly(const PolyCol& p, u32 eval_log, const TwiddleTree& tw) {
    EvalCol e{eval_log, p.coeffs};
    e.values.resize(size_t(1) << eval_log, 0);  // CirclePoly::extend = zero padding
    circle_evaluate(reinterpret_cast<M31*>(e.values.data()), eval_log, tw);
    return e;
}

struct CommitmentTree {
    std::vector<PolyCol> polys;
    std::vector<EvalCol> evals;
    MerkleProver merkle;
    std::vector<ColRef> col_refs() const { std::vector<ColRef> r; for (auto& e : evals) r.push_back({e.values.data(), e.log_size}); return r; }
};

// CommitmentSchemeProver::commit -> CommitmentTreeProver::new: LDE (blowup) + Merkle + mix_root
static inline void commit_tree(CommitmentTree& t, const PcsConfig& cfg, const TwiddleTree& tw, Channel& ch) {
    t.evals.resize(t.polys.size());
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < t.polys.size(); i++) t.evals[i] = evaluate_poly(t.polys[i], t.polys[i].log_size + cfg.log_blowup, tw);
    t.merkle = MerkleProver::commit(t.col_refs());
    ch.mix_root(t.merkle.root());
}

// ---- logUp interaction trace (a5): LogupTraceGenerator::{new_col, write_frac, finalize_col, finalize_last} -----------
struct LogupColSpec { int rel; int n; int cols[7]; };
struct LogupSpec { int n_cols; LogupColSpec c[3]; int d_col; int mode; };  // mode: 0 -> d-1, 1 -> 1-d, 2 -> -1
static inline LogupSpec logup_spec(int comp) {
    const LogupColSpec mem3{0, 3, {0, 1, 2}}, ins3{1, 3, {0, 1, 2}}, proc7{2, 7, {0, 1, 2, 3, 4, 5, 6}};
    switch (comp) {
        case C_MEMORY: return {1, {mem3}, 3, 0};                                   // memory/table.rs:485-518
        case C_INSTRUCTION: return {1, {ins3}, 3, 0};                              // instruction/table.rs:456-490
        case C_PROGRAM: return {1, {ins3}, 3, 1};                                  // program/table.rs:233-265
        case C_PROCESSOR: return {3, {proc7, {1, 3, {1, 2, 3}}, {0, 3, {0, 4, 5}}}, 7, 1};  // processor/table.rs:456-529
        case C_JNZ: case C_JZ: return {1, {proc7}, 11, 0};                         // jump/table.rs:436-475
        case C_EOE: return {1, {proc7}, -1, 2};                                    // end_of_execution/table.rs:220-255
        default: return {1, {proc7}, 7, 0};                                        // instructions/table.rs:466-505
    }
}
static inline const LookupElements& rel_of(const InteractionElements& el, int rel) { return rel == 0 ? el.memory : rel == 1 ? el.instruction : el.processor; }

// Returns 4*n_cols M31 columns of 2^log_size cells (bit-reversed circle-domain order) and the claimed sum.
static inline std::vector<std::vector<u32>> gen_interaction_trace(int comp, const Table& t, const InteractionElements& el, QM31* claimed_sum) {
    LogupSpec sp = logup_spec(comp);
    u32 log_size = t.log_size();
    size_t n = size_t(1) << log_size, rows = t.n_rows;
    std::vector<std::vector<u32>> out(4 * sp.n_cols, std::vector<u32>(n));
    std::vector<QM31> prev(rows, QM31::zero()), den(rows), deninv(rows), cur(rows);
    for (int k = 0; k < sp.n_cols; k++) {
        const LookupElements& le = rel_of(el, sp.c[k].rel);
        for (size_t r = 0; r < rows; r++) {
            M31 v[7];
            for (int j = 0; j < sp.c[k].n; j++) v[j] = M31(t.cols[sp.c[k].cols[j]][r]);
            den[r] = le.combine(v, sp.c[k].n);
        }
        batch_inverse(den.data(), deninv.data(), rows);
        for (size_t r = 0; r < rows; r++) {
            QM31 num;
            if (sp.mode == 2) num = -QM31::one();
            else { QM31 d = QM31(M31(t.cols[sp.d_col][r])); num = sp.mode == 0 ? d - QM31::one() : QM31::one() - d; }
            cur[r] = num * deninv[r] + prev[r];
        }
        if (k + 1 < sp.n_cols) {
            for (size_t r = 0; r < rows; r++) { auto a = cur[r].to_u32(); for (int c = 0; c < 4; c++) for (size_t l = 0; l < 16; l++) out[4 * k + c][(r << 4) + l] = a[c]; }
            prev = cur;
        } else {
            // finalize_last: inclusive prefix sum in coset order over bit-reversed circle-domain storage.
            QM31 acc = QM31::zero();
            for (size_t c = 0; c < n; c++) {
                size_t s = bit_reverse_index((u32)coset_index_to_circle_domain_index(c, log_size), log_size);
                acc = acc + cur[s >> 4];
                auto a = acc.to_u32();
                for (int q = 0; q < 4; q++) out[4 * k + q][s] = a[q];
            }
            *claimed_sum = acc;  // == prefix_sum.at(1): the last coset element is stored at index 1
        }
    }
    return out;
}

// ---- component bookkeeping ---------------------------------------------------------------------------------------------
struct ComponentLayout {
    u32 log_size;
    int n_constraints, n_preproc;   // number of IsFirst fetches (1 or 2)
    size_t main_off, inter_off;     // offsets into tree 1 / tree 2 columns
    u32 n_main, n_inter;
};
static inline std::vector<ComponentLayout> component_layouts(const u32* log_sizes) {
    std::vector<ComponentLayout> L(N_COMPONENTS);
    size_t mo = 0, io = 0;
    for (int c = 0; c < N_COMPONENTS; c++) {
        InfoEvaluator ie = component_info(c);
        L[c] = {log_sizes[c], ie.n_constraints, ie.n_preprocessed, mo, io, N_MAIN_COLS[c], 4 * N_LOGUP_COLS[c]};
        mo += N_MAIN_COLS[c]; io += 4 * N_LOGUP_COLS[c];
    }
    return L;
}

// Components::mask_points + the composition mask (prover::prove). tree -> column -> points.
static inline std::vector<std::vector<std::vector<PointQ>>> mask_points(const std::vector<ComponentLayout>& L, u32 log_max_rows, PointQ p) {
    std::vector<std::vector<std::vector<PointQ>>> mp(4);
    mp[0].assign(log_max_rows - LOG_N_LANES + 1, {});
    for (auto& l : L) mp[0][log_max_rows - l.log_size] = {p};   // IS_FIRST_LOG_SIZES[i] = LOG_MAX_ROWS - i (mod.rs:453-464)
    for (auto& l : L) {
        for (u32 c = 0; c < l.n_main; c++) mp[1].push_back({p});
        PointQ prev = p - into_ef(CanonicCoset{l.log_size}.step());    // point + trace_step.mul_signed(-1)
        for (u32 c = 0; c < l.n_inter; c++) {
            if (c + 4 >= l.n_inter) { if (conventions().logup_mask_order == 1) mp[2].push_back({prev, p}); else mp[2].push_back({p, prev}); }
            else mp[2].push_back({p});
        }
    }
    mp[3].assign(4, {p});
    return mp;
}

// Components::eval_composition_polynomial_at_point
static inline QM31 eval_composition_at_point(const std::vector<ComponentLayout>& L, u32 log_max_rows, const InteractionElements& el,
                                             const QM31* claimed_sums, PointQ p, const std::vector<std::vector<std::vector<QM31>>>& sv, QM31 random_coeff) {
    QM31 acc = QM31::zero();
    for (int c = 0; c < N_COMPONENTS; c++) {
        const auto& l = L[c];
        PointEvaluator pe;
        const auto& pv = sv[0][log_max_rows - l.log_size];
        if (pv.size() != 1) throw std::runtime_error("InvalidStructure");
        QM31 pre[2] = {pv[0], pv[0]};
        pe.preproc = pre;
        pe.trace_vals = &sv[1][l.main_off];
        pe.inter_vals = &sv[2][l.inter_off];
        pe.denom_inverse = inv(coset_vanishing<QM31>(CanonicCoset{l.log_size}.coset(), p));
        pe.random_coeff = random_coeff;
        pe.accumulation = &acc;
        pe.total_sum = claimed_sums[c];
        eval_component(c, pe, el);
    }
    return acc;
}

// ---- quotients (a9) -----------------------------------------------------------------------------------------------------
struct PointCmp {
    bool operator()(const PointQ& a, const PointQ& b) const {
        auto ax = a.x.to_u32(), bx = b.x.to_u32(), ay = a.y.to_u32(), by = b.y.to_u32();
        for (int i = 0; i < 4; i++) if (ax[i] != bx[i]) return ax[i] < bx[i];
        for (int i = 0; i < 4; i++) if (ay[i] != by[i]) return ay[i] < by[i];
        return false;
    }
};
struct SampleBatch { PointQ point; std::vector<std::pair<size_t, QM31>> cols; };
struct PointSample { PointQ point; QM31 value; };
// ColumnSampleBatch::new_vec: group by point through a BTreeMap (ordered by derived Ord of CirclePoint<QM31>).
static inline std::vector<SampleBatch> sample_batches(const std::vector<const std::vector<PointSample>*>& samples) {
    std::map<PointQ, std::vector<std::pair<size_t, QM31>>, PointCmp> g;
    for (size_t ci = 0; ci < samples.size(); ci++) for (auto& s : *samples[ci]) g[s.point].push_back({ci, s.value});
    std::vector<SampleBatch> out;
    for (auto& kv : g) out.push_back({kv.first, kv.second});
    return out;
}
struct QuotientConstants { std::vector<std::vector<std::array<QM31, 3>>> line_coeffs; std::vector<QM31> batch_random_coeffs; };
static inline QuotientConstants quotient_constants(const std::vector<SampleBatch>& sb, QM31 random_coeff) {
    QuotientConstants qc;
    for (auto& b : sb) {
        QM31 alpha = QM31::one();
        std::vector<std::array<QM31, 3>> lc;
        for (auto& cv : b.cols) {
            alpha = alpha * random_coeff;
            // complex_conjugate_line_coeffs
            QM31 a = cv.second.conj() - cv.second;
            QM31 c = b.point.y.conj() - b.point.y;
            QM31 bb = cv.second * c - a * b.point.y;
            lc.push_back({alpha * a, alpha * bb, alpha * c});
        }
        qc.line_coeffs.push_back(lc);
        qc.batch_random_coeffs.push_back(qm31_pow(random_coeff, b.cols.size()));
    }
    return qc;
}
// accumulate_row_quotients
// denominator of one sample batch at a domain point: (Pr.x - D.x) * Pi.y - (Pr.y - D.y) * Pi.x  in CM31 (denominator_inverses)
static inline CM31 quotient_denominator(const SampleBatch& b, PointM dp) {
    CM31 prx = b.point.x.a, pry = b.point.y.a, pix = b.point.x.b, piy = b.point.y.b;
    return (prx - dp.x) * piy - (pry - dp.y) * pix;
}
// deninvs: optional precomputed inverses (one per batch) — values identical to inverting in place.
static inline QM31 accumulate_row_quotients(const std::vector<SampleBatch>& sb, const u32* vals, const QuotientConstants& qc, PointM dp, const CM31* deninvs = nullptr) {
    QM31 row = QM31::zero();
    for (size_t bi = 0; bi < sb.size(); bi++) {
        const auto& b = sb[bi];
        CM31 deninv = deninvs ? deninvs[bi] : inv(quotient_denominator(b, dp));
        QM31 num = QM31::zero();
        for (size_t k = 0; k < b.cols.size(); k++) {
            const auto& lc = qc.line_coeffs[bi][k];
            QM31 value = lc[2] * M31(vals[b.cols[k].first]);
            QM31 linear = lc[0] * dp.y + lc[1];
            num = num + (value - linear);
        }
        row = row * qc.batch_random_coeffs[bi] + mul_cm31(num, deninv);
    }
    return row;
}

// Points of CanonicCoset(log).circle_domain() in natural order.
static inline std::vector<PointM> domain_points(u32 log) {
    auto half = CanonicCoset{log}.half_coset().points();
    std::vector<PointM> pts(half.size() * 2);
#pragma omp parallel for schedule(static) if (simd::enabled() && half.size() >= 4096)
    for (size_t i = 0; i < half.size(); i++) { pts[i] = half[i]; pts[half.size() + i] = -half[i]; }
    return pts;
}

struct SecureCol { u32 log_size; std::vector<u32> c[4]; QM31 at(size_t i) const { return QM31::from_u32(c[0][i], c[1][i], c[2][i], c[3][i]); }
    void set(size_t i, QM31 v) { auto a = v.to_u32(); for (int k = 0; k < 4; k++) c[k][i] = a[k]; }
    void init(u32 log) { log_size = log; for (auto& x : c) x.assign(size_t(1) << log, 0); } };

// ---- FRI (a10) ------------------------------------------------------------------------------------------------------------
// fold_circle_into_line (backend/cpu/fri.rs): dst[i] = dst[i]*alpha^2 + (f0 + alpha*f1), (f0,f1)=ibutterfly(f(p), f(-p), 1/p.y)
static inline void fold_circle_into_line(std::vector<QM31>& dst, const SecureCol& src, QM31 alpha) {
    u32 log = src.log_size;
    auto half = CanonicCoset{log}.half_coset().points();
    QM31 alpha_sq = alpha * alpha;
    std::vector<M31> ys(half.size()), yinv(half.size());
    for (size_t i = 0; i < half.size(); i++) ys[i] = half[i].y;
    batch_inverse(ys.data(), yinv.data(), ys.size());
#pragma omp parallel for schedule(static) if (dst.size() >= 4096)
    for (size_t i = 0; i < dst.size(); i++) {
        size_t pi = bit_reverse_index((u32)i, log - 1);  // domain.at(bit_reverse(i << 1, log)) = half_coset.at(bit_reverse(i, log-1))
        QM31 f0 = src.at(2 * i), f1 = src.at(2 * i + 1);
        ibutterfly(f0, f1, yinv[pi]);
        dst[i] = dst[i] * alpha_sq + (alpha * f1 + f0);
    }
}
// fold_line: line domain = LineDomain(Coset::half_odds(log)); x = domain.at(bit_reverse(i << 1, log))
static inline std::vector<QM31> fold_line(const std::vector<QM31>& eval, u32 log, QM31 alpha) {
    auto pts = Coset::half_odds(log).points();
    std::vector<QM31> out(eval.size() / 2);
    std::vector<M31> xs(pts.size()), xinv(pts.size());
    for (size_t i = 0; i < pts.size(); i++) xs[i] = pts[i].x;
    batch_inverse(xs.data(), xinv.data(), xs.size());
#pragma omp parallel for schedule(static) if (out.size() >= 4096)
    for (size_t i = 0; i < out.size(); i++) {
        QM31 f0 = eval[2 * i], f1 = eval[2 * i + 1];
        ibutterfly(f0, f1, xinv[bit_reverse_index((u32)(i << 1), log)]);
        out[i] = f0 + alpha * f1;
    }
    return out;
}

// Queries::generate / fold
static inline std::vector<size_t> generate_queries(Channel& ch, u32 log_domain_size, u32 n_queries) {
    std::set<size_t> q;
    u32 cnt = 0; u32 mask = (u32)((u64(1) << log_domain_size) - 1);
    for (;;) {
        std::vector<u8> r = ch.draw_random_bytes();          // chunks_exact(4): 8 words per draw (Blake2s, 32 bytes), 7 (Poseidon252, 31 bytes)
        for (size_t k = 0; 4 * k + 4 <= r.size(); k++) {
            u32 w; memcpy(&w, r.data() + 4 * k, 4);
            q.insert(w & mask);
            if (++cnt == n_queries) return std::vector<size_t>(q.begin(), q.end());
        }
    }
}
static inline std::vector<size_t> fold_queries(const std::vector<size_t>& q, u32 n_folds) {
    std::vector<size_t> out;
    for (size_t x : q) { size_t y = x >> n_folds; if (out.empty() || out.back() != y) out.push_back(y); }
    return out;
}
// compute_decommitment_positions_and_witness_evals
template <class AtFn>
static inline void decommitment_positions_and_witness(AtFn at, const std::vector<size_t>& queries, u32 fold_step,
                                                      std::vector<size_t>& positions, std::vector<QM31>& witness) {
    size_t i = 0;
    while (i < queries.size()) {
        size_t j = i;
        while (j < queries.size() && (queries[j] >> fold_step) == (queries[i] >> fold_step)) j++;
        size_t start = (queries[i] >> fold_step) << fold_step;
        size_t qi = i;
        for (size_t pos = start; pos < start + (size_t(1) << fold_step); pos++) {
            positions.push_back(pos);
            if (qi < j && queries[qi] == pos) { qi++; continue; }
            witness.push_back(at(pos));
        }
        i = j;
    }
}

// ---- the prover ----------------------------------------------------------------------------------------------------------------

struct Prover {
    PcsConfig cfg;
    u32 log_max_rows = 24;  // LOG_MAX_ROWS (mod.rs:428; 20 under cfg(test) mod.rs:433)
    std::function<void(const char*, const Channel&)> trace_hook;  // optional transcript tap for tests

    BrainfuckProof prove(const std::vector<Registers>& vm_trace, const std::vector<u32>& code) {
        // ORC_TIMING=1: wall-clock seconds per phase on stderr (where the port's time goes; bench.py's cpu_baseline notes)
        const bool timing = getenv("ORC_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "orc timing %-28s %8.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
            t_last = t;
        };
        // Protocol Setup (mod.rs:479-487)
        TwiddleTree tw = precompute_twiddles(CanonicCoset{log_max_rows + cfg.log_blowup + 2}.circle_domain().half_coset);
        Channel ch;
        std::vector<CommitmentTree> trees(4);
        lap("twiddles");

        // Phase 0 — preprocessed trace: IsFirst(LOG_MAX_ROWS..=LOG_N_LANES) (mod.rs:495-500)
        trees[0].polys.resize(log_max_rows - LOG_N_LANES + 1);
#pragma omp parallel for schedule(dynamic)
        for (u32 i = 0; i <= log_max_rows - LOG_N_LANES; i++) {
            u32 log = log_max_rows - i;
            std::vector<u32> col(size_t(1) << log, 0); col[0] = 1;   // gen_is_first
            trees[0].polys[i] = interpolate_col(col, log, tw);
        }
        lap("preprocessed interpolate");
        commit_tree(trees[0], cfg, tw, ch);
        tap("root0", ch);
        lap("preprocessed commit");

        // Phase 1 — main trace (mod.rs:506-583)
        std::vector<Table> tables = build_tables(vm_trace, code);
        BrainfuckProof bp;
        for (int c = 0; c < N_COMPONENTS; c++) {
            bp.log_sizes[c] = tables[c].log_size();
            if (bp.log_sizes[c] > log_max_rows) throw std::runtime_error("component exceeds LOG_MAX_ROWS");
        }
        {
            std::vector<std::pair<int, int>> jobs;
            for (int c = 0; c < N_COMPONENTS; c++) for (u32 k = 0; k < N_MAIN_COLS[c]; k++) jobs.push_back({c, (int)k});
            trees[1].polys.resize(jobs.size());
#pragma omp parallel for schedule(dynamic)
            for (size_t j = 0; j < jobs.size(); j++)
                trees[1].polys[j] = interpolate_col(broadcast16(tables[jobs[j].first].cols[jobs[j].second]), bp.log_sizes[jobs[j].first], tw);
        }
        for (int c = 0; c < N_COMPONENTS; c++) ch.mix_u64(bp.log_sizes[c]);     // claim.mix_into (mod.rs:102-116, components/mod.rs:132-134)
        lap("tables + main interpolate");
        commit_tree(trees[1], cfg, tw, ch);
        tap("root1", ch);
        lap("main commit");

        // Phase 2 — interaction trace (mod.rs:589-723)
        InteractionElements el;
        { auto f = ch.draw_felts(2); el.memory = LookupElements::make(f[0], f[1]); }        // MemoryElements::draw
        { auto f = ch.draw_felts(2); el.instruction = LookupElements::make(f[0], f[1]); }   // InstructionElements::draw
        { auto f = ch.draw_felts(2); el.processor = LookupElements::make(f[0], f[1]); }     // ProcessorElements::draw
        if (simd::enabled()) {
            // SIMD mode of the port (bench.py's cpu_baseline): the 13 components side by side (the reference's rayon would do no less); commit order kept
            std::vector<std::vector<PolyCol>> per(N_COMPONENTS);
#pragma omp parallel for schedule(dynamic)
            for (int c = 0; c < N_COMPONENTS; c++) {
                auto cols = gen_interaction_trace(c, tables[c], el, &bp.claimed_sums[c]);
                for (auto& col : cols) per[c].push_back(interpolate_col(col, bp.log_sizes[c], tw));
            }
            for (int c = 0; c < N_COMPONENTS; c++) for (auto& pc : per[c]) trees[2].polys.push_back(std::move(pc));
        } else
        for (int c = 0; c < N_COMPONENTS; c++) {
            auto cols = gen_interaction_trace(c, tables[c], el, &bp.claimed_sums[c]);
            for (auto& col : cols) trees[2].polys.push_back(interpolate_col(col, bp.log_sizes[c], tw));
        }
        for (int c = 0; c < N_COMPONENTS; c++) ch.mix_felts(&bp.claimed_sums[c], 1);         // interaction_claim.mix_into (mod.rs:189-203)
        lap("logUp + interpolate");
        commit_tree(trees[2], cfg, tw, ch);
        tap("root2", ch);
        lap("interaction commit");

        // Proof generation — stwo prover::prove (mod.rs:729-732)
        auto L = component_layouts(bp.log_sizes);
        QM31 random_coeff = ch.draw_felt();
        compute_composition(trees, L, el, bp.claimed_sums, random_coeff, tw);
        lap("composition");
        commit_tree(trees[3], cfg, tw, ch);
        tap("root3", ch);
        lap("composition commit");

        PointQ oods = get_random_point(ch);
        auto sample_points = mask_points(L, log_max_rows, oods);
        bp.proof = prove_values(trees, sample_points, ch, tw);
        lap("sampling, quotients, FRI");

        // Sanity check of prover::prove: composition OODS eval must match the constraints evaluated on the sampled mask.
        QM31 comp_eval[4];
        for (int k = 0; k < 4; k++) comp_eval[k] = bp.proof.sampled_values[3][k][0];
        if (from_partial_evals(comp_eval) != eval_composition_at_point(L, log_max_rows, el, bp.claimed_sums, oods, bp.proof.sampled_values, random_coeff))
            throw std::runtime_error("ConstraintsNotSatisfied");
        return bp;
    }

    static PointQ get_random_point(Channel& ch) {
        QM31 t = ch.draw_felt();
        QM31 t2 = t * t;
        QM31 d = inv(t2 + M31(1));
        return PointQ((QM31::one() - t2) * d, (t + t) * d);
    }

    // ComponentProvers::compute_composition_polynomial + DomainEvaluationAccumulator::finalize
    void compute_composition(std::vector<CommitmentTree>& trees, const std::vector<ComponentLayout>& L, const InteractionElements& el,
                             const QM31* claimed_sums, QM31 random_coeff, const TwiddleTree& tw) {
        int total = 0; u32 max_log = 0;
        for (auto& l : L) { total += l.n_constraints; max_log = std::max(max_log, l.log_size + 1); }
        std::vector<QM31> powers(total);
        { QM31 cur = QM31::one(); for (int i = 0; i < total; i++) { powers[i] = cur; cur = cur * random_coeff; } }
        std::vector<SecureCol*> sub(max_log + 1, nullptr);
        std::vector<SecureCol> storage(max_log + 1);
        int remaining = total;
        for (int c = 0; c < N_COMPONENTS; c++) {
            const auto& l = L[c];
            // accum.columns([(eval_log, n_constraints)]): split_off the LAST n powers, then reversed by the component.
            std::vector<QM31> coeff(l.n_constraints);
            for (int j = 0; j < l.n_constraints; j++) coeff[j] = powers[remaining - 1 - j];
            remaining -= l.n_constraints;
            u32 eval_log = l.log_size + 1;
            if (!sub[eval_log]) { storage[eval_log].init(eval_log); sub[eval_log] = &storage[eval_log]; }
            SecureCol& acc = *sub[eval_log];
            // denom_inv[i] = 1 / coset_vanishing(trace coset, eval_domain.at(i)), i < 2^log_expand, bit-reversed (size 2: identity)
            CircleDomain ed = CanonicCoset{eval_log}.circle_domain();
            M31 denom_inv[2];
            for (int i = 0; i < 2; i++) denom_inv[i] = inv(coset_vanishing<M31>(CanonicCoset{l.log_size}.coset(), ed.at(i)));
            std::vector<const u32*> tc(l.n_main), ic(l.n_inter);
            for (u32 k = 0; k < l.n_main; k++) tc[k] = trees[1].evals[l.main_off + k].values.data();
            for (u32 k = 0; k < l.n_inter; k++) ic[k] = trees[2].evals[l.inter_off + k].values.data();
            const u32* isf = trees[0].evals[log_max_rows - l.log_size].values.data();
            size_t n = size_t(1) << eval_log;
#pragma omp parallel for schedule(static)
            for (size_t row = 0; row < n; row++) {
                DomainEvaluator de;
                de.is_first_col = isf; de.trace_cols = tc.data(); de.inter_cols = ic.data();
                de.row = row; de.log_size = l.log_size; de.eval_log = eval_log; de.coeff = coeff.data();
                de.total_sum = claimed_sums[c];
                eval_component(c, de, el);
                acc.set(row, acc.at(row) + de.row_res * denom_inv[row >> l.log_size]);
            }
        }
        if (remaining != 0) throw std::runtime_error("not all random coefficients were used");
        // finalize: small -> large; interpolate at size s, evaluate on the next populated size, add.
        bool have = false;
        PolyCol cur[4];
        for (u32 log = 1; log <= max_log; log++) {
            if (!sub[log]) continue;
            SecureCol& v = *sub[log];
            if (have)
                for (int k = 0; k < 4; k++) {
                    EvalCol e = evaluate_poly(cur[k], log, tw);
                    for (size_t i = 0; i < e.values.size(); i++) v.c[k][i] = (M31(v.c[k][i]) + M31(e.values[i])).v;
                }
            for (int k = 0; k < 4; k++) cur[k] = interpolate_col(v.c[k], log, tw);
            have = true;
        }
        trees[3].polys.clear();
        for (int k = 0; k < 4; k++) trees[3].polys.push_back(cur[k]);
    }

    // CommitmentSchemeProver::prove_values
    StarkProof prove_values(std::vector<CommitmentTree>& trees, const std::vector<std::vector<std::vector<PointQ>>>& sample_points, Channel& ch, const TwiddleTree&) {
        StarkProof pf;
        const bool timing = getenv("ORC_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "orc timing   %-26s %8.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
            t_last = t;
        };
        // OODS sampling (a8)
        std::vector<std::vector<std::vector<PointSample>>> samples(trees.size());
        pf.sampled_values.resize(trees.size());
        for (size_t t = 0; t < trees.size(); t++) {
            samples[t].resize(trees[t].polys.size());
            pf.sampled_values[t].resize(trees[t].polys.size());
#pragma omp parallel for schedule(dynamic)
            for (size_t c = 0; c < trees[t].polys.size(); c++)
                for (auto& pt : sample_points[t][c]) {
                    QM31 v = eval_at_point(reinterpret_cast<const M31*>(trees[t].polys[c].coeffs.data()), trees[t].polys[c].log_size, pt);
                    samples[t][c].push_back({pt, v});
                    pf.sampled_values[t][c].push_back(v);
                }
        }
        { std::vector<QM31> flat; for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); ch.mix_felts(flat.data(), flat.size()); }
        tap("sampled", ch);
        lap("sampling");
        QM31 random_coeff = ch.draw_felt();

        // compute_fri_quotients: group columns by LDE log size (descending, stable), one SecureEvaluation per size.
        std::vector<std::pair<const EvalCol*, const std::vector<PointSample>*>> flat_cols;
        for (size_t t = 0; t < trees.size(); t++) for (size_t c = 0; c < trees[t].evals.size(); c++) flat_cols.push_back({&trees[t].evals[c], &samples[t][c]});
        std::stable_sort(flat_cols.begin(), flat_cols.end(), [](auto& a, auto& b) { return a.first->log_size > b.first->log_size; });
        std::vector<SecureCol> quotients;
        for (size_t i = 0; i < flat_cols.size();) {
            size_t j = i; u32 log = flat_cols[i].first->log_size;
            while (j < flat_cols.size() && flat_cols[j].first->log_size == log) j++;
            std::vector<const std::vector<PointSample>*> smp; std::vector<const u32*> cols;
            for (size_t k = i; k < j; k++) { smp.push_back(flat_cols[k].second); cols.push_back(flat_cols[k].first->values.data()); }
            auto sb = sample_batches(smp);
            auto qc = quotient_constants(sb, random_coeff);
            auto pts = domain_points(log);
            SecureCol q; q.init(log);
            size_t n = size_t(1) << log;
            const size_t CH = 1024, nb = sb.size();
            // SIMD mode of the port (orc_set_simd: bench.py's cpu_baseline): the column loop of accumulate_row_quotients on 16 rows per
            // instruction (simd_port.cpp); denominators and their batch inverse stay the scalar code. Same values.
            const bool use_simd = simd::enabled() && n >= 16;
            std::vector<u32> lc_flat; std::vector<u32> ci_flat; std::vector<size_t> lc_off(nb + 1, 0);
            if (use_simd) {
                for (size_t b = 0; b < nb; b++) {
                    lc_off[b] = ci_flat.size();
                    for (size_t k = 0; k < sb[b].cols.size(); k++) {
                        ci_flat.push_back((u32)sb[b].cols[k].first);
                        for (int w = 0; w < 3; w++) { auto a = qc.line_coeffs[b][k][w].to_u32(); lc_flat.insert(lc_flat.end(), a.begin(), a.end()); }
                    }
                }
                lc_off[nb] = ci_flat.size();
            }
#pragma omp parallel for schedule(static)
            for (size_t r0 = 0; r0 < n; r0 += CH) {
                size_t r1 = std::min(n, r0 + CH);
                std::vector<CM31> den((r1 - r0) * nb), deninv((r1 - r0) * nb);
                for (size_t row = r0; row < r1; row++) for (size_t b = 0; b < nb; b++) den[(row - r0) * nb + b] = quotient_denominator(sb[b], pts[bit_reverse_index((u32)row, log)]);
                batch_inverse(den.data(), deninv.data(), den.size());
                if (use_simd) {
                    const size_t m = r1 - r0;
                    std::vector<u32> ys(m), da(m * nb), db(m * nb);
                    for (size_t row = r0; row < r1; row++) {
                        ys[row - r0] = pts[bit_reverse_index((u32)row, log)].y.v;
                        for (size_t b = 0; b < nb; b++) { da[b * m + (row - r0)] = deninv[(row - r0) * nb + b].a.v; db[b * m + (row - r0)] = deninv[(row - r0) * nb + b].b.v; }
                    }
                    std::vector<simd::QuotientBatch> qb(nb);
                    for (size_t b = 0; b < nb; b++) {
                        qb[b].col_index = ci_flat.data() + lc_off[b]; qb[b].line_coeffs = lc_flat.data() + 12 * lc_off[b]; qb[b].n_cols = lc_off[b + 1] - lc_off[b];
                        auto bc = qc.batch_random_coeffs[b].to_u32(); for (int w = 0; w < 4; w++) qb[b].batch_coeff[w] = bc[w];
                        qb[b].deninv_a = da.data() + b * m; qb[b].deninv_b = db.data() + b * m;
                    }
                    u32* outp[4] = {q.c[0].data(), q.c[1].data(), q.c[2].data(), q.c[3].data()};
                    simd::quotient_rows(cols.data(), ys.data(), qb.data(), nb, r0, r1, outp);
                    continue;
                }
                std::vector<u32> vals(cols.size());
                for (size_t row = r0; row < r1; row++) {
                    for (size_t k = 0; k < cols.size(); k++) vals[k] = cols[k][row];
                    q.set(row, accumulate_row_quotients(sb, vals.data(), qc, pts[bit_reverse_index((u32)row, log)], nb ? &deninv[(row - r0) * nb] : nullptr));
                }
            }
            quotients.push_back(std::move(q));
            i = j;
        }

        lap("quotients");
        // FriProver::commit
        auto coord_refs = [](const SecureCol& s) { std::vector<ColRef> r; for (int k = 0; k < 4; k++) r.push_back({s.c[k].data(), s.log_size}); return r; };
        std::vector<ColRef> first_refs;
        for (auto& q : quotients) for (auto& r : coord_refs(q)) first_refs.push_back(r);
        MerkleProver first_tree = MerkleProver::commit(first_refs);
        ch.mix_root(first_tree.root());
        lap("FRI first-layer tree");
        struct Inner { std::vector<QM31> eval; u32 log; SecureCol sc; MerkleProver tree; };
        std::vector<Inner> inner;
        u32 line_log = quotients[0].log_size - 1;
        std::vector<QM31> layer(size_t(1) << line_log, QM31::zero());
        size_t qi = 0;
        QM31 folding_alpha = ch.draw_felt();
        size_t last_size = size_t(1) << (cfg.log_last_layer_degree_bound + cfg.log_blowup);
        while (layer.size() > last_size) {
            while (qi < quotients.size() && (size_t(1) << (quotients[qi].log_size - 1)) == layer.size()) fold_circle_into_line(layer, quotients[qi++], folding_alpha);
            Inner in; in.eval = layer; in.log = line_log; in.sc.init(line_log);
#pragma omp parallel for schedule(static) if (simd::enabled() && layer.size() >= 4096)
            for (size_t i = 0; i < layer.size(); i++) in.sc.set(i, layer[i]);
            in.tree = MerkleProver::commit(coord_refs(in.sc));
            ch.mix_root(in.tree.root());
            folding_alpha = ch.draw_felt();
            layer = fold_line(in.eval, line_log, folding_alpha);
            line_log--;
            inner.push_back(std::move(in));
        }
        if (qi != quotients.size()) throw std::runtime_error("FRI: not all columns consumed");
        // commit_last_layer
        auto coeffs = line_interpolate(layer, LineDomain{Coset::half_odds(line_log)});
        size_t bound = size_t(1) << cfg.log_last_layer_degree_bound;
        for (size_t i = bound; i < coeffs.size(); i++) if (!coeffs[i].is_zero()) throw std::runtime_error("invalid degree");
        coeffs.resize(bound);
        ch.mix_felts(coeffs.data(), coeffs.size());
        pf.fri_proof.last_layer_coeffs = coeffs;
        pf.fri_proof.last_layer_log_size = cfg.log_last_layer_degree_bound;
        tap("fri_commit", ch);
        lap("FRI folds + layer trees");

        // Proof of work
        pf.proof_of_work = grind(ch, cfg.pow_bits);
        ch.mix_u64(pf.proof_of_work);

        // FRI decommit
        u32 max_col_log = quotients[0].log_size;
        auto queries = generate_queries(ch, max_col_log, cfg.n_queries);
        std::map<u32, std::vector<size_t>> positions_by_log;
        for (auto& q : quotients) positions_by_log[q.log_size] = fold_queries(queries, max_col_log - q.log_size);
        {
            std::map<u32, std::vector<size_t>> dpos;
            for (auto& q : quotients) {
                auto cq = fold_queries(queries, max_col_log - q.log_size);
                std::vector<size_t> pos;
                decommitment_positions_and_witness([&](size_t p) { return q.at(p); }, cq, 1, pos, pf.fri_proof.first_layer.fri_witness);
                dpos[q.log_size] = pos;
            }
            pf.fri_proof.first_layer.decommitment = first_tree.decommit(dpos, first_refs).second;
            pf.fri_proof.first_layer.commitment = first_tree.root();
        }
        auto lq = fold_queries(queries, 1);
        for (auto& in : inner) {
            FriLayerProof lp;
            std::vector<size_t> pos;
            decommitment_positions_and_witness([&](size_t p) { return in.eval[p]; }, lq, 1, pos, lp.fri_witness);
            std::map<u32, std::vector<size_t>> dpos; dpos[in.log] = pos;
            lp.decommitment = in.tree.decommit(dpos, coord_refs(in.sc)).second;
            lp.commitment = in.tree.root();
            pf.fri_proof.inner_layers.push_back(std::move(lp));
            lq = fold_queries(lq, 1);
        }
        // Decommit the FRI queries on the trace trees.
        for (auto& t : trees) {
            auto r = t.merkle.decommit(positions_by_log, t.col_refs());
            pf.queried_values.push_back(r.first);
            pf.decommitments.push_back(r.second);
            pf.commitments.push_back(t.merkle.root());
        }
        return pf;
    }

    std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
    void tap(const char* name, const Channel& ch) {
        if (getenv("ORC_TIMING")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[oracle] %-12s +%.3fs\n", name, std::chrono::duration<double>(t - t_last).count()); t_last = t; }
        if (trace_hook) trace_hook(name, ch);
    }
};

// ---- the verifier (verify_brainfuck mod.rs:738-797 + stwo verify / CommitmentSchemeVerifier::verify_values / FriVerifier) -----------
struct Verifier {
    PcsConfig cfg;
    u32 log_max_rows = 24;

    // returns "" when the proof verifies, else an error string
    std::string verify(const BrainfuckProof& bp) const {
        try { return verify_inner(bp); } catch (const std::exception& e) { return std::string("InvalidStructure: ") + e.what(); }
    }

   private:
    std::string verify_inner(const BrainfuckProof& bp) const {
        const StarkProof& pf = bp.proof;
        if (pf.commitments.size() != 4 || pf.sampled_values.size() != 4 || pf.decommitments.size() != 4 || pf.queried_values.size() != 4) return "InvalidStructure";
        for (int c = 0; c < N_COMPONENTS; c++) if (bp.log_sizes[c] < LOG_N_LANES || bp.log_sizes[c] > log_max_rows) return "InvalidStructure: log_size";
        Channel ch;
        auto L = component_layouts(bp.log_sizes);
        // claim.log_sizes() (mod.rs:118-143): preprocessed tree overwritten with all IS_FIRST_LOG_SIZES
        std::vector<std::vector<u32>> log_sizes(4);
        for (u32 log = log_max_rows; log >= LOG_N_LANES; log--) log_sizes[0].push_back(log);
        for (auto& l : L) { for (u32 k = 0; k < l.n_main; k++) log_sizes[1].push_back(l.log_size); for (u32 k = 0; k < l.n_inter; k++) log_sizes[2].push_back(l.log_size); }
        std::vector<MerkleVerifier> trees;
        auto commit = [&](int t) {
            ch.mix_root(pf.commitments[t]);
            std::vector<u32> ext; for (u32 l : log_sizes[t]) ext.push_back(l + cfg.log_blowup);
            trees.push_back(MerkleVerifier(pf.commitments[t], ext));
        };
        commit(0);
        for (int c = 0; c < N_COMPONENTS; c++) ch.mix_u64(bp.log_sizes[c]);
        commit(1);
        InteractionElements el;
        { auto f = ch.draw_felts(2); el.memory = LookupElements::make(f[0], f[1]); }
        { auto f = ch.draw_felts(2); el.instruction = LookupElements::make(f[0], f[1]); }
        { auto f = ch.draw_felts(2); el.processor = LookupElements::make(f[0], f[1]); }
        // lookup_sum_valid (mod.rs:207-227)
        { QM31 s = QM31::zero(); for (int c = 0; c < N_COMPONENTS; c++) s = s + bp.claimed_sums[c]; if (!s.is_zero()) return "InvalidLookup: Invalid LogUp sum"; }
        for (int c = 0; c < N_COMPONENTS; c++) ch.mix_felts(&bp.claimed_sums[c], 1);
        commit(2);

        // stwo verify()
        QM31 random_coeff = ch.draw_felt();
        u32 comp_log = 0; for (auto& l : L) comp_log = std::max(comp_log, l.log_size + 1);   // composition_log_degree_bound
        log_sizes[3].assign(4, comp_log);
        commit(3);
        PointQ oods = Prover::get_random_point(ch);
        auto sp = mask_points(L, log_max_rows, oods);
        for (int t = 0; t < 4; t++) {
            if (pf.sampled_values[t].size() != sp[t].size()) return "InvalidStructure: sampled_values";
            for (size_t c = 0; c < sp[t].size(); c++) if (pf.sampled_values[t][c].size() != sp[t][c].size()) return "InvalidStructure: sampled_values";
        }
        QM31 comp_eval[4];
        for (int k = 0; k < 4; k++) comp_eval[k] = pf.sampled_values[3][k][0];
        if (from_partial_evals(comp_eval) != eval_composition_at_point(L, log_max_rows, el, bp.claimed_sums, oods, pf.sampled_values, random_coeff)) return "OodsNotMatching";

        // verify_values
        { std::vector<QM31> flat; for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); ch.mix_felts(flat.data(), flat.size()); }
        QM31 q_coeff = ch.draw_felt();
        std::set<u32, std::greater<u32>> col_logs;
        for (auto& t : trees) for (u32 l : t.column_log_sizes) col_logs.insert(l);
        std::vector<u32> column_domain_logs(col_logs.begin(), col_logs.end());   // descending; bound = log - blowup
        // FriVerifier::commit
        const FriProof& fp = pf.fri_proof;
        ch.mix_root(fp.first_layer.commitment);
        QM31 first_alpha = ch.draw_felt();
        u32 layer_bound = column_domain_logs[0] - cfg.log_blowup - 1;    // max_column_bound.fold_to_line()
        std::vector<QM31> inner_alphas;
        for (auto& lp : fp.inner_layers) {
            ch.mix_root(lp.commitment);
            inner_alphas.push_back(ch.draw_felt());
            if (layer_bound == 0) return "InvalidNumFriLayers";
            layer_bound -= 1;
        }
        if (layer_bound != cfg.log_last_layer_degree_bound) return "InvalidNumFriLayers";
        if (fp.last_layer_coeffs.size() > (size_t(1) << cfg.log_last_layer_degree_bound)) return "LastLayerDegreeInvalid";
        // the reference evaluates the last-layer polynomial by folding 2^log_size coefficients (it panics on any other count)
        if (fp.last_layer_log_size > 31 || (size_t(1) << fp.last_layer_log_size) != fp.last_layer_coeffs.size()) return "LastLayerDegreeInvalid";
        ch.mix_felts(fp.last_layer_coeffs.data(), fp.last_layer_coeffs.size());
        // proof of work
        ch.mix_u64(pf.proof_of_work);
        if (ch.trailing_zeros() < cfg.pow_bits) return "ProofOfWork";
        // queries
        u32 max_col_log = column_domain_logs[0];
        auto queries = generate_queries(ch, max_col_log, cfg.n_queries);
        std::map<u32, std::vector<size_t>> positions_by_log;
        for (u32 l : column_domain_logs) positions_by_log[l] = fold_queries(queries, max_col_log - l);
        // Merkle decommitments of the 4 trees
        for (int t = 0; t < 4; t++) { auto e = trees[t].verify(positions_by_log, pf.queried_values[t], pf.decommitments[t]); if (!e.empty()) return "MerkleVerification tree " + std::to_string(t) + ": " + e; }
        // fri_answers
        std::vector<std::vector<PointSample>> flat_samples; std::vector<u32> flat_logs; std::vector<int> flat_tree;
        for (int t = 0; t < 4; t++) for (size_t c = 0; c < sp[t].size(); c++) {
            std::vector<PointSample> s; for (size_t k = 0; k < sp[t][c].size(); k++) s.push_back({sp[t][c][k], pf.sampled_values[t][c][k]});
            flat_samples.push_back(s); flat_logs.push_back(trees[t].column_log_sizes[c]); flat_tree.push_back(t);
        }
        std::vector<size_t> order(flat_logs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return flat_logs[a] > flat_logs[b]; });
        size_t qv_pos[4] = {0, 0, 0, 0};
        std::vector<std::vector<QM31>> fri_answers;
        for (size_t i = 0; i < order.size();) {
            size_t j = i; u32 log = flat_logs[order[i]];
            while (j < order.size() && flat_logs[order[j]] == log) j++;
            std::vector<const std::vector<PointSample>*> smp;
            for (size_t k = i; k < j; k++) smp.push_back(&flat_samples[order[k]]);
            auto sb = sample_batches(smp);
            auto qc = quotient_constants(sb, q_coeff);
            CircleDomain dom = CanonicCoset{log}.circle_domain();
            size_t ncols[4];
            for (int t = 0; t < 4; t++) { auto it = trees[t].n_columns_per_log_size.find(log); ncols[t] = it == trees[t].n_columns_per_log_size.end() ? 0 : it->second; }
            std::vector<QM31> answers;
            for (size_t qpos : positions_by_log[log]) {
                PointM dp = dom.at(bit_reverse_index((u32)qpos, log));
                std::vector<u32> vals;
                for (int t = 0; t < 4; t++) for (size_t k = 0; k < ncols[t]; k++) { if (qv_pos[t] >= pf.queried_values[t].size()) return "InvalidStructure: queried_values"; vals.push_back(pf.queried_values[t][qv_pos[t]++]); }
                answers.push_back(accumulate_row_quotients(sb, vals.data(), qc, dp));
            }
            fri_answers.push_back(answers);
            i = j;
        }
        // FriVerifier::decommit
        // first layer
        struct Sparse { std::vector<std::array<QM31, 2>> evals; std::vector<size_t> starts; };
        auto rebuild = [&](const std::vector<size_t>& q, const std::vector<QM31>& qevals, const std::vector<QM31>& wit, size_t& wi, std::vector<size_t>& positions, Sparse& sp_out) -> bool {
            size_t i = 0, ei = 0;
            while (i < q.size()) {
                size_t j = i; while (j < q.size() && (q[j] >> 1) == (q[i] >> 1)) j++;
                size_t start = (q[i] >> 1) << 1; size_t qi2 = i;
                std::array<QM31, 2> ev;
                for (size_t pos = start; pos < start + 2; pos++) {
                    positions.push_back(pos);
                    if (qi2 < j && q[qi2] == pos) { qi2++; ev[pos - start] = qevals[ei++]; }
                    else { if (wi >= wit.size()) return false; ev[pos - start] = wit[wi++]; }
                }
                sp_out.evals.push_back(ev); sp_out.starts.push_back(start);
                i = j;
            }
            return true;
        };
        if (fri_answers.size() != column_domain_logs.size()) return "InvalidStructure: fri answers";
        std::vector<Sparse> first_sparse(column_domain_logs.size());
        {
            size_t wi = 0; std::map<u32, std::vector<size_t>> dpos; std::vector<u32> dvals; std::vector<u32> mlogs;
            for (size_t k = 0; k < column_domain_logs.size(); k++) {
                u32 log = column_domain_logs[k];
                auto cq = fold_queries(queries, max_col_log - log);
                std::vector<size_t> pos;
                if (!rebuild(cq, fri_answers[k], fp.first_layer.fri_witness, wi, pos, first_sparse[k])) return "FirstLayerEvaluationsInvalid";
                dpos[log] = pos;
                for (auto& ev : first_sparse[k].evals) for (auto& v : ev) { auto a = v.to_u32(); for (u32 x : a) dvals.push_back(x); }
                for (int c = 0; c < 4; c++) mlogs.push_back(log);
            }
            if (wi != fp.first_layer.fri_witness.size()) return "FirstLayerEvaluationsInvalid";
            auto e = MerkleVerifier(fp.first_layer.commitment, mlogs).verify(dpos, dvals, fp.first_layer.decommitment);
            if (!e.empty()) return "FirstLayerCommitmentInvalid: " + e;
        }
        // inner layers
        auto lq = fold_queries(queries, 1);
        std::vector<QM31> lev(lq.size(), QM31::zero());
        size_t fk = 0; QM31 prev_alpha = first_alpha;
        u32 line_log = max_col_log - 1;
        for (size_t li = 0; li < fp.inner_layers.size(); li++) {
            while (fk < column_domain_logs.size() && column_domain_logs[fk] - 1 == line_log) {
                // fold_circle of the sparse evals of column fk into this layer
                u32 clog = column_domain_logs[fk];
                CircleDomain dom = CanonicCoset{clog}.circle_domain();
                if (first_sparse[fk].evals.size() != lev.size()) return "InvalidStructure: sparse evals";
                QM31 a2 = prev_alpha * prev_alpha;
                for (size_t s = 0; s < first_sparse[fk].evals.size(); s++) {
                    PointM p = dom.at(bit_reverse_index((u32)first_sparse[fk].starts[s], clog));
                    QM31 f0 = first_sparse[fk].evals[s][0], f1 = first_sparse[fk].evals[s][1];
                    ibutterfly(f0, f1, inv(p.y));
                    lev[s] = lev[s] * a2 + (f0 + prev_alpha * f1);
                }
                fk++;
            }
            const auto& lp = fp.inner_layers[li];
            size_t wi = 0; std::vector<size_t> pos; Sparse sps;
            if (!rebuild(lq, lev, lp.fri_witness, wi, pos, sps)) return "InnerLayerEvaluationsInvalid";
            if (wi != lp.fri_witness.size()) return "InnerLayerEvaluationsInvalid";
            std::vector<u32> dvals; for (auto& ev : sps.evals) for (auto& v : ev) { auto a = v.to_u32(); for (u32 x : a) dvals.push_back(x); }
            std::map<u32, std::vector<size_t>> dpos; dpos[line_log] = pos;
            auto e = MerkleVerifier(lp.commitment, std::vector<u32>(4, line_log)).verify(dpos, dvals, lp.decommitment);
            if (!e.empty()) return "InnerLayerCommitmentInvalid: " + e;
            // fold_line
            LineDomain ld{Coset::half_odds(line_log)};
            std::vector<QM31> nev;
            for (size_t s = 0; s < sps.evals.size(); s++) {
                M31 x = ld.at(bit_reverse_index((u32)sps.starts[s], line_log));
                QM31 f0 = sps.evals[s][0], f1 = sps.evals[s][1];
                ibutterfly(f0, f1, inv(x));
                nev.push_back(f0 + inner_alphas[li] * f1);
            }
            lq = fold_queries(lq, 1); lev = nev; prev_alpha = inner_alphas[li]; line_log--;
        }
        if (fk != column_domain_logs.size()) return "InvalidStructure: unconsumed first-layer columns";
        // last layer: constant polynomial (degree bound 2^0)
        if (lev.size() != lq.size()) return "InvalidStructure";
        for (size_t i = 0; i < lq.size(); i++) {
            QM31 expect = fp.last_layer_coeffs.empty() ? QM31::zero() : fp.last_layer_coeffs[0];
            if (cfg.log_last_layer_degree_bound != 0) return "unsupported last layer bound";
            if (lev[i] != expect) return "LastLayerEvaluationsInvalid";
        }
        return "";
    }
};

}  // namespace orc
