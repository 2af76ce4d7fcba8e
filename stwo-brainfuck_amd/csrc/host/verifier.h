// verify_brainfuck (crates/brainfuck_prover/src/brainfuck_air/mod.rs:738-797) — the other half of the reference's public interface.
// Verification is CPU work in the reference too (channel replay, a few hundred hashes, a handful of field operations), so this is
// plain host C++: stwo `verify` / `CommitmentSchemeVerifier::verify_values` / `MerkleVerifier::verify` / `FriVerifier` / `fri_answers`
// restated over the product's own field, channel and AIR code (air.h). It also parses the serde_json proof form (proof.h).
#pragma once
#include "circle.h"
#include "proof.h"
#include <map>
#include <array>
#include <set>
#include <string>
#include <stdexcept>
#include <functional>

namespace bf {

// ---- minimal JSON reader (unsigned integers, arrays, objects, null) ------------------------------------------------------------------
struct JVal {
    enum Kind { NUM, ARR, OBJ, NUL, STR } kind = NUL;
    u64 num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal& get(const char* k) const { for (auto& kv : obj) if (kv.first == k) return kv.second; throw std::runtime_error(std::string("missing key ") + k); }
};
struct JsonReader {
    const char* p; const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    JVal parse(int depth = 0) {
        if (depth > 16) throw std::runtime_error("json: too deep");
        ws();
        if (p >= end) throw std::runtime_error("json: eof");
        JVal v;
        if (*p == '[') {
            v.kind = JVal::ARR; p++; ws();
            if (p < end && *p == ']') { p++; return v; }
            for (;;) { v.arr.push_back(parse(depth + 1)); ws(); if (p < end && *p == ',') { p++; continue; } if (p < end && *p == ']') { p++; break; } throw std::runtime_error("json: array"); }
        } else if (*p == '{') {
            v.kind = JVal::OBJ; p++; ws();
            if (p < end && *p == '}') { p++; return v; }
            for (;;) {
                ws(); if (p >= end || *p != '"') throw std::runtime_error("json: key");
                p++; const char* s = p; while (p < end && *p != '"') p++; if (p >= end) throw std::runtime_error("json: key"); std::string k(s, p); p++;
                ws(); if (p >= end || *p != ':') throw std::runtime_error("json: colon"); p++;
                for (auto& kv : v.obj) if (kv.first == k) throw std::runtime_error("json: duplicate key");   // serde_json: "duplicate field"
                v.obj.push_back({k, parse(depth + 1)}); ws();
                if (p < end && *p == ',') { p++; continue; } if (p < end && *p == '}') { p++; break; } throw std::runtime_error("json: object");
            }
        } else if (*p == '"') {   // strings occur only as felt252 hashes ("0x..." hex), never with escapes
            p++; const char* s0 = p; while (p < end && *p != '"' && *p != '\\') p++;
            if (p >= end || *p != '"') throw std::runtime_error("json: string");
            v.kind = JVal::STR; v.str.assign(s0, p); p++;
        } else if (*p == 'n') { if (end - p < 4 || memcmp(p, "null", 4) != 0) throw std::runtime_error("json: null"); p += 4; v.kind = JVal::NUL; }
        else if (*p >= '0' && *p <= '9') {
            // canonical unsigned integers only, as serde_json emits and accepts for u32/u64 fields: no leading zeros, no wrap-around
            v.kind = JVal::NUM; u64 x = 0; const char* s0 = p;
            while (p < end && *p >= '0' && *p <= '9') {
                u64 d = (u64)(*p - '0');
                if (x > (~u64(0) - d) / 10) throw std::runtime_error("json: number out of range");
                x = x * 10 + d; p++;
            }
            if (p - s0 > 1 && *s0 == '0') throw std::runtime_error("json: leading zero");
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) throw std::runtime_error("json: not an integer");
            v.num = x;
        }
        else throw std::runtime_error("json: unexpected character");
        return v;
    }
};
// a sequence field: serde_json refuses any other JSON type (null, object, number) where a Vec is expected
inline const std::vector<JVal>& jv_list(const JVal& v) { if (v.kind != JVal::ARR) throw std::runtime_error("expected a sequence"); return v.arr; }
inline u32 jv_m31(const JVal& v) { if (v.kind != JVal::NUM || v.num >= P31) throw std::runtime_error("bad M31"); return (u32)v.num; }
inline Q31 jv_qm31(const JVal& v) {
    if (v.kind != JVal::ARR || v.arr.size() != 2 || v.arr[0].kind != JVal::ARR || v.arr[1].kind != JVal::ARR || v.arr[0].arr.size() != 2 || v.arr[1].arr.size() != 2) throw std::runtime_error("bad QM31");
    return q_make(jv_m31(v.arr[0].arr[0]), jv_m31(v.arr[0].arr[1]), jv_m31(v.arr[1].arr[0]), jv_m31(v.arr[1].arr[1]));
}
inline Hash32 jv_hash(const JVal& v, bool felt) {
    if (felt != (v.kind == JVal::STR)) throw std::runtime_error("hash form does not match the Merkle channel");
    if (v.kind == JVal::STR) {   // Poseidon252MerkleHasher::Hash = FieldElement252, serialised by starknet-ff as "0x" + minimal lowercase hex
        const std::string& t = v.str;
        if (t.size() < 3 || t.size() > 66 || t[0] != '0' || t[1] != 'x' || (t.size() > 3 && t[2] == '0')) throw std::runtime_error("bad felt252 hash");
        Hash32 h; memset(h.b, 0, 32);
        size_t nd = t.size() - 2;
        for (size_t i = 0; i < nd; i++) {
            char ch = t[t.size() - 1 - i]; int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : -1;
            if (d < 0) throw std::runtime_error("bad felt252 hash digit");
            h.b[i / 2] |= (u8)(d << (4 * (i & 1)));
        }
        if (!fe252::canonical_bytes_in_range(h.b)) throw std::runtime_error("felt252 hash out of range");
        return h;
    }
    if (v.kind != JVal::ARR || v.arr.size() != 32) throw std::runtime_error("bad hash");
    Hash32 h; for (int i = 0; i < 32; i++) { if (v.arr[i].kind != JVal::NUM || v.arr[i].num > 255) throw std::runtime_error("bad hash byte"); h.b[i] = (u8)v.arr[i].num; } return h;
}
inline MerkleDecommitment jv_decommitment(const JVal& v, bool felt) {
    MerkleDecommitment d;
    for (auto& h : jv_list(v.get("hash_witness"))) d.hash_witness.push_back(jv_hash(h, felt));
    for (auto& x : jv_list(v.get("column_witness"))) d.column_witness.push_back(jv_m31(x));
    return d;
}
inline FriLayerProof jv_fri_layer(const JVal& v, bool felt) {
    FriLayerProof l;
    for (auto& q : jv_list(v.get("fri_witness"))) l.fri_witness.push_back(jv_qm31(q));
    l.decommitment = jv_decommitment(v.get("decommitment"), felt);
    l.commitment = jv_hash(v.get("commitment"), felt);
    return l;
}
inline BrainfuckProof proof_from_json(const char* s, size_t len, bool felt = false) {
    JsonReader jr{s, s + len};
    JVal root = jr.parse();
    jr.ws();
    if (jr.p != jr.end) throw std::runtime_error("json: trailing characters");
    BrainfuckProof bp;
    for (int c = 0; c < N_COMPONENTS; c++) {
        const JVal& ls = root.get("claim").get(COMPONENT_NAMES[c]).get("log_size");
        if (ls.kind != JVal::NUM || ls.num > 31) throw std::runtime_error("bad log_size");
        bp.log_sizes[c] = (u32)ls.num;
        bp.claimed_sums[c] = jv_qm31(root.get("interaction_claim").get(COMPONENT_NAMES[c]).get("claimed_sum"));
    }
    const JVal& p = root.get("proof");
    StarkProof& sp = bp.proof;
    for (auto& h : jv_list(p.get("commitments"))) sp.commitments.push_back(jv_hash(h, felt));
    for (auto& t : jv_list(p.get("sampled_values"))) {
        std::vector<std::vector<Q31>> tv;
        for (auto& c : jv_list(t)) { std::vector<Q31> cv; for (auto& q : jv_list(c)) cv.push_back(jv_qm31(q)); tv.push_back(cv); }
        sp.sampled_values.push_back(tv);
    }
    for (auto& d : jv_list(p.get("decommitments"))) sp.decommitments.push_back(jv_decommitment(d, felt));
    for (auto& t : jv_list(p.get("queried_values"))) { std::vector<u32> v; for (auto& x : jv_list(t)) v.push_back(jv_m31(x)); sp.queried_values.push_back(v); }
    if (p.get("proof_of_work").kind != JVal::NUM) throw std::runtime_error("bad proof_of_work");
    sp.proof_of_work = p.get("proof_of_work").num;
    const JVal& f = p.get("fri_proof");
    sp.fri_proof.first_layer = jv_fri_layer(f.get("first_layer"), felt);
    for (auto& l : jv_list(f.get("inner_layers"))) sp.fri_proof.inner_layers.push_back(jv_fri_layer(l, felt));
    for (auto& q : jv_list(f.get("last_layer_poly").get("coeffs"))) sp.fri_proof.last_layer_coeffs.push_back(jv_qm31(q));
    { const JVal& ll = f.get("last_layer_poly").get("log_size"); if (ll.kind != JVal::NUM || ll.num > 31) throw std::runtime_error("bad last layer log_size"); sp.fri_proof.last_layer_log_size = (u32)ll.num; }
    return bp;
}

// ---- verifier ----------------------------------------------------------------------------------------------------------------------------
struct VerifierConfig { u32 pow_bits = 5, log_blowup = 1, log_last_layer_degree_bound = 0, n_queries = 3; };   // PcsConfig::default() (mod.rs:743)

// MerkleVerifier::verify — "" on success, else the error name
inline std::string merkle_verify(const Hash32& root, const std::vector<u32>& column_log_sizes, const std::map<u32, std::vector<size_t>>& queries_per_log,
                                 const std::vector<u32>& queried_values, const MerkleDecommitment& d, const Conventions& node_conv) {
    if (column_log_sizes.empty()) return "";
    std::map<u32, size_t> ncols_at;
    u32 max_log = 0;
    for (u32 l : column_log_sizes) { ncols_at[l]++; max_log = std::max(max_log, l); }
    size_t qv = 0, hw = 0, cw = 0;
    std::vector<std::pair<size_t, Hash32>> last;
    bool have_last = false;
    static const std::vector<size_t> empty;
    for (int log = (int)max_log; log >= 0; log--) {
        size_t ncol = ncols_at.count((u32)log) ? ncols_at[(u32)log] : 0;
        auto it = queries_per_log.find((u32)log);
        const std::vector<size_t>& colq = it == queries_per_log.end() ? empty : it->second;
        std::vector<std::pair<size_t, Hash32>> total;
        size_t pi = 0, qi = 0, hi = 0;
        while (pi < last.size() || qi < colq.size()) {
            size_t node;
            if (pi < last.size() && qi < colq.size()) node = std::min(last[pi].first / 2, colq[qi]);
            else if (pi < last.size()) node = last[pi].first / 2;
            else node = colq[qi];
            while (pi < last.size() && last[pi].first / 2 == node) pi++;
            Hash32 l, r;
            if (have_last) {
                if (hi < last.size() && last[hi].first == 2 * node) l = last[hi++].second;
                else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; l = d.hash_witness[hw++]; }
                if (hi < last.size() && last[hi].first == 2 * node + 1) r = last[hi++].second;
                else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; r = d.hash_witness[hw++]; }
            }
            std::vector<u32> vals(ncol);
            if (qi < colq.size() && colq[qi] == node) {
                qi++;
                if (qv + ncol > queried_values.size()) return "TooFewQueriedValues";
                for (size_t c = 0; c < ncol; c++) vals[c] = queried_values[qv++];
            } else {
                if (cw + ncol > d.column_witness.size()) return "WitnessTooShort";
                for (size_t c = 0; c < ncol; c++) vals[c] = d.column_witness[cw++];
            }
            total.push_back({node, host_hash_node(have_last ? &l : nullptr, have_last ? &r : nullptr, vals.data(), ncol, node_conv)});
        }
        last = total; have_last = true;
    }
    if (hw != d.hash_witness.size()) return "WitnessTooLong";
    if (qv != queried_values.size()) return "TooManyQueriedValues";
    if (cw != d.column_witness.size()) return "WitnessTooLong";
    if (last.size() != 1 || !(last[0].second == root)) return "RootMismatch";
    return "";
}

struct PointLessV {
    bool operator()(const PtQ& a, const PtQ& b) const {
        u32 av[8] = {a.x.a.a, a.x.a.b, a.x.b.a, a.x.b.b, a.y.a.a, a.y.a.b, a.y.b.a, a.y.b.b};
        u32 bv[8] = {b.x.a.a, b.x.a.b, b.x.b.a, b.x.b.b, b.y.a.a, b.y.a.b, b.y.b.a, b.y.b.b};
        for (int i = 0; i < 8; i++) if (av[i] != bv[i]) return av[i] < bv[i];
        return false;
    }
};

// PointEvaluator over the sampled mask values (shared with the prover's final sanity check)
struct HostPointEval : LogupState<HostPointEval, Fq> {
    typedef Fq F;
    Q31 preproc; const std::vector<Q31>* tvals; const std::vector<Q31>* ivals; int ti = 0, ii = 0;
    Q31 denom_inverse, random_coeff; Q31* acc;
    int cur_pos = 0;   // position of mask offset 0 in a last logUp column's sampled pair (Conventions::logup_mask_order)
    Fq is_first() { return {preproc}; }
    Fq trace() { return {tvals[ti++].at(0)}; }
    Fq cst(u32 k) { return {q_from_m(k)}; }
    static Q31 combine(const std::vector<Q31>* v, int off) {   // SecureField::from_partial_evals
        Q31 r = v[0].at(off);
        r = q_add(r, q_mul(v[1].at(off), q_make(0, 1, 0, 0)));
        r = q_add(r, q_mul(v[2].at(off), q_make(0, 0, 1, 0)));
        r = q_add(r, q_mul(v[3].at(off), q_make(0, 0, 0, 1)));
        return r;
    }
    Fq inter_cur() { Fq v{combine(ivals + ii, 0)}; ii += 4; return v; }
    void inter_cur_prev(Fq& cur, Fq& prev) { cur.v = combine(ivals + ii, cur_pos); prev.v = combine(ivals + ii, 1 - cur_pos); ii += 4; }
    void constraint(Fq cv) { *acc = q_add(q_mul(*acc, random_coeff), q_mul(denom_inverse, cv.v)); }
};
inline void host_point_eval(int comp, HostPointEval& pe, const Lookups& el) {
    switch (comp) {
        case C_MEMORY: air_eval<C_MEMORY>(pe, el); break;
        case C_INSTRUCTION: air_eval<C_INSTRUCTION>(pe, el); break;
        case C_PROGRAM: air_eval<C_PROGRAM>(pe, el); break;
        case C_PROCESSOR: air_eval<C_PROCESSOR>(pe, el); break;
        case C_JNZ: air_eval<C_JNZ>(pe, el); break;
        case C_JZ: air_eval<C_JZ>(pe, el); break;
        case C_INPUT: air_eval<C_INPUT>(pe, el); break;
        case C_LEFT: air_eval<C_LEFT>(pe, el); break;
        case C_MINUS: air_eval<C_MINUS>(pe, el); break;
        case C_OUTPUT: air_eval<C_OUTPUT>(pe, el); break;
        case C_PLUS: air_eval<C_PLUS>(pe, el); break;
        case C_RIGHT: air_eval<C_RIGHT>(pe, el); break;
        default: air_eval<C_EOE>(pe, el); break;
    }
}
// Components::eval_composition_polynomial_at_point
inline Q31 eval_composition_at_point(const u32* log_sizes, const Q31* claimed_sums, u32 log_max_rows, const Lookups& el, PtQ oods,
                                     const std::vector<std::vector<std::vector<Q31>>>& sv, Q31 random_coeff, const Conventions& cv = Conventions()) {
    Q31 acc = q_zero();
    size_t mo = 0, io = 0;
    for (int k = 0; k < N_COMPONENTS; k++) {
        HostPointEval pe;
        pe.preproc = sv.at(0).at(log_max_rows - log_sizes[k]).at(0);
        pe.tvals = &sv.at(1).at(mo);
        pe.ivals = &sv.at(2).at(io);
        pe.denom_inverse = q_inv(coset_vanishing_q(log_sizes[k], oods));
        pe.random_coeff = random_coeff; pe.acc = &acc; pe.total_sum = claimed_sums[k]; pe.cur_pos = cv.logup_mask_order == 1 ? 1 : 0;
        host_point_eval(k, pe, el);
        mo += n_main_cols(k); io += 4 * n_logup_cols(k);
    }
    return acc;
}

struct QuotientConstantsH { std::vector<PtQ> points; std::vector<std::vector<std::pair<u32, std::array<Q31, 3>>>> entries; std::vector<Q31> batch_coeff; };
inline std::vector<size_t> v_fold_queries(const std::vector<size_t>& q, u32 n) {
    std::vector<size_t> o;
    for (size_t x : q) { size_t y = x >> n; if (o.empty() || o.back() != y) o.push_back(y); }
    return o;
}

// Returns "" when the proof verifies, else the reason.
inline std::string verify_brainfuck(const BrainfuckProof& bp, u32 log_max_rows, const Conventions& cv = Conventions(), VerifierConfig cfg = VerifierConfig()) {
    try {
        const StarkProof& pf = bp.proof;
        if (pf.commitments.size() != 4 || pf.sampled_values.size() != 4 || pf.decommitments.size() != 4 || pf.queried_values.size() != 4) return "InvalidStructure";
        for (int k = 0; k < N_COMPONENTS; k++) if (bp.log_sizes[k] < LOG_N_LANES || bp.log_sizes[k] > log_max_rows) return "InvalidStructure: log_size";
        Channel ch(cv);
        // claim.log_sizes() with the preprocessed tree overwritten by IS_FIRST_LOG_SIZES (mod.rs:118-143)
        std::vector<std::vector<u32>> col_logs(4);
        for (u32 log = log_max_rows; log >= LOG_N_LANES; log--) col_logs[0].push_back(log + cfg.log_blowup);
        u32 comp_log = 0;
        for (int k = 0; k < N_COMPONENTS; k++) {
            for (u32 j = 0; j < n_main_cols(k); j++) col_logs[1].push_back(bp.log_sizes[k] + cfg.log_blowup);
            for (u32 j = 0; j < 4 * n_logup_cols(k); j++) col_logs[2].push_back(bp.log_sizes[k] + cfg.log_blowup);
            comp_log = std::max(comp_log, bp.log_sizes[k] + 1);
        }
        col_logs[3].assign(4, comp_log + cfg.log_blowup);
        ch.mix_root(pf.commitments[0]);
        for (int k = 0; k < N_COMPONENTS; k++) ch.mix_u64(bp.log_sizes[k]);
        ch.mix_root(pf.commitments[1]);
        Lookups el;
        { Q31 z, a; ch.draw_two_felts(z, a); el.memory = make_lookup(z, a); }
        { Q31 z, a; ch.draw_two_felts(z, a); el.instruction = make_lookup(z, a); }
        { Q31 z, a; ch.draw_two_felts(z, a); el.processor = make_lookup(z, a); }
        { Q31 s = q_zero(); for (int k = 0; k < N_COMPONENTS; k++) s = q_add(s, bp.claimed_sums[k]); if (!q_is_zero(s)) return "InvalidLookup: Invalid LogUp sum"; }   // mod.rs:207-227
        for (int k = 0; k < N_COMPONENTS; k++) ch.mix_felts(&bp.claimed_sums[k], 1);
        ch.mix_root(pf.commitments[2]);
        Q31 random_coeff = ch.draw_felt();
        ch.mix_root(pf.commitments[3]);
        PtQ oods;
        { Q31 t = ch.draw_felt(); Q31 t2 = q_mul(t, t); Q31 d = q_inv(q_addm(t2, 1)); oods.x = q_mul(q_sub(q_one(), t2), d); oods.y = q_mul(q_add(t, t), d); }
        // mask points
        std::vector<std::vector<std::vector<PtQ>>> sp(4);
        sp[0].assign(col_logs[0].size(), {});
        for (int k = 0; k < N_COMPONENTS; k++) sp[0][log_max_rows - bp.log_sizes[k]] = {oods};
        for (int k = 0; k < N_COMPONENTS; k++) {
            for (u32 j = 0; j < n_main_cols(k); j++) sp[1].push_back({oods});
            PtQ prev = pq_add(oods, pq_neg(to_q(index_to_point(subgroup_gen(bp.log_sizes[k])))));
            u32 ni = 4 * n_logup_cols(k);
            for (u32 j = 0; j < ni; j++) {
                if (j + 4 >= ni) { if (cv.logup_mask_order == 1) sp[2].push_back({prev, oods}); else sp[2].push_back({oods, prev}); }
                else sp[2].push_back({oods});
            }
        }
        sp[3].assign(4, {oods});
        for (int t = 0; t < 4; t++) {
            if (pf.sampled_values[t].size() != sp[t].size()) return "InvalidStructure: sampled_values";
            for (size_t c = 0; c < sp[t].size(); c++) if (pf.sampled_values[t][c].size() != sp[t][c].size()) return "InvalidStructure: sampled_values";
        }
        {
            std::vector<Q31> ce[4] = {pf.sampled_values[3][0], pf.sampled_values[3][1], pf.sampled_values[3][2], pf.sampled_values[3][3]};
            Q31 want = eval_composition_at_point(bp.log_sizes, bp.claimed_sums, log_max_rows, el, oods, pf.sampled_values, random_coeff, cv);
            if (!q_eq(HostPointEval::combine(ce, 0), want)) return "OodsNotMatching";
        }
        { std::vector<Q31> flat; for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& v : c) flat.push_back(v); ch.mix_felts(flat.data(), flat.size()); }
        Q31 q_coeff = ch.draw_felt();
        std::set<u32, std::greater<u32>> logs_set;
        for (auto& t : col_logs) for (u32 l : t) logs_set.insert(l);
        std::vector<u32> dom_logs(logs_set.begin(), logs_set.end());
        // FriVerifier::commit
        const FriProof& fp = pf.fri_proof;
        ch.mix_root(fp.first_layer.commitment);
        Q31 first_alpha = ch.draw_felt();
        u32 layer_bound = dom_logs[0] - cfg.log_blowup - 1;
        std::vector<Q31> alphas;
        for (auto& lp : fp.inner_layers) { ch.mix_root(lp.commitment); alphas.push_back(ch.draw_felt()); if (layer_bound == 0) return "InvalidNumFriLayers"; layer_bound--; }
        if (layer_bound != cfg.log_last_layer_degree_bound) return "InvalidNumFriLayers";
        if (fp.last_layer_coeffs.size() > (size_t(1) << cfg.log_last_layer_degree_bound)) return "LastLayerDegreeInvalid";
        // LinePoly::eval_at_point folds the coefficients over log_size doublings and asserts len == 2^log_size (stwo utils::fold): a proof
        // whose log_size does not match its coefficient count makes the reference's verifier panic
        if (fp.last_layer_log_size > 31 || fp.last_layer_coeffs.size() != (size_t(1) << fp.last_layer_log_size)) return "LastLayerDegreeInvalid";
        ch.mix_felts(fp.last_layer_coeffs.data(), fp.last_layer_coeffs.size());
        ch.mix_u64(pf.proof_of_work);
        if (ch.trailing_zeros() < cfg.pow_bits) return "ProofOfWork";
        u32 max_log = dom_logs[0];
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
        for (u32 l : dom_logs) positions_by_log[l] = v_fold_queries(queries, max_log - l);
        for (int t = 0; t < 4; t++) { std::string e = merkle_verify(pf.commitments[t], col_logs[t], positions_by_log, pf.queried_values[t], pf.decommitments[t], cv); if (!e.empty()) return "MerkleVerification tree " + std::to_string(t) + ": " + e; }
        // fri_answers: per LDE size (descending), the quotient value at every query position
        struct Flat { int t; size_t c; u32 log; };
        std::vector<Flat> flat;
        for (int t = 0; t < 4; t++) for (size_t c = 0; c < sp[t].size(); c++) flat.push_back({t, c, col_logs[t][c]});
        std::stable_sort(flat.begin(), flat.end(), [](const Flat& a, const Flat& b) { return a.log > b.log; });
        size_t qv_pos[4] = {0, 0, 0, 0};
        std::vector<std::vector<Q31>> answers;
        for (size_t i = 0; i < flat.size();) {
            size_t j = i; u32 log = flat[i].log;
            while (j < flat.size() && flat[j].log == log) j++;
            std::map<PtQ, std::vector<std::pair<u32, Q31>>, PointLessV> by_point;
            for (size_t k = i; k < j; k++) for (size_t s = 0; s < sp[flat[k].t][flat[k].c].size(); s++)
                by_point[sp[flat[k].t][flat[k].c][s]].push_back({(u32)(k - i), pf.sampled_values[flat[k].t][flat[k].c][s]});
            size_t ncols[4] = {0, 0, 0, 0};
            for (int t = 0; t < 4; t++) for (u32 l : col_logs[t]) if (l == log) ncols[t]++;
            std::vector<Q31> ans;
            for (size_t qpos : positions_by_log[log]) {
                PtM dp = canonic_domain_at(log, bit_rev((u32)qpos, log));
                std::vector<u32> vals;
                for (int t = 0; t < 4; t++) for (size_t k = 0; k < ncols[t]; k++) { if (qv_pos[t] >= pf.queried_values[t].size()) return "InvalidStructure: queried_values"; vals.push_back(pf.queried_values[t][qv_pos[t]++]); }
                Q31 row = q_zero();
                for (auto& kv : by_point) {
                    const PtQ& pt = kv.first;
                    Q31 alpha = q_one(), num = q_zero();
                    Q31 cc = q_sub(q_conj(pt.y), pt.y);
                    for (auto& cv : kv.second) {
                        alpha = q_mul(alpha, q_coeff);
                        Q31 a = q_sub(q_conj(cv.second), cv.second);
                        Q31 b = q_sub(q_mul(cv.second, cc), q_mul(a, pt.y));
                        Q31 value = q_mulm(q_mul(alpha, cc), vals.at(cv.first));
                        Q31 linear = q_add(q_mulm(q_mul(alpha, a), dp.y), q_mul(alpha, b));
                        num = q_add(num, q_sub(value, linear));
                    }
                    C31 dx = pt.x.a; dx.a = m_sub(dx.a, dp.x);
                    C31 dy = pt.y.a; dy.a = m_sub(dy.a, dp.y);
                    C31 den = c_sub(c_mul(dx, pt.y.b), c_mul(dy, pt.x.b));
                    row = q_add(q_mul(row, q_pow(q_coeff, kv.second.size())), q_mulc(num, c_inv(den)));
                }
                ans.push_back(row);
            }
            answers.push_back(ans);
            i = j;
        }
        if (answers.size() != dom_logs.size()) return "InvalidStructure: fri answers";
        // FriVerifier::decommit
        struct Sparse { std::vector<std::array<Q31, 2>> evals; std::vector<size_t> starts; };
        auto rebuild = [&](const std::vector<size_t>& q, const std::vector<Q31>& qevals, const std::vector<Q31>& wit, size_t& wi, std::vector<size_t>& positions, Sparse& out) -> bool {
            size_t i = 0, ei = 0;
            while (i < q.size()) {
                size_t j = i; while (j < q.size() && (q[j] >> 1) == (q[i] >> 1)) j++;
                size_t start = (q[i] >> 1) << 1, qi2 = i;
                std::array<Q31, 2> ev;
                for (size_t pos = start; pos < start + 2; pos++) {
                    positions.push_back(pos);
                    if (qi2 < j && q[qi2] == pos) { qi2++; if (ei >= qevals.size()) return false; ev[pos - start] = qevals[ei++]; }
                    else { if (wi >= wit.size()) return false; ev[pos - start] = wit[wi++]; }
                }
                out.evals.push_back(ev); out.starts.push_back(start);
                i = j;
            }
            return true;
        };
        auto words_of = [](const Sparse& s) { std::vector<u32> v; for (auto& ev : s.evals) for (auto& q : ev) { v.push_back(q.a.a); v.push_back(q.a.b); v.push_back(q.b.a); v.push_back(q.b.b); } return v; };
        std::vector<Sparse> first_sparse(dom_logs.size());
        {
            size_t wi = 0; std::map<u32, std::vector<size_t>> dpos; std::vector<u32> dvals, mlogs;
            for (size_t k = 0; k < dom_logs.size(); k++) {
                std::vector<size_t> pos;
                if (!rebuild(v_fold_queries(queries, max_log - dom_logs[k]), answers[k], fp.first_layer.fri_witness, wi, pos, first_sparse[k])) return "FirstLayerEvaluationsInvalid";
                dpos[dom_logs[k]] = pos;
                auto w = words_of(first_sparse[k]); dvals.insert(dvals.end(), w.begin(), w.end());
                for (int c = 0; c < 4; c++) mlogs.push_back(dom_logs[k]);
            }
            if (wi != fp.first_layer.fri_witness.size()) return "FirstLayerEvaluationsInvalid";
            std::string e = merkle_verify(fp.first_layer.commitment, mlogs, dpos, dvals, fp.first_layer.decommitment, cv);
            if (!e.empty()) return "FirstLayerCommitmentInvalid: " + e;
        }
        auto lq = v_fold_queries(queries, 1);
        std::vector<Q31> lev(lq.size(), q_zero());
        size_t fk = 0; Q31 prev_alpha = first_alpha; u32 line_log = max_log - 1;
        for (size_t li = 0; li < fp.inner_layers.size(); li++) {
            while (fk < dom_logs.size() && dom_logs[fk] - 1 == line_log) {
                if (first_sparse[fk].evals.size() != lev.size()) return "InvalidStructure: sparse evals";
                Q31 a2 = q_mul(prev_alpha, prev_alpha);
                for (size_t s = 0; s < lev.size(); s++) {
                    PtM p = canonic_domain_at(dom_logs[fk], bit_rev((u32)first_sparse[fk].starts[s], dom_logs[fk]));
                    Q31 fpv = first_sparse[fk].evals[s][0], fnv = first_sparse[fk].evals[s][1];
                    Q31 f0 = q_add(fpv, fnv), f1 = q_mulm(q_sub(fpv, fnv), m_inv(p.y));
                    lev[s] = q_add(q_mul(lev[s], a2), q_add(f0, q_mul(prev_alpha, f1)));
                }
                fk++;
            }
            const FriLayerProof& lp = fp.inner_layers[li];
            size_t wi = 0; std::vector<size_t> pos; Sparse sps;
            if (!rebuild(lq, lev, lp.fri_witness, wi, pos, sps) || wi != lp.fri_witness.size()) return "InnerLayerEvaluationsInvalid";
            std::map<u32, std::vector<size_t>> dpos; dpos[line_log] = pos;
            std::string e = merkle_verify(lp.commitment, std::vector<u32>(4, line_log), dpos, words_of(sps), lp.decommitment, cv);
            if (!e.empty()) return "InnerLayerCommitmentInvalid: " + e;
            std::vector<Q31> nev;
            for (size_t s = 0; s < sps.evals.size(); s++) {
                // LineDomain(Coset::half_odds(line_log)).at(bit_reverse(start)): x-coordinate
                u32 idx = subgroup_gen(line_log + 2) + subgroup_gen(line_log) * bit_rev((u32)sps.starts[s], line_log);
                u32 x = index_to_point(idx).x;
                Q31 fx = sps.evals[s][0], fn = sps.evals[s][1];
                Q31 f0 = q_add(fx, fn), f1 = q_mulm(q_sub(fx, fn), m_inv(x));
                nev.push_back(q_add(f0, q_mul(alphas[li], f1)));
            }
            lq = v_fold_queries(lq, 1); lev = nev; prev_alpha = alphas[li]; line_log--;
        }
        if (fk != dom_logs.size()) return "InvalidStructure: unconsumed first-layer columns";
        if (cfg.log_last_layer_degree_bound != 0) return "unsupported last layer bound";
        for (size_t i = 0; i < lq.size(); i++) {
            Q31 expect = fp.last_layer_coeffs.empty() ? q_zero() : fp.last_layer_coeffs[0];
            if (!q_eq(lev.at(i), expect)) return "LastLayerEvaluationsInvalid";
        }
        return "";
    } catch (const std::exception& e) { return std::string("InvalidStructure: ") + e.what(); }
}

}  // namespace bf
