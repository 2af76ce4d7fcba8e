// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
// serde_json-compatible (compact) writer/reader for BrainfuckProof, following the derive(Serialize) shapes at
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:71-99,170-185 and components/mod.rs:71-93 (Claim{log_size,_marker},
// InteractionClaim{claimed_sum}); the stwo StarkProof/CommitmentSchemeProof/FriProof field order is recalled (UNPINNED).
#pragma once
#include "prover.h"
#include <cstdlib>
#include <memory>

namespace orc {

static const char* const CLAIM_KEYS[N_COMPONENTS] = {"memory", "instruction", "program", "processor", "jump_if_not_zero", "jump_if_zero",
    "input_instruction", "left_instruction", "minus_instruction", "output_instruction", "plus_instruction", "right_instruction", "end_of_execution"};

struct JsonWriter {
    std::string s;
    void num(u64 v) { s += std::to_string(v); }
    void qm31(const QM31& q) { auto a = q.to_u32(); s += "[["; num(a[0]); s += ","; num(a[1]); s += "],["; num(a[2]); s += ","; num(a[3]); s += "]]"; }
    // Blake2sHash: 32 byte values; FieldElement252 (Poseidon252 variant): "0x" + minimal lowercase hex of the canonical value
    void hash(const Hash32& h) {
        if (conventions().merkle_channel == 1) {
            char buf[3]; std::string hex;
            for (int i = 31; i >= 0; i--) { snprintf(buf, sizeof buf, "%02x", h.b[i]); hex += buf; }
            size_t nz = hex.find_first_not_of('0');
            s += "\"0x" + (nz == std::string::npos ? std::string("0") : hex.substr(nz)) + "\"";
            return;
        }
        s += "["; for (int i = 0; i < 32; i++) { if (i) s += ","; num(h.b[i]); } s += "]";
    }
    template <class T, class Fn> void arr(const std::vector<T>& v, Fn f) { s += "["; for (size_t i = 0; i < v.size(); i++) { if (i) s += ","; f(v[i]); } s += "]"; }
    void decommitment(const MerkleDecommitment& d) {
        s += "{\"hash_witness\":"; arr(d.hash_witness, [&](const Hash32& h) { hash(h); });
        s += ",\"column_witness\":"; arr(d.column_witness, [&](u32 v) { num(v); }); s += "}";
    }
    void fri_layer(const FriLayerProof& l) {
        s += "{\"fri_witness\":"; arr(l.fri_witness, [&](const QM31& q) { qm31(q); });
        s += ",\"decommitment\":"; decommitment(l.decommitment);
        s += ",\"commitment\":"; hash(l.commitment); s += "}";
    }
};

static inline std::string proof_to_json(const BrainfuckProof& bp) {
    JsonWriter w;
    w.s += "{\"claim\":{";
    for (int c = 0; c < N_COMPONENTS; c++) { if (c) w.s += ","; w.s += "\""; w.s += CLAIM_KEYS[c]; w.s += "\":{\"log_size\":"; w.num(bp.log_sizes[c]); w.s += ",\"_marker\":null}"; }
    w.s += "},\"interaction_claim\":{";
    for (int c = 0; c < N_COMPONENTS; c++) { if (c) w.s += ","; w.s += "\""; w.s += CLAIM_KEYS[c]; w.s += "\":{\"claimed_sum\":"; w.qm31(bp.claimed_sums[c]); w.s += "}"; }
    const StarkProof& p = bp.proof;
    w.s += "},\"proof\":{\"commitments\":"; w.arr(p.commitments, [&](const Hash32& h) { w.hash(h); });
    w.s += ",\"sampled_values\":";
    w.arr(p.sampled_values, [&](const std::vector<std::vector<QM31>>& t) { w.arr(t, [&](const std::vector<QM31>& c) { w.arr(c, [&](const QM31& q) { w.qm31(q); }); }); });
    w.s += ",\"decommitments\":"; w.arr(p.decommitments, [&](const MerkleDecommitment& d) { w.decommitment(d); });
    w.s += ",\"queried_values\":"; w.arr(p.queried_values, [&](const std::vector<u32>& v) { w.arr(v, [&](u32 x) { w.num(x); }); });
    w.s += ",\"proof_of_work\":"; w.num(p.proof_of_work);
    w.s += ",\"fri_proof\":{\"first_layer\":"; w.fri_layer(p.fri_proof.first_layer);
    w.s += ",\"inner_layers\":"; w.arr(p.fri_proof.inner_layers, [&](const FriLayerProof& l) { w.fri_layer(l); });
    w.s += ",\"last_layer_poly\":{\"coeffs\":"; w.arr(p.fri_proof.last_layer_coeffs, [&](const QM31& q) { w.qm31(q); });
    w.s += ",\"log_size\":"; w.num(p.fri_proof.last_layer_log_size); w.s += "}}}}";
    return w.s;
}

// Minimal JSON value tree (numbers are unsigned integers; null supported).
struct JVal {
    enum Kind { NUM, ARR, OBJ, NUL, STR } kind = NUL;
    u64 num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal& get(const char* k) const { for (auto& kv : obj) if (kv.first == k) return kv.second; throw std::runtime_error(std::string("missing key ") + k); }
};
struct JsonParser {
    const char* p; const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    JVal parse() {
        ws();
        if (p >= end) throw std::runtime_error("json: eof");
        JVal v;
        if (*p == '[') {
            v.kind = JVal::ARR; p++; ws();
            if (p < end && *p == ']') { p++; return v; }
            for (;;) { v.arr.push_back(parse()); ws(); if (p < end && *p == ',') { p++; continue; } if (p < end && *p == ']') { p++; break; } throw std::runtime_error("json: array"); }
        } else if (*p == '{') {
            v.kind = JVal::OBJ; p++; ws();
            if (p < end && *p == '}') { p++; return v; }
            for (;;) {
                ws(); if (p >= end || *p != '"') throw std::runtime_error("json: key");
                p++; const char* s = p; while (p < end && *p != '"') p++; std::string k(s, p); p++;
                ws(); if (p >= end || *p != ':') throw std::runtime_error("json: colon"); p++;
                for (auto& kv : v.obj) if (kv.first == k) throw std::runtime_error("json: duplicate key");   // serde: "duplicate field"
                v.obj.push_back({k, parse()}); ws();
                if (p < end && *p == ',') { p++; continue; } if (p < end && *p == '}') { p++; break; } throw std::runtime_error("json: object");
            }
        } else if (*p == '"') { p++; const char* s0 = p; while (p < end && *p != '"') p++; if (p >= end) throw std::runtime_error("json: string"); v.kind = JVal::STR; v.str.assign(s0, p); p++; }
        else if (*p == 'n') { if (end - p < 4 || std::string(p, p + 4) != "null") throw std::runtime_error("json: null"); p += 4; v.kind = JVal::NUL; }
        else if (*p >= '0' && *p <= '9') {
            // what serde_json takes for a u32 / u64 field: a plain decimal integer below 2^64 without leading zeros, fraction or exponent
            // (a reader that wraps modulo 2^64 would accept 2^64 + v in place of v)
            v.kind = JVal::NUM; u64 x = 0; const char* first = p;
            for (; p < end && *p >= '0' && *p <= '9'; p++) {
                const u64 digit = (u64)(*p - '0');
                if (x > 1844674407370955161ull || (x == 1844674407370955161ull && digit > 5)) throw std::runtime_error("json: integer does not fit 64 bits");
                x = x * 10 + digit;
            }
            if (*first == '0' && p - first > 1) throw std::runtime_error("json: leading zero");
            if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) throw std::runtime_error("json: not an integer");
            v.num = x;
        }
        else throw std::runtime_error("json: unexpected char");
        return v;
    }
};

// a Vec field: serde_json takes a JSON array there and nothing else (an empty list written as null or {} must not pass)
static inline const std::vector<JVal>& j_seq(const JVal& v) { if (v.kind != JVal::ARR) throw std::runtime_error("expected a sequence"); return v.arr; }
// a `u32` field of the reference structs holding a log2 size (serde refuses what does not fit the type; a size above 31 cannot be meant)
static inline u32 j_log_size(const JVal& v) { if (v.kind != JVal::NUM || v.num > 31) throw std::runtime_error("bad log_size"); return (u32)v.num; }
static inline u32 j_m31(const JVal& v) { if (v.kind != JVal::NUM || v.num >= P) throw std::runtime_error("bad M31"); return (u32)v.num; }
static inline QM31 j_qm31(const JVal& v) {
    if (v.kind != JVal::ARR || v.arr.size() != 2 || v.arr[0].kind != JVal::ARR || v.arr[1].kind != JVal::ARR || v.arr[0].arr.size() != 2 || v.arr[1].arr.size() != 2) throw std::runtime_error("bad QM31");
    return QM31::from_u32(j_m31(v.arr[0].arr[0]), j_m31(v.arr[0].arr[1]), j_m31(v.arr[1].arr[0]), j_m31(v.arr[1].arr[1]));
}
static inline Hash32 j_hash(const JVal& v) {
    if (conventions().merkle_channel == 1) {
        if (v.kind != JVal::STR || v.str.size() < 3 || v.str.size() > 66 || v.str.compare(0, 2, "0x") != 0) throw std::runtime_error("bad felt hash");
        Hash32 h; memset(h.b, 0, 32);
        for (size_t i = 0; i + 2 < v.str.size(); i++) {
            char ch = v.str[v.str.size() - 1 - i]; int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : -1;
            if (d < 0) throw std::runtime_error("bad felt hash digit");
            h.b[i / 2] |= (u8)(d << (4 * (i & 1)));
        }
        u64 c[4]; memcpy(c, h.b, 32);
        if (Felt::ge_p(c)) throw std::runtime_error("felt hash out of range");
        return h;
    }
    if (v.kind != JVal::ARR || v.arr.size() != 32) throw std::runtime_error("bad hash");
    Hash32 h; for (int i = 0; i < 32; i++) { if (v.arr[i].kind != JVal::NUM || v.arr[i].num > 255) throw std::runtime_error("bad hash byte"); h.b[i] = (u8)v.arr[i].num; } return h;
}
static inline MerkleDecommitment j_decommitment(const JVal& v) {
    MerkleDecommitment d;
    for (auto& h : j_seq(v.get("hash_witness"))) d.hash_witness.push_back(j_hash(h));
    for (auto& x : j_seq(v.get("column_witness"))) d.column_witness.push_back(j_m31(x));
    return d;
}
static inline FriLayerProof j_fri_layer(const JVal& v) {
    FriLayerProof l;
    for (auto& q : j_seq(v.get("fri_witness"))) l.fri_witness.push_back(j_qm31(q));
    l.decommitment = j_decommitment(v.get("decommitment"));
    l.commitment = j_hash(v.get("commitment"));
    return l;
}
static inline BrainfuckProof proof_from_json(const char* s, size_t len) {
    JsonParser jp{s, s + len};
    JVal root = jp.parse();
    jp.ws();
    if (jp.p != jp.end) throw std::runtime_error("json: bytes after the proof object");   // serde_json::from_slice refuses trailing characters
    BrainfuckProof bp;
    for (int c = 0; c < N_COMPONENTS; c++) {
        bp.log_sizes[c] = j_log_size(root.get("claim").get(CLAIM_KEYS[c]).get("log_size"));
        bp.claimed_sums[c] = j_qm31(root.get("interaction_claim").get(CLAIM_KEYS[c]).get("claimed_sum"));
    }
    const JVal& p = root.get("proof");
    StarkProof& sp = bp.proof;
    for (auto& h : j_seq(p.get("commitments"))) sp.commitments.push_back(j_hash(h));
    for (auto& t : j_seq(p.get("sampled_values"))) {
        std::vector<std::vector<QM31>> tv;
        for (auto& c : j_seq(t)) { std::vector<QM31> cv; for (auto& q : j_seq(c)) cv.push_back(j_qm31(q)); tv.push_back(cv); }
        sp.sampled_values.push_back(tv);
    }
    for (auto& d : j_seq(p.get("decommitments"))) sp.decommitments.push_back(j_decommitment(d));
    for (auto& t : j_seq(p.get("queried_values"))) { std::vector<u32> v; for (auto& x : j_seq(t)) v.push_back(j_m31(x)); sp.queried_values.push_back(v); }
    if (p.get("proof_of_work").kind != JVal::NUM) throw std::runtime_error("bad proof_of_work");
    sp.proof_of_work = p.get("proof_of_work").num;
    const JVal& f = p.get("fri_proof");
    sp.fri_proof.first_layer = j_fri_layer(f.get("first_layer"));
    for (auto& l : j_seq(f.get("inner_layers"))) sp.fri_proof.inner_layers.push_back(j_fri_layer(l));
    for (auto& q : j_seq(f.get("last_layer_poly").get("coeffs"))) sp.fri_proof.last_layer_coeffs.push_back(j_qm31(q));
    sp.fri_proof.last_layer_log_size = j_log_size(f.get("last_layer_poly").get("log_size"));
    return bp;
}

}  // namespace orc
