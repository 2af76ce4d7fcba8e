// Proof object and its serde_json-compatible (compact) serialisation — the "proof bytes" of the metric.
// Shapes follow the derives at crates/brainfuck_prover/src/brainfuck_air/mod.rs:71-99 (BrainfuckProof, BrainfuckClaim),
// :170-185 (BrainfuckInteractionClaim), components/mod.rs:71-93 (Claim{log_size,_marker}, InteractionClaim{claimed_sum}) and
// stwo's StarkProof(CommitmentSchemeProof{commitments, sampled_values, decommitments, queried_values, proof_of_work, fri_proof}).
#pragma once
#include "hash.h"
#include "tables.h"
#include <string>

namespace bf {

struct MerkleDecommitment { std::vector<Hash32> hash_witness; std::vector<u32> column_witness; };
struct FriLayerProof { std::vector<Q31> fri_witness; MerkleDecommitment decommitment; Hash32 commitment; };
struct FriProof { FriLayerProof first_layer; std::vector<FriLayerProof> inner_layers; std::vector<Q31> last_layer_coeffs; u32 last_layer_log_size = 0; };
struct StarkProof {
    std::vector<Hash32> commitments;
    std::vector<std::vector<std::vector<Q31>>> sampled_values;
    std::vector<MerkleDecommitment> decommitments;
    std::vector<std::vector<u32>> queried_values;
    u64 proof_of_work = 0;
    FriProof fri_proof;
};
struct BrainfuckProof { u32 log_sizes[N_COMPONENTS]; Q31 claimed_sums[N_COMPONENTS]; StarkProof proof; };

namespace json {
inline void num(std::string& s, u64 v) { char buf[24]; int n = 0; if (!v) buf[n++] = '0'; while (v) { buf[n++] = (char)('0' + v % 10); v /= 10; } while (n) s.push_back(buf[--n]); }
inline void qm31(std::string& s, const Q31& q) { s += "[["; num(s, q.a.a); s += ','; num(s, q.a.b); s += "],["; num(s, q.b.a); s += ','; num(s, q.b.b); s += "]]"; }
// Blake2sHash: array of 32 byte values. FieldElement252 (felt = true, Poseidon252MerkleHasher::Hash): "0x" + minimal lowercase hex, the
// human-readable serde form of starknet-ff 0.3.7 (Cargo.lock:861); h holds the canonical value as 32 little-endian bytes.
inline void hash(std::string& s, const Hash32& h, bool felt) {
    if (felt) {
        static const char* HEX = "0123456789abcdef";
        s += "\"0x";
        int top = 63; while (top > 0 && ((h.b[top / 2] >> (4 * (top & 1))) & 15) == 0) top--;
        for (int i = top; i >= 0; i--) s.push_back(HEX[(h.b[i / 2] >> (4 * (i & 1))) & 15]);
        s += '"';
        return;
    }
    // 32 byte values as decimal numbers: most of a proof's text (a proof holds ~1000 hashes). One table lookup and one append per byte.
    struct ByteText { char t[256][4]; u8 n[256]; ByteText() { for (int v = 0; v < 256; v++) { int k = 0; if (v >= 100) t[v][k++] = (char)('0' + v / 100); if (v >= 10) t[v][k++] = (char)('0' + v / 10 % 10); t[v][k++] = (char)('0' + v % 10); n[v] = (u8)k; } } };
    static const ByteText bt;
    char buf[32 * 4 + 2]; int n = 0;
    buf[n++] = '[';
    for (int i = 0; i < 32; i++) { if (i) buf[n++] = ','; const u8 v = h.b[i]; for (int k = 0; k < bt.n[v]; k++) buf[n++] = bt.t[v][k]; }
    buf[n++] = ']';
    s.append(buf, (size_t)n);
}
template <class T, class Fn> void arr(std::string& s, const std::vector<T>& v, Fn f) { s += '['; for (size_t i = 0; i < v.size(); i++) { if (i) s += ','; f(v[i]); } s += ']'; }
inline void decommitment(std::string& s, const MerkleDecommitment& d, bool felt) {
    s += "{\"hash_witness\":"; arr(s, d.hash_witness, [&](const Hash32& h) { hash(s, h, felt); });
    s += ",\"column_witness\":"; arr(s, d.column_witness, [&](u32 v) { num(s, v); }); s += '}';
}
inline void fri_layer(std::string& s, const FriLayerProof& l, bool felt) {
    s += "{\"fri_witness\":"; arr(s, l.fri_witness, [&](const Q31& q) { qm31(s, q); });
    s += ",\"decommitment\":"; decommitment(s, l.decommitment, felt);
    s += ",\"commitment\":"; hash(s, l.commitment, felt); s += '}';
}
}  // namespace json

// felt_hashes: the proof was made with Poseidon252MerkleChannel (hashes are felt252 values)
inline std::string proof_to_json(const BrainfuckProof& bp, bool felt_hashes = false) {
    using namespace json;
    const bool felt = felt_hashes;
    std::string s;
    s.reserve(1 << 17);
    s += "{\"claim\":{";
    for (int c = 0; c < N_COMPONENTS; c++) { if (c) s += ','; s += '"'; s += COMPONENT_NAMES[c]; s += "\":{\"log_size\":"; num(s, bp.log_sizes[c]); s += ",\"_marker\":null}"; }
    s += "},\"interaction_claim\":{";
    for (int c = 0; c < N_COMPONENTS; c++) { if (c) s += ','; s += '"'; s += COMPONENT_NAMES[c]; s += "\":{\"claimed_sum\":"; qm31(s, bp.claimed_sums[c]); s += '}'; }
    const StarkProof& p = bp.proof;
    s += "},\"proof\":{\"commitments\":"; arr(s, p.commitments, [&](const Hash32& h) { hash(s, h, felt); });
    s += ",\"sampled_values\":";
    arr(s, p.sampled_values, [&](const std::vector<std::vector<Q31>>& t) { arr(s, t, [&](const std::vector<Q31>& c) { arr(s, c, [&](const Q31& q) { qm31(s, q); }); }); });
    s += ",\"decommitments\":"; arr(s, p.decommitments, [&](const MerkleDecommitment& d) { decommitment(s, d, felt); });
    s += ",\"queried_values\":"; arr(s, p.queried_values, [&](const std::vector<u32>& v) { arr(s, v, [&](u32 x) { num(s, x); }); });
    s += ",\"proof_of_work\":"; num(s, p.proof_of_work);
    s += ",\"fri_proof\":{\"first_layer\":"; fri_layer(s, p.fri_proof.first_layer, felt);
    s += ",\"inner_layers\":"; arr(s, p.fri_proof.inner_layers, [&](const FriLayerProof& l) { fri_layer(s, l, felt); });
    s += ",\"last_layer_poly\":{\"coeffs\":"; arr(s, p.fri_proof.last_layer_coeffs, [&](const Q31& q) { qm31(s, q); });
    s += ",\"log_size\":"; num(s, p.fri_proof.last_layer_log_size); s += "}}}}";
    return s;
}

}  // namespace bf
