// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). PARITY UNPINNED.
// Restates stwo@31e8dbc `core/vcs/{prover,verifier,blake2_merkle}.rs`: mixed-degree Merkle tree; node(i) absorbs
// left || right (when a deeper layer exists) and then the LE-u32 value of each column-of-this-size at row i.
// How it absorbs them is `Conventions::merkle_node_hash` (blake2s.h): the default is the published `Blake2sMerkleHasher::hash_node` of this
// period — state = 0; state = compress(state, left||right, 0,0,0,0); then compress(state, chunk, 0,0,0,0) for the column words in
// zero-padded chunks of 16 (rem = 15 - ((len + 15) % 16)); no parameter block, no byte counter, no finalisation flag — the
// alternative is the RFC 7693 hash of the same byte string.
// Reference call sites: tree_builder.commit(channel) crates/brainfuck_prover/src/brainfuck_air/mod.rs:500,583,723.
#pragma once
#include "blake2s.h"
#include "simd_port.h"
#include <map>
#include <string>
#include <algorithm>

namespace orc {

// Blake2sMerkleHasher::hash_node
static inline Hash32 hash_node(const Hash32* left, const Hash32* right, const u32* vals, size_t n_vals) {
    if (conventions().merkle_channel == 1) {
        // Poseidon252MerkleHasher::hash_node: values = [left, right]? ++ one felt per block of 8 column values (zero padded),
        // word = word * 2^31 + x; poseidon_hash_many(values)
        std::vector<Felt> values;
        if (left) { values.push_back(Felt::from_le_bytes(left->b)); values.push_back(Felt::from_le_bytes(right->b)); }
        for (size_t o = 0; o < n_vals; o += 8) {
            Felt word = Felt::raw(0, 0, 0, 0);
            for (size_t k = 0; k < 8; k++) word = fold_m31(word, o + k < n_vals ? vals[o + k] : 0u);
            values.push_back(word);
        }
        Hash32 out; poseidon_hash_many(values).to_le_bytes(out.b); return out;
    }
    if (conventions().merkle_node_hash == 0) {
        u32 st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[16];
        if (left) { memcpy(m, left->b, 32); memcpy(m + 8, right->b, 32); blake2s_compress(st, m, 0, 0, 0, 0); }
        for (size_t o = 0; o < n_vals; o += 16) {
            size_t take = std::min<size_t>(16, n_vals - o);
            memset(m, 0, sizeof m); memcpy(m, vals + o, 4 * take);
            blake2s_compress(st, m, 0, 0, 0, 0);
        }
        Hash32 out; memcpy(out.b, st, 32); return out;
    }
    Blake2s s;
    if (left) { s.update(left->b, 32); s.update(right->b, 32); }
    if (n_vals) s.update(vals, 4 * n_vals);
    return s.finalize();
}

struct ColRef { const u32* data; u32 log_size; };

struct MerkleDecommitment { std::vector<Hash32> hash_witness; std::vector<u32> column_witness; };

struct MerkleProver {
    // layers[k] = hashes of the layer of log size k (layers[0] = root layer)
    std::vector<std::vector<Hash32>> layers;

    static std::vector<ColRef> sort_cols(std::vector<ColRef> cols) {
        std::stable_sort(cols.begin(), cols.end(), [](const ColRef& a, const ColRef& b) { return a.log_size > b.log_size; });
        return cols;
    }
    // MerkleProver::commit
    static MerkleProver commit(const std::vector<ColRef>& columns) {
        MerkleProver mp;
        if (columns.empty()) { mp.layers.push_back({hash_node(nullptr, nullptr, nullptr, 0)}); return mp; }
        auto cols = sort_cols(columns);
        u32 max_log = cols[0].log_size;
        mp.layers.resize(max_log + 1);
        size_t ci = 0;
        for (int log = (int)max_log; log >= 0; log--) {
            std::vector<ColRef> lc;
            while (ci < cols.size() && cols[ci].log_size == (u32)log) lc.push_back(cols[ci++]);
            size_t n = size_t(1) << log;
            auto& layer = mp.layers[log];
            layer.resize(n);
            const std::vector<Hash32>* prev = (log < (int)max_log) ? &mp.layers[log + 1] : nullptr;
            // SIMD mode of the port (orc_set_simd: bench.py's cpu_baseline): 16 nodes per vector compression, same bytes
            if (simd::enabled() && n >= 16 && conventions().merkle_channel == 0 && conventions().merkle_node_hash == 0) {
                std::vector<const u32*> ptrs(lc.size());
                for (size_t c = 0; c < lc.size(); c++) ptrs[c] = lc[c].data;
                simd::merkle_layer_blake2s(prev ? (*prev)[0].b : nullptr, ptrs.data(), ptrs.size(), n, layer[0].b);
                continue;
            }
#pragma omp parallel for schedule(static) if (n >= 4096)
            for (size_t i = 0; i < n; i++) {
                u32 vals_buf[64]; std::vector<u32> big; u32* vals = vals_buf;
                if (lc.size() > 64) { big.resize(lc.size()); vals = big.data(); }
                for (size_t c = 0; c < lc.size(); c++) vals[c] = lc[c].data[i];
                layer[i] = hash_node(prev ? &(*prev)[2 * i] : nullptr, prev ? &(*prev)[2 * i + 1] : nullptr, vals, lc.size());
            }
        }
        return mp;
    }
    Hash32 root() const { return layers[0][0]; }

    // MerkleProver::decommit: returns (queried_values, decommitment)
    std::pair<std::vector<u32>, MerkleDecommitment> decommit(const std::map<u32, std::vector<size_t>>& queries_per_log_size,
                                                             const std::vector<ColRef>& columns) const {
        std::vector<u32> queried_values;
        MerkleDecommitment d;
        auto cols = sort_cols(columns);
        size_t ci = 0;
        std::vector<size_t> last_layer_queries;
        for (int log = (int)layers.size() - 1; log >= 0; log--) {
            std::vector<ColRef> lc;
            while (ci < cols.size() && cols[ci].log_size == (u32)log) lc.push_back(cols[ci++]);
            const std::vector<Hash32>* prev_hashes = (log + 1 < (int)layers.size()) ? &layers[log + 1] : nullptr;
            std::vector<size_t> layer_total;
            static const std::vector<size_t> empty;
            auto it = queries_per_log_size.find((u32)log);
            const std::vector<size_t>& colq = it == queries_per_log_size.end() ? empty : it->second;
            size_t pi = 0, qi = 0;
            while (pi < last_layer_queries.size() || qi < colq.size()) {
                size_t node;
                if (pi < last_layer_queries.size() && qi < colq.size()) node = std::min(last_layer_queries[pi] / 2, colq[qi]);
                else if (pi < last_layer_queries.size()) node = last_layer_queries[pi] / 2;
                else node = colq[qi];
                if (prev_hashes) {
                    if (pi < last_layer_queries.size() && last_layer_queries[pi] == 2 * node) pi++;
                    else d.hash_witness.push_back((*prev_hashes)[2 * node]);
                    if (pi < last_layer_queries.size() && last_layer_queries[pi] == 2 * node + 1) pi++;
                    else d.hash_witness.push_back((*prev_hashes)[2 * node + 1]);
                }
                if (qi < colq.size() && colq[qi] == node) { qi++; for (auto& c : lc) queried_values.push_back(c.data[node]); }
                else for (auto& c : lc) d.column_witness.push_back(c.data[node]);
                layer_total.push_back(node);
            }
            last_layer_queries = layer_total;
        }
        return {queried_values, d};
    }
};

// MerkleVerifier::verify. Returns empty string on success, else the error name.
struct MerkleVerifier {
    Hash32 root;
    std::vector<u32> column_log_sizes;
    std::map<u32, size_t> n_columns_per_log_size;
    MerkleVerifier() {}
    MerkleVerifier(Hash32 r, std::vector<u32> ls) : root(r), column_log_sizes(std::move(ls)) {
        for (u32 l : column_log_sizes) n_columns_per_log_size[l]++;
    }
    std::string verify(const std::map<u32, std::vector<size_t>>& queries_per_log_size, const std::vector<u32>& queried_values,
                       const MerkleDecommitment& d) const {
        if (column_log_sizes.empty()) return "";
        u32 max_log = *std::max_element(column_log_sizes.begin(), column_log_sizes.end());
        size_t qv = 0, hw = 0, cw = 0;
        std::vector<std::pair<size_t, Hash32>> last;
        bool have_last = false;
        for (int log = (int)max_log; log >= 0; log--) {
            auto nit = n_columns_per_log_size.find((u32)log);
            size_t ncol = nit == n_columns_per_log_size.end() ? 0 : nit->second;
            std::vector<std::pair<size_t, Hash32>> total;
            static const std::vector<size_t> empty;
            auto it = queries_per_log_size.find((u32)log);
            const std::vector<size_t>& colq = it == queries_per_log_size.end() ? empty : it->second;
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
                total.push_back({node, hash_node(have_last ? &l : nullptr, have_last ? &r : nullptr, vals.data(), ncol)});
            }
            last = total;
            have_last = true;
        }
        if (hw != d.hash_witness.size()) return "WitnessTooLong";
        if (qv != queried_values.size()) return "TooManyQueriedValues";
        if (cw != d.column_witness.size()) return "WitnessTooLong";
        if (last.size() != 1) return "RootMismatch";
        if (last[0].second != root) return "RootMismatch";
        return "";
    }
};

}  // namespace orc
