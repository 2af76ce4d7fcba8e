// Host-side preparation of the FRI-quotient kernel arguments: ColumnSampleBatch::new_vec + quotient_constants of stwo's
// `core/pcs/quotients.rs` (called from `prover::prove`, reference mod.rs:732). A few hundred QM31 operations per proof; the per-row
// work is k_quotients (quotient.hip).
#pragma once
#include "circle.h"
#include "../kernels.h"
#include <map>
#include <algorithm>
#include <stdexcept>
#include <vector>

namespace bf {

// Order of BTreeMap<CirclePoint<SecureField>, _>: derived Ord, x before y, coordinates in (a.a, a.b, b.a, b.b) order.
struct PointLess {
    bool operator()(const PtQ& a, const PtQ& b) const {
        u32 av[8] = {a.x.a.a, a.x.a.b, a.x.b.a, a.x.b.b, a.y.a.a, a.y.a.b, a.y.b.a, a.y.b.b};
        u32 bv[8] = {b.x.a.a, b.x.a.b, b.x.b.a, b.x.b.b, b.y.a.a, b.y.a.b, b.y.b.a, b.y.b.b};
        for (int i = 0; i < 8; i++) if (av[i] != bv[i]) return av[i] < bv[i];
        return false;
    }
};

struct ColumnSample { PtQ point; Q31 value; };

// samples[k] = the (point, value) samples of column k of one size group, in mask order.
inline void build_quotient_batches(const std::vector<std::vector<ColumnSample>>& samples, Q31 random_coeff,
                                   std::vector<QuotientBatch>& batches, std::vector<QuotientEntry>& entries) {
    const size_t b0 = batches.size();
    std::map<PtQ, std::vector<std::pair<u32, Q31>>, PointLess> by_point;
    for (size_t k = 0; k < samples.size(); k++)
        for (auto& s : samples[k]) by_point[s.point].push_back({(u32)k, s.value});
    // Weights. Within a batch column k carries alpha^(k+1) (alpha restarts per batch), and the row value is the Horner sum over the batches,
    // acc = acc * alpha^(n_b) + term_b, i.e. sum_b term_b * w_b with w_b = the product of the LATER batches' coefficients = alpha^(columns of the
    // later batches). term_b = (sum_k c_k f_k - (A y + B)) / den_b is linear in (c_k, A, B), so w_b is folded into them — and since w_b is itself
    // a power of alpha, column k of batch b simply gets alpha^(k + 1 + columns of the later batches): ONE run of powers for the whole size group,
    // no separate weight products (r04; exact field arithmetic, so the constants are the same values as before).
    size_t total = 0;
    for (auto& kv : by_point) total += kv.second.size();
    std::vector<Q31> power(total + 1);
    power[0] = q_one();
    for (size_t i = 1; i <= total; i++) power[i] = q_mul(power[i - 1], random_coeff);
    // a Q31 times a "pure u" element (0, d): (x.a + x.b u) (d u) = x.b d (2 + i) + x.a d u — half the products of a general q_mul
    auto mul_pure = [](const Q31& x, const C31& d) { return Q31{c_mulR(c_mul(x.b, d)), c_mul(x.a, d)}; };
    size_t later = total;
    for (auto& kv : by_point) {
        const PtQ& pt = kv.first;
        later -= kv.second.size();                 // columns of the batches after this one
        QuotientBatch qb{};
        qb.prx = pt.x.a; qb.pry = pt.y.a; qb.pix = pt.x.b; qb.piy = pt.y.b;
        qb.kden = c_sub(c_mul(qb.prx, qb.piy), c_mul(qb.pry, qb.pix));   // constant part of the row denominators
        qb.a_sum = q_zero(); qb.b_sum = q_zero();
        // complex_conjugate_line_coeffs: a = conj(v) - v, c = conj(P.y) - P.y, b = v*c - a*P.y; all scaled by the column's weight.
        // conj flips the u half, so a = (0, -2 v.b) and c = (0, -2 P.y.b) are pure-u elements (c is the same for every column of the batch).
        const C31 cc = c_neg(c_add(pt.y.b, pt.y.b));
        size_t k = 0;
        for (auto& cv : kv.second) {
            const Q31& wgt = power[++k + later];
            const C31 a = c_neg(c_add(cv.second.b, cv.second.b));
            const Q31 b = q_sub(mul_pure(cv.second, cc), mul_pure(pt.y, a));
            qb.a_sum = q_add(qb.a_sum, mul_pure(wgt, a));
            qb.b_sum = q_add(qb.b_sum, q_mul(wgt, b));
            QuotientEntry qe{}; qe.c = mul_pure(wgt, cc); qe.col = cv.first;
            entries.push_back(qe);
        }
        qb.batch_coeff = power[kv.second.size()];      // kept for reference: already folded into the weights
        qb.n_cols = (u32)kv.second.size();
        batches.push_back(qb);
    }
    (void)b0;
}

// The prover's form of the same: the sample points of a proof are a handful (the out-of-domain point and its shifts by the components' trace
// steps), so a sample names its point by INDEX into `points` and a column has at most two of them — no map, no per-column vectors, one pass
// (this runs with the GPU idle behind the sampled values: r04). Same batches, same entry order, same constants as build_quotient_batches.
struct ColSamples { u32 n; u32 point[2]; Q31 value[2]; };
inline void build_quotient_batches_indexed(const ColSamples* cols, size_t n_cols, const std::vector<PtQ>& points, Q31 random_coeff,
                                           std::vector<QuotientBatch>& batches, std::vector<QuotientEntry>& entries) {
    constexpr u32 MAXP = 64;
    if (points.size() > MAXP) throw std::runtime_error("quotients: too many sample points");
    u32 cnt[MAXP] = {0};
    size_t total = 0;
    for (size_t k = 0; k < n_cols; k++) for (u32 s = 0; s < cols[k].n; s++) { cnt[cols[k].point[s]]++; total++; }
    // batches in BTreeMap order of the points that occur
    u32 order[MAXP], n_pts = 0;
    for (u32 p = 0; p < points.size(); p++) if (cnt[p]) order[n_pts++] = p;
    std::sort(order, order + n_pts, [&](u32 a, u32 b) { return PointLess()(points[a], points[b]); });
    // two indices may name equal points (two components of one size): they form ONE batch, as in the map
    u32 batch_of[MAXP], n_batches = 0, start[MAXP + 1], later[MAXP];
    u32 size[MAXP] = {0};
    for (u32 i = 0; i < n_pts; i++) {
        if (i && !PointLess()(points[order[i - 1]], points[order[i]])) batch_of[order[i]] = n_batches - 1;
        else batch_of[order[i]] = n_batches++;
        size[batch_of[order[i]]] += cnt[order[i]];
    }
    start[0] = 0;
    for (u32 b = 0; b < n_batches; b++) start[b + 1] = start[b] + size[b];
    for (u32 b = 0; b < n_batches; b++) later[b] = (u32)total - start[b + 1];
    std::vector<Q31> power(total + 1);
    power[0] = q_one();
    for (size_t i = 1; i <= total; i++) power[i] = q_mul(power[i - 1], random_coeff);
    auto mul_pure = [](const Q31& x, const C31& d) { return Q31{c_mulR(c_mul(x.b, d)), c_mul(x.a, d)}; };
    const size_t b0 = batches.size(), e0 = entries.size();
    batches.resize(b0 + n_batches);
    entries.resize(e0 + total);
    C31 cc[MAXP];
    for (u32 i = 0; i < n_pts; i++) {
        const u32 b = batch_of[order[i]];
        const PtQ& pt = points[order[i]];
        QuotientBatch& qb = batches[b0 + b];
        qb = QuotientBatch{};
        qb.prx = pt.x.a; qb.pry = pt.y.a; qb.pix = pt.x.b; qb.piy = pt.y.b;
        qb.kden = c_sub(c_mul(qb.prx, qb.piy), c_mul(qb.pry, qb.pix));
        qb.a_sum = q_zero(); qb.b_sum = q_zero();
        qb.n_cols = size[b]; qb.batch_coeff = power[size[b]];
        cc[b] = c_neg(c_add(pt.y.b, pt.y.b));
    }
    u32 fill[MAXP] = {0};
    for (size_t k = 0; k < n_cols; k++)
        for (u32 s = 0; s < cols[k].n; s++) {
            const u32 p = cols[k].point[s], b = batch_of[p];
            const u32 pos = fill[b]++;                                   // columns in ascending order within a batch, as the map's vectors
            const Q31& wgt = power[pos + 1 + later[b]];
            const Q31& v = cols[k].value[s];
            const C31 a = c_neg(c_add(v.b, v.b));
            const Q31 bb = q_sub(mul_pure(v, cc[b]), mul_pure(points[p].y, a));
            QuotientBatch& qb = batches[b0 + b];
            qb.a_sum = q_add(qb.a_sum, mul_pure(wgt, a));
            qb.b_sum = q_add(qb.b_sum, q_mul(wgt, bb));
            QuotientEntry& qe = entries[e0 + start[b] + pos];
            qe = QuotientEntry{}; qe.c = mul_pure(wgt, cc[b]); qe.col = (u32)k;
        }
}

}  // namespace bf
