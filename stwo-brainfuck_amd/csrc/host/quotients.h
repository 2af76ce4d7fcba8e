// Host-side preparation of the FRI-quotient kernel arguments: ColumnSampleBatch::new_vec + quotient_constants of stwo's
// `core/pcs/quotients.rs` (called from `prover::prove`, reference mod.rs:732). A few hundred QM31 operations per proof; the per-row
// work is k_quotients (quotient.hip).
#pragma once
#include "circle.h"
#include "../kernels.h"
#include <map>
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
    for (auto& kv : by_point) {
        const PtQ& pt = kv.first;
        QuotientBatch qb{};
        qb.prx = pt.x.a; qb.pry = pt.y.a; qb.pix = pt.x.b; qb.piy = pt.y.b;
        qb.kden = c_sub(c_mul(qb.prx, qb.piy), c_mul(qb.pry, qb.pix));   // constant part of the row denominators
        qb.a_sum = q_zero(); qb.b_sum = q_zero();
        Q31 alpha = q_one();
        for (auto& cv : kv.second) {
            alpha = q_mul(alpha, random_coeff);
            // complex_conjugate_line_coeffs: a = conj(v) - v, c = conj(P.y) - P.y, b = v*c - a*P.y; all scaled by alpha
            Q31 a = q_sub(q_conj(cv.second), cv.second);
            Q31 cc = q_sub(q_conj(pt.y), pt.y);
            Q31 b = q_sub(q_mul(cv.second, cc), q_mul(a, pt.y));
            qb.a_sum = q_add(qb.a_sum, q_mul(alpha, a));
            qb.b_sum = q_add(qb.b_sum, q_mul(alpha, b));
            QuotientEntry qe{}; qe.c = q_mul(alpha, cc); qe.col = cv.first;
            entries.push_back(qe);
        }
        qb.batch_coeff = q_pow(random_coeff, kv.second.size());
        qb.n_cols = (u32)kv.second.size();
        batches.push_back(qb);
    }
    // The row value is the Horner sum over batches, acc = acc * batch_coeff_b + term_b, i.e. sum_b term_b * w_b with w_b = the product of the
    // later batches' coefficients. term_b = (sum_k c_k f_k - (A y + B)) / den_b is linear in (c_k, A, B), so the weight is folded into
    // them here and the kernel only adds the terms (exact field arithmetic: the same values, 16 products less per row and extra batch).
    Q31 w = q_one();
    size_t e_end = entries.size();
    for (size_t b = batches.size(); b-- > b0;) {
        QuotientBatch& qb = batches[b];
        const size_t e_begin = e_end - qb.n_cols;
        if (b + 1 < batches.size()) {
            qb.a_sum = q_mul(qb.a_sum, w); qb.b_sum = q_mul(qb.b_sum, w);
            for (size_t e = e_begin; e < e_end; e++) entries[e].c = q_mul(entries[e].c, w);
        }
        w = q_mul(w, qb.batch_coeff);
        e_end = e_begin;
    }
}

}  // namespace bf
