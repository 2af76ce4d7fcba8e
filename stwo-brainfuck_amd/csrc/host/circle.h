// Host-side circle-group helpers for the handful of points the protocol needs on the CPU (OODS point, mask points, vanishing
// polynomial inverses, quotient line coefficients). Bulk domain arithmetic lives in the kernels.
// Semantics: stwo `core/circle.rs`, `core/poly/circle/canonic.rs`, `core/constraints.rs` (coset_vanishing).
#pragma once
#include "../m31.h"

namespace bf {

struct PtM { u32 x, y; };
struct PtQ { Q31 x, y; };

inline PtM pm_add(PtM a, PtM b) { return {m_sub(m_mul(a.x, b.x), m_mul(a.y, b.y)), m_add(m_mul(a.x, b.y), m_mul(a.y, b.x))}; }
inline PtM pm_neg(PtM a) { return {a.x, m_neg(a.y)}; }
inline PtQ pq_add(PtQ a, PtQ b) { return {q_sub(q_mul(a.x, b.x), q_mul(a.y, b.y)), q_add(q_mul(a.x, b.y), q_mul(a.y, b.x))}; }
inline PtQ pq_neg(PtQ a) { return {a.x, q_neg(a.y)}; }
inline PtQ to_q(PtM p) { return {q_from_m(p.x), q_from_m(p.y)}; }

// G^idx for the circle generator G = (2, 1268011823) of order 2^31
inline PtM index_to_point(u32 idx) {
    idx &= 0x7fffffffu;
    PtM res{1, 0}, cur{2u, 1268011823u};
    while (idx) { if (idx & 1) res = pm_add(res, cur); cur = pm_add(cur, cur); idx >>= 1; }
    return res;
}
inline u32 subgroup_gen(u32 log) { return 1u << (31 - log); }
inline Q31 q_double_x(Q31 x) { Q31 s = q_mul(x, x); return q_subm(q_add(s, s), 1); }
inline u32 m_double_x(u32 x) { u32 s = m_sqr(x); return m_sub(m_add(s, s), 1); }

// CanonicCoset(log): coset = odds(log) = { G^(2^(30-log) * (2k+1)) }, trace step = G^(2^(31-log)).
// coset_vanishing(CanonicCoset(log).coset, p): rotate by -initial + step/2, take x, double (log - 1) times.
inline Q31 coset_vanishing_q(u32 log, PtQ p) {
    u32 initial = subgroup_gen(log + 1), step = subgroup_gen(log);
    p = pq_add(pq_add(p, pq_neg(to_q(index_to_point(initial)))), to_q(index_to_point(step >> 1)));
    Q31 x = p.x;
    for (u32 i = 1; i < log; i++) x = q_double_x(x);
    return x;
}
inline u32 coset_vanishing_m(u32 log, PtM p) {
    u32 initial = subgroup_gen(log + 1), step = subgroup_gen(log);
    p = pm_add(pm_add(p, pm_neg(index_to_point(initial))), index_to_point(step >> 1));
    u32 x = p.x;
    for (u32 i = 1; i < log; i++) x = m_double_x(x);
    return x;
}
// CanonicCoset(log).circle_domain().at(i): half_coset = half_odds(log - 1) = { G^(2^(30-log) + i * 2^(32-log)) }, conjugates after.
inline PtM canonic_domain_at(u32 log, u32 i) {
    u32 half = 1u << (log - 1);
    u32 initial = subgroup_gen(log + 1), step = subgroup_gen(log - 1);
    if (i < half) return index_to_point(initial + step * i);
    return pm_neg(index_to_point(initial + step * (i - half)));
}

}  // namespace bf
