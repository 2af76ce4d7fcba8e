// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). PARITY UNPINNED.
// Restates stwo@31e8dbc `core/circle.rs`, `core/poly/circle/{canonic,domain}.rs`, `core/poly/line.rs` (domain part),
// `core/poly/circle/ops.rs` + `core/backend/cpu/circle.rs` (twiddles, circle FFT, eval_at_point), `core/utils.rs`.
// Reference call sites: CanonicCoset::new(log).circle_domain() crates/brainfuck_prover/src/components/memory/table.rs:106,
// precompute_twiddles crates/brainfuck_prover/src/brainfuck_air/mod.rs:480-484, interpolate/evaluate via
// tree_builder.extend_evals/commit mod.rs:497-500,550-583,690-723.
#pragma once
#include "simd_port.h"
#include "field.h"
#include <algorithm>

namespace orc {

template <class F>
struct CirclePoint {
    F x, y;
    CirclePoint() {}
    CirclePoint(F x_, F y_) : x(x_), y(y_) {}
    CirclePoint operator+(const CirclePoint& o) const { return CirclePoint(x * o.x - y * o.y, x * o.y + y * o.x); }
    CirclePoint operator-() const { return CirclePoint(x, -y); }  // conjugate == group inverse
    CirclePoint operator-(const CirclePoint& o) const { return *this + (-o); }
    CirclePoint dbl() const { return *this + *this; }
    bool operator==(const CirclePoint& o) const { return x == o.x && y == o.y; }
};
using PointM = CirclePoint<M31>;
using PointQ = CirclePoint<QM31>;

static inline M31 double_x(M31 x) { M31 s = x * x; return s + s - M31(1); }
static inline QM31 double_x(QM31 x) { QM31 s = x * x; return s + s - M31(1); }
static inline PointQ into_ef(PointM p) { return PointQ(QM31(p.x), QM31(p.y)); }

// stwo circle.rs: M31_CIRCLE_GEN = (2, 1268011823), M31_CIRCLE_LOG_ORDER = 31
constexpr u32 CIRCLE_LOG_ORDER = 31;
constexpr u32 CIRCLE_ORDER_MASK = 0x7fffffffu;  // indices live in Z / 2^31
static inline PointM circle_gen() { return PointM(M31(2), M31(1268011823u)); }

// CirclePointIndex::to_point — double-and-add of the generator.
static inline PointM index_to_point(u32 idx) {
    idx &= CIRCLE_ORDER_MASK;
    PointM res(M31(1), M31(0)), cur = circle_gen();
    while (idx) { if (idx & 1) res = res + cur; cur = cur.dbl(); idx >>= 1; }
    return res;
}
static inline u32 subgroup_gen(u32 log_size) { return 1u << (CIRCLE_LOG_ORDER - log_size); }

struct Coset {
    u32 initial_index, step_size, log_size;
    static Coset make(u32 initial_index, u32 log_size) { return Coset{initial_index & CIRCLE_ORDER_MASK, subgroup_gen(log_size) & CIRCLE_ORDER_MASK, log_size}; }
    static Coset subgroup(u32 log_size) { return make(0, log_size); }
    static Coset odds(u32 log_size) { return make(subgroup_gen(log_size + 1), log_size); }
    static Coset half_odds(u32 log_size) { return make(subgroup_gen(log_size + 2), log_size); }
    size_t size() const { return size_t(1) << log_size; }
    u32 index_at(size_t i) const { return (u32)((initial_index + (u64)step_size * i) & CIRCLE_ORDER_MASK); }
    PointM at(size_t i) const { return index_to_point(index_at(i)); }
    PointM initial() const { return index_to_point(initial_index); }
    PointM step() const { return index_to_point(step_size); }
    Coset dbl() const { return Coset{(initial_index * 2) & CIRCLE_ORDER_MASK, (step_size * 2) & CIRCLE_ORDER_MASK, log_size ? log_size - 1 : 0}; }
    // All points, by repeated addition of the step (Coset::iter()).
    std::vector<PointM> points() const {
        std::vector<PointM> out(size());
        if (simd::enabled() && out.size() >= (size_t(1) << 14)) {
            // SIMD mode of the port (bench.py's cpu_baseline): the walk in parallel chunks, each started at its own index — the group law is exact, so
            // the points are the same words as the serial walk's
            const size_t chunk = size_t(1) << 12, n = out.size();
            const PointM s = step();
#pragma omp parallel for schedule(static)
            for (size_t c0 = 0; c0 < n; c0 += chunk) {
                PointM p = at(c0);
                for (size_t i = c0; i < c0 + chunk; i++) { out[i] = p; p = p + s; }
            }
            return out;
        }
        PointM p = initial(), s = step();
        for (size_t i = 0; i < out.size(); i++) { out[i] = p; p = p + s; }
        return out;
    }
};

// CircleDomain = half_coset ∪ conjugate(half_coset); natural index i<half -> half_coset.at(i), else -half_coset.at(i-half).
struct CircleDomain {
    Coset half_coset;
    u32 log_size() const { return half_coset.log_size + 1; }
    size_t size() const { return size_t(1) << log_size(); }
    u32 index_at(size_t i) const {
        size_t h = half_coset.size();
        if (i < h) return half_coset.index_at(i);
        return (0u - half_coset.index_at(i - h)) & CIRCLE_ORDER_MASK;
    }
    PointM at(size_t i) const { return index_to_point(index_at(i)); }
};

// CanonicCoset::new(log) = Coset::odds(log); circle_domain() = CircleDomain(half_odds(log-1)).
struct CanonicCoset {
    u32 log;
    Coset coset() const { return Coset::odds(log); }
    Coset half_coset() const { return Coset::half_odds(log - 1); }
    CircleDomain circle_domain() const { return CircleDomain{half_coset()}; }
    u32 step_size() const { return subgroup_gen(log); }
    PointM step() const { return index_to_point(step_size()); }
};

// LineDomain over a coset: at(i) = coset.at(i).x
struct LineDomain {
    Coset coset;
    u32 log_size() const { return coset.log_size; }
    size_t size() const { return coset.size(); }
    M31 at(size_t i) const { return coset.at(i).x; }
    LineDomain dbl() const { return LineDomain{coset.dbl()}; }
};

// stwo core/utils.rs
static inline size_t coset_index_to_circle_domain_index(size_t coset_index, u32 log_domain_size) {
    if ((coset_index & 1) == 0) return coset_index / 2;
    return ((size_t(2) << log_domain_size) - coset_index) / 2;
}
static inline size_t circle_domain_index_to_coset_index(size_t d, u32 log_domain_size) {
    size_t n = size_t(1) << log_domain_size;
    if (d < n / 2) return d * 2;
    return (n - 1 - d) * 2 + 1;
}
// stwo core/utils.rs offset_bit_reversed_circle_domain_index
static inline size_t offset_bit_reversed_circle_domain_index(size_t i, u32 domain_log_size, u32 eval_log_size, long offset) {
    long prev = (long)bit_reverse_index((u32)i, eval_log_size);
    long half = 1L << (eval_log_size - 1);
    long step = offset * (1L << (eval_log_size - domain_log_size - 1));
    if (prev < half) prev = (((prev + step) % half) + half) % half;
    else prev = ((((prev - step) % half) + half) % half) + half;
    return bit_reverse_index((u32)prev, eval_log_size);
}

// stwo core/constraints.rs coset_vanishing
template <class F>
static inline F coset_vanishing(Coset coset, CirclePoint<F> p);
template <>
inline M31 coset_vanishing<M31>(Coset coset, PointM p) {
    p = p - coset.initial() + index_to_point(coset.step_size >> 1);
    M31 x = p.x;
    for (u32 i = 1; i < coset.log_size; i++) x = double_x(x);
    return x;
}
template <>
inline QM31 coset_vanishing<QM31>(Coset coset, PointQ p) {
    p = p - into_ef(coset.initial()) + into_ef(index_to_point(coset.step_size >> 1));
    QM31 x = p.x;
    for (u32 i = 1; i < coset.log_size; i++) x = double_x(x);
    return x;
}

// ---------------------------------------------------------------------------------------------
// Twiddles (stwo core/poly/twiddles.rs + backend/cpu/circle.rs slow_precompute_twiddles)
// ---------------------------------------------------------------------------------------------
struct TwiddleTree {
    Coset root_coset;
    std::vector<M31> twiddles, itwiddles;
};

static inline std::vector<M31> slow_precompute_twiddles(Coset coset) {
    std::vector<M31> tw;
    tw.reserve(coset.size());
    u32 logn = coset.log_size;
    for (u32 l = 0; l < logn; l++) {
        size_t i0 = tw.size(), half = coset.size() / 2;
        if (simd::enabled() && half >= (size_t(1) << 14)) {
            // SIMD mode of the port: the layer's points in parallel chunks, written straight to their bit-reversed places (same words)
            tw.resize(i0 + half);
            const size_t chunk = size_t(1) << 12;
            const PointM s = coset.step();
            u32 lg = 0; while ((size_t(1) << lg) < half) lg++;
            M31* dst = tw.data() + i0;
#pragma omp parallel for schedule(static)
            for (size_t c0 = 0; c0 < half; c0 += chunk) {
                PointM p = coset.at(c0);
                for (size_t i = c0; i < c0 + chunk; i++) { dst[bit_reverse_index((u32)i, lg)] = p.x; p = p + s; }
            }
            coset = coset.dbl();
            continue;
        }
        PointM p = coset.initial(), s = coset.step();
        for (size_t i = 0; i < half; i++) { tw.push_back(p.x); p = p + s; }
        bit_reverse(tw.data() + i0, half);
        coset = coset.dbl();
    }
    tw.push_back(M31(1));
    return tw;
}
static inline TwiddleTree precompute_twiddles(Coset coset) {
    TwiddleTree t;
    t.root_coset = coset;
    t.twiddles = slow_precompute_twiddles(coset);
    t.itwiddles.resize(t.twiddles.size());
    batch_inverse(t.twiddles.data(), t.itwiddles.data(), t.twiddles.size());
    return t;
}

// domain_line_twiddles_from_tree: layer i (i = 0 is the first line layer, the largest) of a domain with half_coset log k
// reads buf[len - 2*2^(k-1-i) .. len - 2^(k-1-i)).
static inline const M31* line_twiddles_layer(const std::vector<M31>& buf, u32 half_log, u32 layer, size_t* len_out) {
    size_t len = size_t(1) << (half_log - 1 - layer);
    *len_out = len;
    return buf.data() + (buf.size() - 2 * len);
}

static inline void butterfly(M31& v0, M31& v1, M31 t) { M31 tmp = v1 * t; v1 = v0 - tmp; v0 = v0 + tmp; }
static inline void ibutterfly(M31& v0, M31& v1, M31 it) { M31 tmp = v0; v0 = tmp + v1; v1 = (tmp - v1) * it; }
static inline void ibutterfly(QM31& v0, QM31& v1, M31 it) { QM31 tmp = v0; v0 = tmp + v1; v1 = (tmp - v1) * it; }

// Circle twiddle h of the circle layer derived from first line layer: chunks [x, y] -> [y, -y, -x, x]
static inline M31 circle_twiddle(const M31* first_line, size_t h) {
    M31 x = first_line[(h >> 2) * 2], y = first_line[(h >> 2) * 2 + 1];
    switch (h & 3) { case 0: return y; case 1: return -y; case 2: return -x; default: return x; }
}

// CircleEvaluation (bit-reversed order) -> CirclePoly coefficients. CpuBackend::interpolate, log_size >= 3.
static inline void circle_interpolate(M31* values, u32 log_size, const TwiddleTree& tw) {
    assert(log_size >= 3);
    if (simd::enabled() && log_size >= 5) {      // SIMD mode of the port (orc_set_simd: bench.py's cpu_baseline): same values, 16 lanes per instruction
        simd::circle_ifft(reinterpret_cast<u32*>(values), log_size, reinterpret_cast<const u32*>(tw.itwiddles.data()), tw.itwiddles.size(), inv(M31(u32(1) << log_size)).v);
        return;
    }
    u32 half_log = log_size - 1;
    size_t n = size_t(1) << log_size;
    size_t l0len; const M31* l0 = line_twiddles_layer(tw.itwiddles, half_log, 0, &l0len);
    for (size_t h = 0; h < n / 2; h++) { M31 t = circle_twiddle(l0, h); ibutterfly(values[2 * h], values[2 * h + 1], t); }
    for (u32 layer = 0; layer < half_log; layer++) {
        size_t len; const M31* lt = line_twiddles_layer(tw.itwiddles, half_log, layer, &len);
        u32 i = layer + 1;
        for (size_t h = 0; h < len; h++)
            for (size_t l = 0; l < (size_t(1) << i); l++) {
                size_t idx0 = (h << (i + 1)) + l, idx1 = idx0 + (size_t(1) << i);
                ibutterfly(values[idx0], values[idx1], lt[h]);
            }
    }
    M31 ninv = inv(M31((u32)n));
    for (size_t i = 0; i < n; i++) values[i] = values[i] * ninv;
}

// CirclePoly coefficients (already zero-extended to 2^log_size) -> evaluation on CanonicCoset(log_size).circle_domain(),
// bit-reversed order. CpuBackend::evaluate.
static inline void circle_evaluate(M31* values, u32 log_size, const TwiddleTree& tw) {
    assert(log_size >= 3);
    if (simd::enabled() && log_size >= 5) {
        simd::circle_fft(reinterpret_cast<u32*>(values), log_size, reinterpret_cast<const u32*>(tw.twiddles.data()), tw.twiddles.size());
        return;
    }
    u32 half_log = log_size - 1;
    size_t n = size_t(1) << log_size;
    for (int layer = (int)half_log - 1; layer >= 0; layer--) {
        size_t len; const M31* lt = line_twiddles_layer(tw.twiddles, half_log, (u32)layer, &len);
        u32 i = (u32)layer + 1;
        for (size_t h = 0; h < len; h++)
            for (size_t l = 0; l < (size_t(1) << i); l++) {
                size_t idx0 = (h << (i + 1)) + l, idx1 = idx0 + (size_t(1) << i);
                butterfly(values[idx0], values[idx1], lt[h]);
            }
    }
    size_t l0len; const M31* l0 = line_twiddles_layer(tw.twiddles, half_log, 0, &l0len);
    for (size_t h = 0; h < n / 2; h++) { M31 t = circle_twiddle(l0, h); butterfly(values[2 * h], values[2 * h + 1], t); }
}

// CpuBackend::eval_at_point — fold of the coefficients with [.., double_x(x), x, y] (mappings reversed).
static inline QM31 eval_at_point(const M31* coeffs, u32 log_size, PointQ p) {
    if (log_size == 0) return QM31(coeffs[0]);
    std::vector<QM31> mappings;
    mappings.push_back(p.y);
    QM31 x = p.x;
    for (u32 i = 1; i < log_size; i++) { mappings.push_back(x); x = double_x(x); }
    // Iterative equivalent of the recursive `fold`: the factor for bit b (LSB = 0) of the coefficient index is mappings[b].
    size_t n = size_t(1) << log_size;
    std::vector<QM31> cur(n / 2);
    for (size_t i = 0; i < n / 2; i++) cur[i] = QM31(coeffs[2 * i]) + mappings[0] * coeffs[2 * i + 1];
    size_t m = n / 2;
    for (u32 b = 1; b < log_size; b++) {
        for (size_t i = 0; i < m / 2; i++) cur[i] = cur[2 * i] + cur[2 * i + 1] * mappings[b];
        m /= 2;
    }
    return cur[0];
}

// Line iFFT used for the FRI last layer (stwo core/poly/line.rs LineEvaluation::interpolate → line_ifft), natural coefficient order out.
static inline std::vector<QM31> line_interpolate(std::vector<QM31> values, LineDomain domain) {
    size_t n = values.size();
    bit_reverse(values.data(), n);  // LineEvaluation<BitReversedOrder> -> natural
    // line_ifft on natural order values
    LineDomain d = domain;
    for (size_t chunk = n; chunk > 1; chunk >>= 1) {
        // twiddles: inverse of x-coords of first half of the domain
        std::vector<M31> itw(chunk / 2);
        for (size_t i = 0; i < chunk / 2; i++) itw[i] = inv(d.at(i));
        for (size_t c = 0; c < n; c += chunk)
            for (size_t i = 0; i < chunk / 2; i++) ibutterfly(values[c + i], values[c + chunk / 2 + i], itw[i]);
        d = d.dbl();
    }
    M31 ninv = inv(M31((u32)n));
    for (auto& v : values) v = v * ninv;
    // LinePoly stores bit-reversed coefficients; into_ordered_coefficients() bit-reverses back.
    // line_ifft leaves them in bit-reversed order already -> ordered = bit_reverse(values).
    bit_reverse(values.data(), n);
    return values;
}

}  // namespace orc
