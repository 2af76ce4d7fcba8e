// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// PARITY UNPINNED: the arithmetic restated here lives in the un-vendored dependency
// stwo-prover 0.1.1 @ 31e8dbcc4752240b596774743946c561ab5b9cd1 (reference Cargo.toml:41, Cargo.lock:881-883).
// No golden vector for any field/FFT/hash output exists in the reference, so this file
// restates the published algorithm of stwo `core/fields/{m31,cm31,qm31}.rs` and is pinned only by
// algebraic identities and by the reference's own call sites (e.g. `.inverse()` crates/brainfuck_vm/src/machine.rs:225).
#pragma once
#include "simd_port.h"
#include <cstdint>
#include <cstddef>
#include <vector>
#include <array>
#include <cassert>
#include <stdexcept>

namespace orc {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;

constexpr u32 P = 0x7fffffffu;  // 2^31 - 1

// The byte-level conventions of stwo@31e8dbc that cannot be checked offline, each behind one named switch.
// The values mirror include/bfhip.h `bfhip_conventions`; the product and the oracle are always run with the same setting.
struct Conventions {
    u32 merkle_node_hash = 0;   // 0: zero state + raw compressions (blake2_merkle.rs as published for this period); 1: RFC 7693 Blake2s-256 of the message
    u32 mix_u64 = 0;            // 0: raw compression of [lo, hi, 0..] on the digest words; 1: Blake2s-256(digest || LE64(n) zero padded to 32 bytes)
    u32 logup_mask_order = 0;   // 0: offsets [0, -1] on a component's last logUp column; 1: [-1, 0]
    u32 merkle_channel = 0;     // protocol variant, not a convention: 0 Blake2sMerkleChannel (the reference), 1 Poseidon252MerkleChannel (BASELINE config 5)
};
static inline Conventions& conventions() { static Conventions c; return c; }


// stwo core/fields/m31.rs: M31::reduce — valid for x in [0, P^2).
static inline u32 m31_reduce(u64 x) { return (u32)((((((x >> 31) + x + 1) >> 31) + x)) & P); }

struct M31 {
    u32 v;
    constexpr M31() : v(0) {}
    constexpr explicit M31(u32 x) : v(x) {}           // caller guarantees x < P ("from_u32_unchecked")
    static M31 from(u64 x) { return M31((u32)(x % P)); }
    bool operator==(const M31& o) const { return v == o.v; }
    bool operator!=(const M31& o) const { return v != o.v; }
    bool is_zero() const { return v == 0; }
};
static inline M31 operator+(M31 a, M31 b) { u32 s = a.v + b.v; return M31(s >= P ? s - P : s); }
static inline M31 operator-(M31 a, M31 b) { return M31(a.v >= b.v ? a.v - b.v : a.v + P - b.v); }
static inline M31 operator-(M31 a) { return M31(a.v == 0 ? 0 : P - a.v); }
static inline M31 operator*(M31 a, M31 b) { return M31(m31_reduce((u64)a.v * b.v)); }
static inline M31& operator+=(M31& a, M31 b) { a = a + b; return a; }
static inline M31& operator-=(M31& a, M31 b) { a = a - b; return a; }
static inline M31& operator*=(M31& a, M31 b) { a = a * b; return a; }
static inline M31 m31_pow(M31 b, u64 e) { M31 r(1); while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; } return r; }
// stwo m31.rs `inverse` = x^(P-2) (reference call site: crates/brainfuck_vm/src/machine.rs:225).
static inline M31 inv(M31 a) { if (a.v == 0) throw std::runtime_error("M31 inverse of zero"); return m31_pow(a, P - 2); }

// CM31 = M31[i]/(i^2+1)  (stwo core/fields/cm31.rs)
struct CM31 {
    M31 a, b;
    constexpr CM31() {}
    constexpr CM31(M31 a_, M31 b_) : a(a_), b(b_) {}
    bool operator==(const CM31& o) const { return a == o.a && b == o.b; }
    bool operator!=(const CM31& o) const { return !(*this == o); }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
};
static inline CM31 operator+(CM31 x, CM31 y) { return CM31(x.a + y.a, x.b + y.b); }
static inline CM31 operator-(CM31 x, CM31 y) { return CM31(x.a - y.a, x.b - y.b); }
static inline CM31 operator-(CM31 x) { return CM31(-x.a, -x.b); }
static inline CM31 operator*(CM31 x, CM31 y) { return CM31(x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a); }
static inline CM31 operator*(CM31 x, M31 y) { return CM31(x.a * y, x.b * y); }
static inline CM31 operator-(CM31 x, M31 y) { return CM31(x.a - y, x.b); }
static inline CM31 inv(CM31 x) { M31 n = inv(x.a * x.a + x.b * x.b); return CM31(x.a * n, -(x.b * n)); }

// QM31 = CM31[u]/(u^2 - (2+i))  (stwo core/fields/qm31.rs, R = CM31(2,1)); SECURE_EXTENSION_DEGREE = 4
// (reference use: crates/brainfuck_prover/src/components/mod.rs:15,121-122).
struct QM31 {
    CM31 a, b;
    constexpr QM31() {}
    constexpr QM31(CM31 a_, CM31 b_) : a(a_), b(b_) {}
    explicit QM31(M31 x) : a(x, M31(0)), b() {}
    static QM31 from_u32(u32 a0, u32 a1, u32 a2, u32 a3) { return QM31(CM31(M31(a0), M31(a1)), CM31(M31(a2), M31(a3))); }
    static QM31 from_m31_array(const M31* c) { return QM31(CM31(c[0], c[1]), CM31(c[2], c[3])); }
    std::array<u32, 4> to_u32() const { return {a.a.v, a.b.v, b.a.v, b.b.v}; }
    M31 coord(int i) const { return i == 0 ? a.a : i == 1 ? a.b : i == 2 ? b.a : b.b; }
    static QM31 zero() { return QM31(); }
    static QM31 one() { return QM31(M31(1)); }
    bool operator==(const QM31& o) const { return a == o.a && b == o.b; }
    bool operator!=(const QM31& o) const { return !(*this == o); }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
    // stwo qm31.rs complex_conjugate: (a + bu) -> (a - bu)
    QM31 conj() const { return QM31(a, -b); }
};
static inline CM31 cm31_mul_R(CM31 x) { return CM31(x.a + x.a - x.b, x.a + x.b + x.b); }  // x * (2 + i)
static inline QM31 operator+(QM31 x, QM31 y) { return QM31(x.a + y.a, x.b + y.b); }
static inline QM31 operator-(QM31 x, QM31 y) { return QM31(x.a - y.a, x.b - y.b); }
static inline QM31 operator-(QM31 x) { return QM31(-x.a, -x.b); }
static inline QM31 operator*(QM31 x, QM31 y) { return QM31(x.a * y.a + cm31_mul_R(x.b * y.b), x.a * y.b + x.b * y.a); }
static inline QM31 operator*(QM31 x, M31 y) { return QM31(x.a * y, x.b * y); }
static inline QM31 operator*(M31 y, QM31 x) { return x * y; }
static inline QM31 operator+(QM31 x, M31 y) { return QM31(CM31(x.a.a + y, x.a.b), x.b); }
static inline QM31 operator-(QM31 x, M31 y) { return QM31(CM31(x.a.a - y, x.a.b), x.b); }
static inline QM31& operator+=(QM31& x, QM31 y) { x = x + y; return x; }
static inline QM31& operator-=(QM31& x, QM31 y) { x = x - y; return x; }
static inline QM31& operator*=(QM31& x, QM31 y) { x = x * y; return x; }
static inline QM31 mul_cm31(QM31 x, CM31 y) { return QM31(x.a * y, x.b * y); }
static inline QM31 inv(QM31 x) {
    // (a + bu)^-1 = (a - bu) / (a^2 - R b^2)
    CM31 d = inv(x.a * x.a - cm31_mul_R(x.b * x.b));
    return QM31(x.a * d, -(x.b * d));
}
static inline QM31 qm31_pow(QM31 b, u64 e) { QM31 r = QM31::one(); while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; } return r; }
// stwo qm31.rs from_partial_evals: sum_i evals[i] * basis_i, basis = {1, i, u, iu}
static inline QM31 from_partial_evals(const QM31 e[4]) {
    return e[0] + e[1] * QM31::from_u32(0, 1, 0, 0) + e[2] * QM31::from_u32(0, 0, 1, 0) + e[3] * QM31::from_u32(0, 0, 0, 1);
}

// Montgomery batch inversion (stwo FieldExpOps::batch_inverse) — values identical to elementwise inverse.
template <class F>
static inline void batch_inverse(const F* src, F* dst, size_t n) {
    if (n == 0) return;
    if (simd::enabled() && n >= (size_t(1) << 15)) {
        // SIMD mode of the port (bench.py's cpu_baseline): independent chunks in parallel — an inverse is unique, so the words are the same
        const size_t chunk = size_t(1) << 13;
#pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < n; c0 += chunk) {
            const size_t m = std::min(chunk, n - c0);
            std::vector<F> pref(m);
            F acc = src[c0];
            pref[0] = acc;
            for (size_t i = 1; i < m; i++) { acc = acc * src[c0 + i]; pref[i] = acc; }
            F ia = inv(acc);
            for (size_t i = m - 1; i > 0; i--) { dst[c0 + i] = ia * pref[i - 1]; ia = ia * src[c0 + i]; }
            dst[c0] = ia;
        }
        return;
    }
    std::vector<F> pref(n);
    F acc = src[0];
    pref[0] = acc;
    for (size_t i = 1; i < n; i++) { acc = acc * src[i]; pref[i] = acc; }
    F ia = inv(acc);
    for (size_t i = n - 1; i > 0; i--) { dst[i] = ia * pref[i - 1]; ia = ia * src[i]; }
    dst[0] = ia;
}

static inline u32 bit_reverse_index(u32 i, u32 log_size) {
    if (log_size == 0) return i;
    u32 r = 0;
    for (u32 k = 0; k < log_size; k++) r |= ((i >> k) & 1u) << (log_size - 1 - k);
    return r;
}
template <class T>
static inline void bit_reverse(T* v, size_t n) {
    u32 log = 0; while ((size_t(1) << log) < n) log++;
    for (size_t i = 0; i < n; i++) { size_t j = bit_reverse_index((u32)i, log); if (i < j) std::swap(v[i], v[j]); }
}

}  // namespace orc
