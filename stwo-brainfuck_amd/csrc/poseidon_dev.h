// Device-side Stark-field (felt252) arithmetic, Hades permutation and the Poseidon sponge for gfx950 — shared by poseidon.hip (Merkle
// layers of the Poseidon252 variant) and tools/ubench_poseidon.hip (register-only rate of the same code).
// p = 2^251 + 17 * 2^192 + 1, 8 x 32-bit limbs, Montgomery form with R = 2^256. p = 1 (mod 2^32), so the Montgomery factor of every CIOS
// step is m = -t0 and m * p touches only limbs 0, 6, 7.
#pragma once
#include "m31.h"

namespace bf {

struct Fe { u32 l[8]; };

__device__ __forceinline__ Fe fe_load_const(const u32* p) { Fe r; for (int i = 0; i < 8; i++) r.l[i] = p[i]; return r; }

__device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b, const u32* __restrict__ P) {
    Fe r; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.l[i] + b.l[i]; r.l[i] = (u32)c; c >>= 32; }
    Fe s; u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)r.l[i] - P[i] - br; s.l[i] = (u32)d; br = d >> 63; }
    return br ? r : s;
}
__device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b, const u32* __restrict__ P) {
    Fe r; u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)a.l[i] - b.l[i] - br; r.l[i] = (u32)d; br = d >> 63; }
    if (br) { u64 c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (u64)r.l[i] + P[i]; r.l[i] = (u32)c; c >>= 32; } }
    return r;
}
// ---- lazy (weakly reduced) arithmetic used by the Hades permutation ---------------------------------------------------------------------
// The 8 limbs hold 256 bits and p < 2^251.0001, so sums of up to 15 p fit. State elements are kept in [0, 2p) ("weak form") between rounds,
// the additions of the linear layer are plain 256-bit additions, and one cheap weak reduction per output replaces twelve
// compare-and-subtract modular additions per round. The Montgomery product needs a * b < p * 2^256, i.e. operands up to 5 p, and returns
// a value below a * b / 2^256 + p < 2 p without a final subtraction. Only the squeezed hash is brought to canonical form.

// 96-bit multiply-accumulate (hi : lo) += sum a_i * b_i for one column of the product: per term v_mad_u64_u32 adds the 64-bit product into `lo`
// and leaves the carry in VCC, one v_addc_co_u32 folds it into `hi` — two instructions per limb product. One asm statement per COLUMN:
// the compiler pads every asm statement with an s_nop, and its own u64 formulation of a carry-save row takes three to four
// instructions plus register moves per product.
#define BF_MAC(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define BF_MAC_OUT : "+v"(lo), "+v"(hi)
__device__ __forceinline__ void mac_col(u64& lo, u32& hi, u32 a0, u32 b0) { asm(BF_MAC("%2", "%3") BF_MAC_OUT : "v"(a0), "v"(b0) : "vcc"); }
__device__ __forceinline__ void mac_col(u64& lo, u32& hi, u32 a0, u32 b0, u32 a1, u32 b1) {
    asm(BF_MAC("%2", "%3") BF_MAC("%4", "%5") BF_MAC_OUT : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void mac_col(u64& lo, u32& hi, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2) {
    asm(BF_MAC("%2", "%3") BF_MAC("%4", "%5") BF_MAC("%6", "%7") BF_MAC_OUT : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2) : "vcc");
}
__device__ __forceinline__ void mac_col(u64& lo, u32& hi, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) {
    asm(BF_MAC("%2", "%3") BF_MAC("%4", "%5") BF_MAC("%6", "%7") BF_MAC("%8", "%9") BF_MAC_OUT : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc");
}
#undef BF_MAC_OUT
// column K of a * b: products a_i * b_(K - i), i in [max(0, K - 7), min(7, K)], in groups of at most four per asm statement
template <int K> __device__ __forceinline__ void comba_col(u64& lo, u32& hi, const Fe& a, const Fe& b) {
    constexpr int i0 = K < 8 ? 0 : K - 7, i1 = K < 8 ? K : 7, n = i1 - i0 + 1;
    constexpr int n4 = n >= 4 ? 4 : n;
    if constexpr (n4 == 1) mac_col(lo, hi, a.l[i0], b.l[K - i0]);
    else if constexpr (n4 == 2) mac_col(lo, hi, a.l[i0], b.l[K - i0], a.l[i0 + 1], b.l[K - i0 - 1]);
    else if constexpr (n4 == 3) mac_col(lo, hi, a.l[i0], b.l[K - i0], a.l[i0 + 1], b.l[K - i0 - 1], a.l[i0 + 2], b.l[K - i0 - 2]);
    else mac_col(lo, hi, a.l[i0], b.l[K - i0], a.l[i0 + 1], b.l[K - i0 - 1], a.l[i0 + 2], b.l[K - i0 - 2], a.l[i0 + 3], b.l[K - i0 - 3]);
    constexpr int r = n - n4, j0 = i0 + n4;      // remaining products of the column (at most four)
    if constexpr (r == 1) mac_col(lo, hi, a.l[j0], b.l[K - j0]);
    else if constexpr (r == 2) mac_col(lo, hi, a.l[j0], b.l[K - j0], a.l[j0 + 1], b.l[K - j0 - 1]);
    else if constexpr (r == 3) mac_col(lo, hi, a.l[j0], b.l[K - j0], a.l[j0 + 1], b.l[K - j0 - 1], a.l[j0 + 2], b.l[K - j0 - 2]);
    else if constexpr (r == 4) mac_col(lo, hi, a.l[j0], b.l[K - j0], a.l[j0 + 1], b.l[K - j0 - 1], a.l[j0 + 2], b.l[K - j0 - 2], a.l[j0 + 3], b.l[K - j0 - 3]);
}
template <int K> __device__ __forceinline__ void comba_cols(u32 (&T)[16], u64& lo, u32& hi, const Fe& a, const Fe& b) {
    comba_col<K>(lo, hi, a, b);
    T[K] = (u32)lo; lo = (lo >> 32) | ((u64)hi << 32); hi = 0;
    if constexpr (K < 14) comba_cols<K + 1>(T, lo, hi, a, b);
}
// Montgomery product a * b * 2^-256 mod p in weak form: operands < 5 p (a * b < p * 2^256), result < 2 p.
// Product scanning (Comba) for the 512-bit product, then one reduction sweep: p = 1 + 17 * 2^192 + 2^251 = 1 (mod 2^32), so the factor of step
// i is m_i = -T'[i] and m_i * p only (a) zeroes limb i, leaving a carry that is simply "some limb up to i was non-zero", and (b) adds
// 17 m_i at limb i + 6 and m_i * 2^27 across limbs i + 7, i + 8.
__device__ __forceinline__ Fe fe_mul_weak(const Fe& a, const Fe& b) {
    u32 T[16];
    u64 lo = 0; u32 hi = 0;
    comba_cols<0>(T, lo, hi, a, b);
    T[15] = (u32)lo;
    u32 A[8], B[8], C[8];          // what m_i * p adds at limbs i + 6, i + 7, i + 8
    auto terms = [&](int i, u32 m) { const u64 m17 = (u64)m * 17u; A[i] = (u32)m17; B[i] = (u32)(m17 >> 32) | (m << 27); C[i] = m >> 5; };
    u32 nz = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { terms(i, 0u - (T[i] + c)); nz |= T[i]; c = nz ? 1u : 0u; }
    Fe r;
#pragma unroll
    for (int k = 6; k < 16; k++) {
        // limb k = T[k] + A[k-6] + B[k-7] + C[k-8] + carry; C and the carry are small (27 bits + a few units): pre-added without overflow
        u32 small = c + (k >= 8 ? C[k - 8] : 0u);
        u32 c1 = 0, c2 = 0, c3 = 0;
        u32 v = T[k];
        if (k - 6 < 8) v = __builtin_addc(v, A[k - 6], 0u, &c1);
        if (k >= 7 && k - 7 < 8) v = __builtin_addc(v, B[k - 7], 0u, &c2);
        v = __builtin_addc(v, small, 0u, &c3);
        c = c1 + c2 + c3;
        if (k < 8) { terms(k, 0u - v); c += v ? 1u : 0u; }     // limbs 6, 7 are still reduction steps: zero them, carry "was non-zero"
        else r.l[k - 8] = v;
    }
    return r;
}
// [0, 2^255) -> [0, 2 p): subtract (q - 1) p with q = x >> 251 (the estimate q p may exceed x by a little, (q - 1) p never does)
__device__ __forceinline__ Fe fe_weak_reduce(const Fe& x) {
    const u32 q = x.l[7] >> 27;
    const u32 k = q ? q - 1 : 0u;
    Fe r; u32 br = 0;
    r.l[0] = __builtin_subc(x.l[0], k, 0u, &br);
#pragma unroll
    for (int i = 1; i < 6; i++) r.l[i] = __builtin_subc(x.l[i], 0u, br, &br);
    r.l[6] = __builtin_subc(x.l[6], 17u * k, br, &br);
    r.l[7] = __builtin_subc(x.l[7], k << 27, br, &br);
    return r;
}
__device__ __forceinline__ Fe add256(const Fe& a, const Fe& b) {
    Fe r; u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    return r;
}
__device__ __forceinline__ Fe sub256(const Fe& a, const Fe& b) {
    Fe r; u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __builtin_subc(a.l[i], b.l[i], br, &br);
    return r;
}
// x + 6 p  (6 p = 6 + 102 * 2^192 + 6 * 2^251: limbs {6, 0, 0, 0, 0, 0, 102, 0x30000000})
__device__ __forceinline__ Fe add_6p(const Fe& x) {
    Fe r; u32 c = 0;
    r.l[0] = __builtin_addc(x.l[0], 6u, 0u, &c);
#pragma unroll
    for (int i = 1; i < 6; i++) r.l[i] = __builtin_addc(x.l[i], 0u, c, &c);
    r.l[6] = __builtin_addc(x.l[6], 102u, c, &c);
    r.l[7] = __builtin_addc(x.l[7], 0x30000000u, c, &c);
    return r;
}
// weak form [0, 2 p) -> canonical [0, p)
__device__ __forceinline__ Fe fe_canon(const Fe& x, const u32* __restrict__ P) {
    Fe s; u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = __builtin_subc(x.l[i], P[i], br, &br);
    return br ? x : s;
}
// Montgomery product with canonical result (conversions in and out of Montgomery form)
__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b, const u32* __restrict__ P) { return fe_canon(fe_mul_weak(a, b), P); }

// Hades permutation on weak-form Montgomery elements: state in [0, 2 p) in and out.
__device__ __forceinline__ void hades(Fe s[3], const u32* __restrict__ ark, const u32* __restrict__ P) {
    (void)P;
    for (int r = 0; r < 91; r++) {
        const u32* k = ark + (size_t)r * 24;
        s[0] = add256(s[0], fe_load_const(k));              // < 3 p
        s[1] = add256(s[1], fe_load_const(k + 8));
        s[2] = add256(s[2], fe_load_const(k + 16));
        const bool full = r < 4 || r >= 87;
        if (full) {
            Fe q0 = fe_mul_weak(s[0], s[0]), q1 = fe_mul_weak(s[1], s[1]);
            s[0] = fe_mul_weak(q0, s[0]); s[1] = fe_mul_weak(q1, s[1]);   // < 2 p
        }
        Fe q2 = fe_mul_weak(s[2], s[2]);
        s[2] = fe_mul_weak(q2, s[2]);
        // MDS [[3,1,1],[1,-1,1],[1,1,-2]]: t = s0 + s1 + s2 (< 8 p); t + 2 s0, t + 6 p - 2 s1, t + 6 p - 3 s2 (each < 14 p < 2^255, non-negative)
        const Fe t = add256(add256(s[0], s[1]), s[2]);
        const Fe t6 = add_6p(t);
        const Fe d0 = add256(s[0], s[0]), d1 = add256(s[1], s[1]), d2 = add256(add256(s[2], s[2]), s[2]);
        s[0] = fe_weak_reduce(add256(t, d0)); s[1] = fe_weak_reduce(sub256(t6, d1)); s[2] = fe_weak_reduce(sub256(t6, d2));
    }
}

struct Sponge {
    Fe s[3]; u32 count;
    const u32* ark; const u32* P; const u32* R1;
    __device__ __forceinline__ void init(const u32* ark_, const u32* P_, const u32* R1_) {
        ark = ark_; P = P_; R1 = R1_; count = 0;
        for (int k = 0; k < 3; k++) for (int i = 0; i < 8; i++) s[k].l[i] = 0;
    }
    __device__ __forceinline__ void absorb(const Fe& v) {      // v in Montgomery form, canonical or weak (< 2 p)
        if ((count & 1) == 0) s[0] = fe_weak_reduce(add256(s[0], v));
        else { s[1] = fe_weak_reduce(add256(s[1], v)); hades(s, ark, P); }
        count++;
    }
    __device__ __forceinline__ Fe finish() {                    // poseidon_hash_many padding: a single one; result in weak form
        Fe one = fe_load_const(R1);
        if (count & 1) s[1] = fe_weak_reduce(add256(s[1], one)); else s[0] = fe_weak_reduce(add256(s[0], one));
        hades(s, ark, P);
        return s[0];
    }
};

}  // namespace bf
