// Device-side Stark-field (felt252) arithmetic, Hades permutation and the Poseidon sponge for gfx950 — shared by poseidon.hip (Merkle
// layers of the Poseidon252 variant) and tools/ubench_poseidon.hip (register-only rate of the same code).
// p = 2^251 + 17 * 2^192 + 1. Inside the kernels an element is 9 limbs of 29 bits (F9) in Montgomery form with R = 2^261 = 2^(9*29);
// the 8 x 32-bit word form (Fe) exists only where hashes and packed column values enter and leave.
//
// Why 29-bit limbs. A limb product is < 2^58 (< 2^60 for the lazily added operands below) and a column of the 9 x 9 product has at most
// 9 of them, so a whole column — products, the two reduction terms and the carry of the column below — is one u64 accumulator fed by
// v_mad_u64_u32 with no carry instruction at all; with 32-bit limbs every mad needs a v_addc into a third word (64 extra instructions per
// product) and the reduction has to chase carries across words. The linear layer gains as much: limb-wise additions have no carry chain
// (on gfx950 a v_addc that consumes the VCC of the previous one costs a wait state, so a 256-bit add is 16 issue slots, not 8), small
// multiples and differences are formed per limb, and one 8-step carry sweep per output restores the limb bound.
// tools/ubench_poseidon.hip: 490 M permutations/s with 8 x 32-bit limbs -> see DESIGN.md §9 for the 9 x 29 figures.
#pragma once
#include "m31.h"

namespace bf {

struct Fe { u32 l[8]; };          // 256-bit little-endian words
struct F9 { u32 l[9]; };          // 29-bit limbs: "normalised" = l[0..7] < 2^29; l[8] holds bits 232 and up

static constexpr u32 M29 = (1u << 29) - 1;
static constexpr u32 F9_C6 = 17u << 18;      // p = 1 + F9_C6 * 2^(29*6) + 2^19 * 2^(29*8)
// round-constant table, 92 rows of 6 x 9 words (tools/gen_poseidon_constants.py). Full rounds: K0, K1, K2 (added to the S-box inputs) and
// L0, L1, L2 (added by the linear layer: multiples of p, spread over the limbs, that keep every limb-wise difference and the reduction
// non-negative). Partial rounds: K2 and Lc only (hades_partial). Row 91: the constants of the change of state around the partial rounds.
static constexpr u32 F9_ROUND_WORDS = 54;

__device__ __forceinline__ Fe fe_load_const(const u32* p) { Fe r; for (int i = 0; i < 8; i++) r.l[i] = p[i]; return r; }
__device__ __forceinline__ F9 f9_load_const(const u32* p) { F9 r; for (int i = 0; i < 9; i++) r.l[i] = p[i]; return r; }
__device__ __forceinline__ F9 f9_zero() { F9 r; for (int i = 0; i < 9; i++) r.l[i] = 0; return r; }

__device__ __forceinline__ F9 to_f9(const Fe& a) {
    F9 r;
    r.l[0] = a.l[0] & M29;
    r.l[1] = __builtin_amdgcn_alignbit(a.l[1], a.l[0], 29) & M29;
    r.l[2] = __builtin_amdgcn_alignbit(a.l[2], a.l[1], 26) & M29;
    r.l[3] = __builtin_amdgcn_alignbit(a.l[3], a.l[2], 23) & M29;
    r.l[4] = __builtin_amdgcn_alignbit(a.l[4], a.l[3], 20) & M29;
    r.l[5] = __builtin_amdgcn_alignbit(a.l[5], a.l[4], 17) & M29;
    r.l[6] = __builtin_amdgcn_alignbit(a.l[6], a.l[5], 14) & M29;
    r.l[7] = __builtin_amdgcn_alignbit(a.l[7], a.l[6], 11) & M29;
    r.l[8] = a.l[7] >> 8;
    return r;
}
__device__ __forceinline__ Fe from_f9(const F9& a) {       // normalised, a.l[8] < 2^24
    Fe r;
    r.l[0] = a.l[0] | (a.l[1] << 29);
    r.l[1] = (a.l[1] >> 3) | (a.l[2] << 26);
    r.l[2] = (a.l[2] >> 6) | (a.l[3] << 23);
    r.l[3] = (a.l[3] >> 9) | (a.l[4] << 20);
    r.l[4] = (a.l[4] >> 12) | (a.l[5] << 17);
    r.l[5] = (a.l[5] >> 15) | (a.l[6] << 14);
    r.l[6] = (a.l[6] >> 18) | (a.l[7] << 11);
    r.l[7] = (a.l[7] >> 21) | (a.l[8] << 8);
    return r;
}
// weak form [0, 2 p) -> canonical [0, p)
__device__ __forceinline__ Fe fe_canon(const Fe& x, const u32* __restrict__ P) {
    Fe s; u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = __builtin_subc(x.l[i], P[i], br, &br);
    return br ? x : s;
}

// ---- Montgomery product ----------------------------------------------------------------------------------------------------------------
// p = 1 (mod 2^29): the factor of step i is m_i = -T_i mod 2^29, and m_i * p adds m_i at column i (clearing its low 29 bits),
// m_i * 17 * 2^18 at column i + 6 and m_i * 2^19 at column i + 8. Columns are produced in order (product scanning); column k < 9 yields
// m_k, column k >= 9 yields result limb k - 9.
// SQUARE: b == a and a2 = 2 a limb-wise — column k is sum_{i < k - i} (2 a_i) a_(k-i) (+ a_(k/2)^2): 45 limb products instead of 81, same column sums.
template <int K, bool SQUARE> __device__ __forceinline__ void f9_columns(F9& r, u32 (&m)[9], u64& acc, const F9& a, const F9& b, const F9& a2) {
    constexpr int i0 = K < 9 ? 0 : K - 8, i1 = K < 9 ? K : 8;
    if constexpr (SQUARE) {
#pragma unroll
        for (int i = i0; 2 * i < K; i++) acc += (u64)a2.l[i] * a.l[K - i];
        if constexpr (K % 2 == 0) acc += (u64)a.l[K / 2] * a.l[K / 2];
    } else {
#pragma unroll
        for (int i = i0; i <= i1; i++) acc += (u64)a.l[i] * b.l[K - i];
    }
    if constexpr (K >= 6 && K - 6 < 9) acc += (u64)m[K - 6] * F9_C6;
    if constexpr (K >= 8) acc += (u64)m[K - 8] * (1u << 19);
    if constexpr (K < 9) { m[K] = (0u - (u32)acc) & M29; acc += m[K]; }
    else r.l[K - 9] = (u32)acc & M29;
    acc >>= 29;
    if constexpr (K < 16) f9_columns<K + 1, SQUARE>(r, m, acc, a, b, a2);
}
// a * b * 2^-261 mod p. Operands: limbs < 2^30 (one lazy addition on top of normalised limbs: 9 * 2^60 + reduction terms < 2^64), values
// < 2^256 (a * b < p * 2^261). Result: normalised, < a * b / 2^261 + p < 2 p.
__device__ __forceinline__ F9 f9_mul(const F9& a, const F9& b) {
    F9 r; u32 m[9]; u64 acc = 0;
    f9_columns<0, false>(r, m, acc, a, b, a);
    r.l[8] = (u32)acc;
    return r;
}
__device__ __forceinline__ F9 f9_sqr(const F9& a) {          // same contract as f9_mul(a, a): limbs < 2^30, so 2 a limbs < 2^31
    F9 a2;
#pragma unroll
    for (int i = 0; i < 9; i++) a2.l[i] = a.l[i] << 1;
    F9 r; u32 m[9]; u64 acc = 0;
    f9_columns<0, true>(r, m, acc, a, a, a2);
    r.l[8] = (u32)acc;
    return r;
}
__device__ __forceinline__ F9 f9_cube(const F9& a) { return f9_mul(f9_sqr(a), a); }
// limb-wise a + k, k a table entry (normalised constants): limbs < 2^30
__device__ __forceinline__ F9 f9_add_const(const F9& a, const u32* __restrict__ k) {
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + k[i];
    return r;
}
// carry sweep: limbs < 2^32 -> normalised (same value)
__device__ __forceinline__ void f9_normalise(F9& x) {
#pragma unroll
    for (int i = 0; i < 8; i++) { x.l[i + 1] += x.l[i] >> 29; x.l[i] &= M29; }
}
// Output of the linear layer (limbs < 2^32, value < 32 p, with the slack the L constants put into limbs 0, 6 and 8) -> normalised, < 3 p:
// q = bits 251 and up, estimated from limb 8 and the carry of limb 7 alone (at most one too small); subtract (q - 1) p — p touches only
// limbs 0, 6, 8 — then sweep the carries. x - (q - 1) p stays positive (q <= x / 2^251) and below 3 * 2^251.
__device__ __forceinline__ F9 f9_reduce(F9 x) {
    const u32 q = (x.l[8] + (x.l[7] >> 29)) >> 19;
    const u32 k = max(q, 1u) - 1u;
    x.l[0] -= k; x.l[6] -= __umul24(k, F9_C6); x.l[8] -= k << 19;       // k <= 24, F9_C6 < 2^23: the full-rate 24-bit multiply
    f9_normalise(x);
    return x;
}

// Hades permutation (width 3, x^3, 4 full + 83 partial + 4 full rounds, MDS [[3,1,1],[1,-1,1],[1,1,-2]]) on Montgomery forms.
// State in: normalised limbs, values < 6 p. State out: normalised, < 3 p.
// Linear layer of a FULL round per limb: o0 = 3 c0 + c1 + c2 + L0, o1 = c0 + c2 + L1 - c1, o2 = c0 + c1 + L2 - 2 c2 where c_i are the S-box
// outputs (the partial rounds run on another state: hades_partial below). L1 and L2 carry 10 p spread so that limb i of L1 exceeds
// any c1 limb and limb i of L2 any 2 c2 limb, L0 carries 2 p; all three leave f9_reduce its slack. Bounds: tools/gen_poseidon_constants.py.
template <bool FULL> __device__ __forceinline__ void hades_round(F9 (&s)[3], const u32* __restrict__ t) {
    F9 c0 = s[0], c1 = s[1];
    if (FULL) { c0 = f9_cube(f9_add_const(s[0], t)); c1 = f9_cube(f9_add_const(s[1], t + 9)); }
    const F9 c2 = f9_cube(f9_add_const(s[2], t + 18));
    F9 o0, o1, o2;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o0.l[i] = 3u * c0.l[i] + c1.l[i] + c2.l[i] + t[27 + i];
        o1.l[i] = c0.l[i] + c2.l[i] + t[36 + i] - c1.l[i];
        o2.l[i] = c0.l[i] + c1.l[i] + t[45 + i] - 2u * c2.l[i];
    }
    s[0] = f9_reduce(o0); s[1] = f9_reduce(o1); s[2] = f9_reduce(o2);
}
// ---- the 83 partial rounds on the state (T1, T2, c) -------------------------------------------------------------------------------------
// Only element 2 passes the S-box there, so elements 0 and 1 are carried by ONE second-order sequence (derivation and constants:
// tools/gen_poseidon_constants.py): per round  y = (c + K2)^3,  c' = 2 T1 + Lc - 2 y,  T' = 2 T1 + 4 T2 + y  — two outputs to bring back to
// normalised limbs instead of three, a linear layer of 2 x 9 x 3 instructions instead of 3 x 9 x 3-4, and no table constant in the T update
// (r03: 520 instructions per partial round, of which ~220 were the linear layer and its three carry sweeps; r04: ~455).
// (a << S) + b in ONE instruction. The compiler prefers a shift and an add (or add3) for the two linear forms below — 6 instructions per limb
// instead of 4; v_lshl_add_u32 is spelled out. _s: b is a table constant (wave-uniform, in an SGPR), _v: b is a lane value.
template <int S> __device__ __forceinline__ u32 lshl_add_v(u32 a, u32 b) { u32 r; asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(S), "v"(b)); return r; }
template <int S> __device__ __forceinline__ u32 lshl_add_s(u32 a, u32 b) { u32 r; asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(S), "s"(b)); return r; }
// x normalised -> x / 2 mod p with limbs < 2^29 + 2^22 (lazily normalised): (x + p) / 2 = (x >> 1) + (p + 1) / 2 for odd x
__device__ __forceinline__ F9 f9_half(const F9& x) {
    const u32 odd = 0u - (x.l[0] & 1u);
    F9 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (x.l[i] >> 1) | ((x.l[i + 1] & 1u) << 28);
    r.l[8] = x.l[8] >> 1;
    r.l[0] += odd & 1u; r.l[6] += odd & (17u << 17); r.l[8] += odd & (1u << 18);      // (p + 1) / 2 = 2^250 + 17 * 2^191 + 1
    return r;
}
// s[0], s[1] (normalised, < 3 p) -> T1 = (s0 + s1) / 2, T2 = (s0 - s1) / 4, both normalised and < 4 p. S8 = 8 p spread (table row 91).
__device__ __forceinline__ void hades_partial_begin(const F9 (&s)[3], F9& T1, F9& T2, const u32* __restrict__ S8) {
    F9 sum, dif;
#pragma unroll
    for (int i = 0; i < 9; i++) { sum.l[i] = s[0].l[i] + s[1].l[i]; dif.l[i] = s[0].l[i] + S8[i] - s[1].l[i]; }
    f9_normalise(sum); f9_normalise(dif);
    T1 = f9_half(sum); f9_normalise(T1);
    T2 = f9_half(dif); f9_normalise(T2);
    T2 = f9_half(T2); f9_normalise(T2);
}
// One partial round. T1 (the newer), T2 (the older): normalised (limbs 0..7 < 2^29 EXACTLY — the T update has no spare bit), values < 4 p;
// c: normalised, < 3 p. The new T replaces the OLDER one in place (T2 becomes the newest): the caller alternates the two arguments
// instead of moving 18 registers per round.
__device__ __forceinline__ void hades_partial(const F9& T1, F9& T2, F9& c, const u32* __restrict__ t) {
    const F9 y = f9_cube(f9_add_const(c, t + 18));
    F9 cn, Tn;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        // 2 T1 + Lc - 2 y as 2 (T1 - y) + Lc: the difference may wrap, the sum is right mod 2^32 and lies in [0, 2^32) (Lc carries 10 p spread:
        // every limb >= 2 y_i + 2^29 - 1) — a subtraction and one v_lshl_add per limb, and no 2 T1 shared with the line below
        cn.l[i] = lshl_add_s<1>(T1.l[i] - y.l[i], t[45 + i]);
        Tn.l[i] = lshl_add_v<1>(T1.l[i], lshl_add_v<2>(T2.l[i], y.l[i]));      // <= 7 (2^29 - 1) per limb, value < 26 p
    }
    c = f9_reduce(cn);
    // T: subtract (q - 1) p with q = bits 251 and up estimated from limb 8 alone (at most one too small: the limbs below add < 9 to it),
    // on top of a value-neutral bias — + 2^29 on limb 0, + 2^29 - 1 on limbs 1..7, - 1 on limb 8 — that keeps limbs 0 and 6 positive under
    // the subtraction and is added where the carry arrives (one v_add3 per step); limbs stay <= 8 * 2^29 - 7 + 7 < 2^32. Result: < 3 * 2^251.
    const u32 q = Tn.l[8] >> 19;
    const u32 k = max(q, 1u) - 1u;                                     // <= 25
    Tn.l[0] += (1u << 29) - k; Tn.l[6] -= __umul24(k, F9_C6); Tn.l[8] -= (k << 19) + 1u;
#pragma unroll
    for (int i = 0; i < 8; i++) { Tn.l[i + 1] += (Tn.l[i] >> 29) + (i < 7 ? M29 : 0u); Tn.l[i] &= M29; }
    T2 = Tn;
}
// back to (s0, s1): s0 = T1 + 2 T2 + A, s1 = T1 - 2 T2 + B (A, B: table row 91, with their spread multiples of p)
__device__ __forceinline__ void hades_partial_end(F9 (&s)[3], const F9& T1, const F9& T2, const F9& c, const u32* __restrict__ t) {
    F9 a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = T1.l[i] + 2u * T2.l[i] + t[27 + i]; b.l[i] = T1.l[i] + t[36 + i] - 2u * T2.l[i]; }
    s[0] = f9_reduce(a); s[1] = f9_reduce(b); s[2] = c;
}
__device__ __forceinline__ void hades(F9 (&s)[3], const u32* __restrict__ table) {
    for (int r = 0; r < 4; r++) hades_round<true>(s, table + r * F9_ROUND_WORDS);
    {
        F9 T1, T2, c = s[2];
        hades_partial_begin(s, T1, T2, table + 91 * F9_ROUND_WORDS);
        for (int r = 4; r < 86; r += 2) {                                   // 41 pairs: after a pair T1 is the newer again
            hades_partial(T1, T2, c, table + r * F9_ROUND_WORDS);
            hades_partial(T2, T1, c, table + (r + 1) * F9_ROUND_WORDS);
        }
        hades_partial(T1, T2, c, table + 86 * F9_ROUND_WORDS);              // round 86: T2 is the newest (T^_86), T1 = T^_85
        hades_partial_end(s, T2, T1, c, table + 91 * F9_ROUND_WORDS);
    }
    for (int r = 87; r < 91; r++) hades_round<true>(s, table + r * F9_ROUND_WORDS);
}

// device constants block (poseidon.hip uploads it): P[8] (words), R1[9], R2[9] (F9: 2^261 mod p, 2^522 mod p), table[92][54]
struct PoseidonConsts {
    const u32* P; const u32* R1; const u32* R2; const u32* table;
    __device__ __forceinline__ explicit PoseidonConsts(const u32* base) : P(base), R1(base + 8), R2(base + 17), table(base + 26) {}
};
static constexpr u32 POSEIDON_CONSTS_WORDS = 26 + 92 * F9_ROUND_WORDS;

// canonical words -> Montgomery F9 (< 2 p), and back
__device__ __forceinline__ F9 f9_from_canonical(const Fe& x, const PoseidonConsts& pc) { return f9_mul(to_f9(x), f9_load_const(pc.R2)); }
__device__ __forceinline__ Fe f9_to_canonical(const F9& x, const PoseidonConsts& pc) {
    F9 one = f9_zero(); one.l[0] = 1;
    return fe_canon(from_f9(f9_mul(x, one)), pc.P);
}

struct Sponge {          // starknet poseidon_hash_many: rate 2, padding 1 0*
    F9 s[3]; u32 count;
    const PoseidonConsts pc;
    __device__ __forceinline__ explicit Sponge(const PoseidonConsts& pc_) : count(0), pc(pc_) { s[0] = f9_zero(); s[1] = f9_zero(); s[2] = f9_zero(); }
    static __device__ __forceinline__ void add_to(F9& x, const F9& v) {      // state < 3 p, v < 2 p (normalised): sum < 6 p, normalised again
#pragma unroll
        for (int i = 0; i < 9; i++) x.l[i] += v.l[i];
        f9_normalise(x);
    }
    __device__ __forceinline__ void absorb(const F9& v) {
        if ((count & 1) == 0) add_to(s[0], v);
        else { add_to(s[1], v); hades(s, pc.table); }
        count++;
    }
    __device__ __forceinline__ F9 finish() {
        const F9 one = f9_load_const(pc.R1);
        if (count & 1) add_to(s[1], one); else add_to(s[0], one);
        hades(s, pc.table);
        return s[0];
    }
};

}  // namespace bf
