// M31 / CM31 / QM31 arithmetic shared by host code and gfx950 kernels.
// Replaces stwo's `core/fields/{m31,cm31,qm31}.rs` + SIMD `PackedM31/PackedQM31` as used by the reference at
// crates/brainfuck_prover/src/components/mod.rs:13-19 and memory/table.rs:9-18.
// Storage is a canonical u32 in [0, P); QM31 columns are 4 x u32 SoA (SecureColumnByCoords).
#pragma once
#include <cstdint>
#include <cstddef>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BF_HD __host__ __device__ __forceinline__
#else
#define BF_HD inline
#endif

namespace bf {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;

constexpr u32 P31 = 0x7fffffffu;

// Byte-level conventions of stwo@31e8dbc that cannot be confirmed offline, one named switch each; the numbering
// is that of include/bfhip.h `bfhip_conventions`. Defaults = the published stwo code of the period to the best of our reconstruction.
struct Conventions {
    u32 merkle_node_hash = 0;   // 0 BFHIP_MERKLE_STWO_COMPRESS: zero state, raw compress(state, block, 0,0,0,0) per 64-byte block; 1 BFHIP_MERKLE_RFC7693
    u32 mix_u64 = 0;            // 0 BFHIP_MIX_U64_COMPRESS: raw compression on the digest words; 1 BFHIP_MIX_U64_HASH: Blake2s(digest || pad32(LE64 n))
    u32 logup_mask_order = 0;   // 0 BFHIP_LOGUP_MASK_CUR_PREV: offsets [0, -1] on a component's last logUp column; 1 BFHIP_LOGUP_MASK_PREV_CUR: [-1, 0]
    // Not a convention but a protocol variant carried in the same struct: which MerkleChannel the proof uses.
    u32 merkle_channel = 0;     // 0 BFHIP_CHANNEL_BLAKE2S: Blake2sMerkleChannel (the reference, mod.rs:56,486-487); 1 BFHIP_CHANNEL_POSEIDON252
};

BF_HD u32 m_add(u32 a, u32 b) { u32 s = a + b; u32 t = s - P31; return t < s ? t : s; }           // min(s, s-P) with wraparound
BF_HD u32 m_sub(u32 a, u32 b) { u32 s = a - b; u32 t = s + P31; return t < s ? t : s; }           // a-b or a-b+P
// Lazy reduction for dot products: a product of canonical values is < 2^62, so three products fit on top of a folded accumulator
// (< 2^34) in 64 bits; m_fold brings an accumulator back below 2^34, m_canon to the canonical representative.
BF_HD u64 m_fold(u64 x) { return (x & P31) + (x >> 31); }
BF_HD u32 m_canon(u64 x) { x = m_fold(m_fold(x)); u32 r = (u32)x; return r >= P31 ? r - P31 : r; }   // x < 2^64 -> < 2^34 -> < 2^31 + 8
BF_HD u32 m_neg(u32 a) { return a ? P31 - a : 0; }
BF_HD u32 m_reduce64(u64 x) {  // x < P^2 -> [0, P)
    u32 lo = (u32)x & P31, hi = (u32)(x >> 31);
    u32 s = lo + hi; u32 t = s - P31; return t < s ? t : s;
}
BF_HD u32 m_mul(u32 a, u32 b) { return m_reduce64((u64)a * b); }
// Product by a constant that is kept DOUBLED (w2 = 2 w < 2^32, w canonical): the 64-bit product a * w2 = 2 (a w) has (a w) >> 31 in its high
// word and ((a w) & P) << 1 in its low word, so the fold is high + (low >> 1) — one shift and one add where m_reduce64 needs a mask, a
// funnel shift and an add. 5 VALU instructions per product instead of 6 (r04: the butterflies' twiddles are staged doubled).
BF_HD u32 m_mul_pre2(u32 a, u32 w2) {
    u64 p = (u64)a * w2;
    u32 s = (u32)(p >> 32) + ((u32)p >> 1); u32 t = s - P31; return t < s ? t : s;
}
// Product by 2^k, 0 <= k <= 30: 2^31 = 1 (mod p), so it is the rotation of the 31-bit word by k — shift, shift, and-or: 3 instructions where the
// general product needs 6. a < 2^31; a canonical value (not all 31 bits set) stays canonical. The inverse transforms scale by 2^-n = 2^((31 - n % 31) % 31).
BF_HD u32 m_mul_pow2(u32 a, u32 k) { return ((a << k) & P31) | (a >> (31u - k)); }
BF_HD u32 m_sqr(u32 a) { return m_mul(a, a); }
BF_HD u32 m_inv_pow2(u32 n) { return 1u << ((31u - n % 31u) % 31u); }   // 2^-n: 2^31 = 1 (mod P)
BF_HD u32 m_pow(u32 b, u32 e) { u32 r = 1; while (e) { if (e & 1) r = m_mul(r, b); b = m_mul(b, b); e >>= 1; } return r; }
// x^(P-2) with the 2^31-3 addition chain (37 multiplications)
BF_HD u32 m_sqn(u32 x, int n) { for (int i = 0; i < n; i++) x = m_sqr(x); return x; }
BF_HD u32 m_inv(u32 x) {
    u32 t0 = m_mul(m_sqn(x, 2), x);            // x^5
    u32 t1 = m_mul(m_sqr(t0), t0);             // x^15
    u32 t2 = m_mul(m_sqn(t1, 3), t0);          // x^125
    u32 t3 = m_mul(m_sqr(t2), t0);             // x^255
    u32 t4 = m_mul(m_sqn(t3, 8), t3);          // x^65535
    u32 t5 = m_mul(m_sqn(t4, 8), t3);          // x^16777215
    return m_mul(m_sqn(t5, 7), t2);            // x^2147483645
}

struct C31 { u32 a, b; };
BF_HD C31 c_add(C31 x, C31 y) { return {m_add(x.a, y.a), m_add(x.b, y.b)}; }
BF_HD C31 c_sub(C31 x, C31 y) { return {m_sub(x.a, y.a), m_sub(x.b, y.b)}; }
BF_HD C31 c_neg(C31 x) { return {m_neg(x.a), m_neg(x.b)}; }
BF_HD C31 c_mul(C31 x, C31 y) { return {m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a))}; }
BF_HD C31 c_mulm(C31 x, u32 y) { return {m_mul(x.a, y), m_mul(x.b, y)}; }
BF_HD C31 c_mulR(C31 x) { return {m_sub(m_add(x.a, x.a), x.b), m_add(x.a, m_add(x.b, x.b))}; }  // * (2 + i)
BF_HD C31 c_inv(C31 x) { u32 n = m_inv(m_add(m_sqr(x.a), m_sqr(x.b))); return {m_mul(x.a, n), m_neg(m_mul(x.b, n))}; }

struct Q31 { C31 a, b; };
BF_HD Q31 q_make(u32 a0, u32 a1, u32 a2, u32 a3) { return {{a0, a1}, {a2, a3}}; }
BF_HD Q31 q_zero() { return {{0, 0}, {0, 0}}; }
BF_HD Q31 q_one() { return {{1, 0}, {0, 0}}; }
BF_HD Q31 q_from_m(u32 x) { return {{x, 0}, {0, 0}}; }
BF_HD Q31 q_add(Q31 x, Q31 y) { return {c_add(x.a, y.a), c_add(x.b, y.b)}; }
BF_HD Q31 q_sub(Q31 x, Q31 y) { return {c_sub(x.a, y.a), c_sub(x.b, y.b)}; }
BF_HD Q31 q_neg(Q31 x) { return {c_neg(x.a), c_neg(x.b)}; }
BF_HD Q31 q_mul(Q31 x, Q31 y) { return {c_add(c_mul(x.a, y.a), c_mulR(c_mul(x.b, y.b))), c_add(c_mul(x.a, y.b), c_mul(x.b, y.a))}; }
BF_HD Q31 q_mulm(Q31 x, u32 y) { return {c_mulm(x.a, y), c_mulm(x.b, y)}; }
// x * y for a y that is a constant of the kernel (the FRI folds: alpha and alpha^2). Multiplication by y is a 4 x 4 matrix over M31:
//   out0 = a0 c0 - a1 c1 + b0 e0 - b1 e1     out1 = a0 c1 + a1 c0 + b0 e1 + b1 e0          (y = (c0 + c1 i) + (c2 + c3 i) u,
//   out2 = a0 c2 - a1 c3 + b0 c0 - b1 c1     out3 = a0 c3 + a1 c2 + b0 c1 + b1 c0           e = (2 + i)(c2 + c3 i), x = (a0 + a1 i) + (b0 + b1 i) u)
// With the negated entries kept as p - c, every component is a sum of FOUR products of values <= p: < 2^64, one u64 accumulator and no
// intermediate reduction, then one reduction through 2^32 = 2 (mod p). 16 multiply-adds + 4 x 6 instructions instead of 16 x 6 + 12 x 3
// for q_mul (r04: the fold kernels turned out VALU-bound, 0.86 of the issue slots: profiles/r04_field_kernels_clock.txt).
struct QConst { u32 c0, c1, c2, c3, e0, e1, nc1, nc3, ne1; };
BF_HD QConst q_const(Q31 y) {
    const C31 e = c_mulR(y.b);
    return {y.a.a, y.a.b, y.b.a, y.b.b, e.a, e.b, P31 - y.a.b, P31 - y.b.b, P31 - e.b};
}
BF_HD u32 m_red4(u64 x) {      // any u64 -> canonical: hi * 2^32 + lo = 2 hi + lo < 2^34, folded once more to < 2^31 + 8
    const u64 t = ((x >> 32) << 1) + (u32)x;
    const u32 s = ((u32)t & P31) + (u32)(t >> 31); const u32 d = s - P31; return d < s ? d : s;
}
BF_HD Q31 q_mul_const(Q31 x, const QConst& k) {
    const u64 a0 = x.a.a, a1 = x.a.b, b0 = x.b.a, b1 = x.b.b;
    return q_make(m_red4(a0 * k.c0 + a1 * k.nc1 + b0 * k.e0 + b1 * k.ne1), m_red4(a0 * k.c1 + a1 * k.c0 + b0 * k.e1 + b1 * k.e0),
                  m_red4(a0 * k.c2 + a1 * k.nc3 + b0 * k.c0 + b1 * k.nc1), m_red4(a0 * k.c3 + a1 * k.c2 + b0 * k.c1 + b1 * k.c0));
}
BF_HD Q31 q_mulc(Q31 x, C31 y) { return {c_mul(x.a, y), c_mul(x.b, y)}; }
BF_HD Q31 q_addm(Q31 x, u32 y) { x.a.a = m_add(x.a.a, y); return x; }
BF_HD Q31 q_subm(Q31 x, u32 y) { x.a.a = m_sub(x.a.a, y); return x; }
BF_HD Q31 q_conj(Q31 x) { return {x.a, c_neg(x.b)}; }
BF_HD Q31 q_inv(Q31 x) { C31 d = c_inv(c_sub(c_mul(x.a, x.a), c_mulR(c_mul(x.b, x.b)))); return {c_mul(x.a, d), c_neg(c_mul(x.b, d))}; }
BF_HD bool q_eq(Q31 x, Q31 y) { return x.a.a == y.a.a && x.a.b == y.a.b && x.b.a == y.b.a && x.b.b == y.b.b; }
BF_HD bool q_is_zero(Q31 x) { return (x.a.a | x.a.b | x.b.a | x.b.b) == 0; }
BF_HD Q31 q_pow(Q31 b, u64 e) { Q31 r = q_one(); while (e) { if (e & 1) r = q_mul(r, b); b = q_mul(b, b); e >>= 1; } return r; }

BF_HD u32 bit_rev(u32 i, u32 log) {
#if defined(__HIP_DEVICE_COMPILE__)
    return log ? (__brev(i) >> (32 - log)) : 0;
#else
    u32 r = 0; for (u32 k = 0; k < log; k++) r |= ((i >> k) & 1u) << (log - 1 - k); return r;
#endif
}

}  // namespace bf
