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
// 96-bit multiply-accumulate (hi : lo) += a * b: v_mad_u64_u32 adds the 64-bit product into `lo` and leaves the carry in VCC, one
// v_addc_co_u32 folds it into `hi` — two instructions per limb product (the compiler's own u64 formulation of a carry-save row takes
// three to four plus register moves: measured 2.6x more issue slots per field product).
__device__ __forceinline__ void mac96(u64& lo, u32& hi, u32 a, u32 b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
// Montgomery product a * b * 2^-256 mod p; inputs < p, output < p.
// Product scanning (Comba): column k of the 512-bit product is the 96-bit sum of its a_i * b_(k-i); then a separate Montgomery
// reduction sweep: p = 1 + 17 * 2^192 + 2^251 = 1 (mod 2^32), so the factor of step i is m = -T[i] and m * p only adds m at limb i
// (which zeroes it), 17 m at limb i + 6 and m * 2^27 across limbs i + 7, i + 8.
__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b, const u32* __restrict__ P) {
    u32 T[16];
    u64 lo = 0; u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { const int j = k - i; if (j >= 0 && j < 8) mac96(lo, hi, a.l[i], b.l[j]); }
        T[k] = (u32)lo; lo = (lo >> 32) | ((u64)hi << 32); hi = 0;
    }
    T[15] = (u32)lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 m = 0u - T[i];
        u32 c = T[i] != 0 ? 1u : 0u;            // carry out of T[i] + m (the limb itself becomes zero)
#pragma unroll
        for (int j = i + 1; j < i + 6; j++) T[j] = __builtin_addc(T[j], 0u, c, &c);
        const u64 m17 = (u64)m * 17u;
        u32 c2;
        T[i + 6] = __builtin_addc(T[i + 6], (u32)m17, c, &c);
        T[i + 7] = __builtin_addc(T[i + 7], (u32)(m17 >> 32), c, &c);
        T[i + 7] = __builtin_addc(T[i + 7], m << 27, 0u, &c2);
        if (i + 8 < 16) {
            T[i + 8] = __builtin_addc(T[i + 8], m >> 5, c, &c);
            u32 c3; T[i + 8] = __builtin_addc(T[i + 8], 0u, c2, &c3); c += c3;
#pragma unroll
            for (int j = i + 9; j < 16; j++) T[j] = __builtin_addc(T[j], 0u, c, &c);
        }
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = T[8 + i];
    // conditional subtraction (result < 2p)
    Fe s; u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = __builtin_subc(r.l[i], P[i], br, &br);
    return br ? r : s;
}

__device__ __forceinline__ void hades(Fe s[3], const u32* __restrict__ ark, const u32* __restrict__ P) {
    for (int r = 0; r < 91; r++) {
        const u32* k = ark + (size_t)r * 24;
        s[0] = fe_add(s[0], fe_load_const(k), P);
        s[1] = fe_add(s[1], fe_load_const(k + 8), P);
        s[2] = fe_add(s[2], fe_load_const(k + 16), P);
        bool full = r < 4 || r >= 87;
        if (full) {
            Fe q0 = fe_mul(s[0], s[0], P), q1 = fe_mul(s[1], s[1], P);
            s[0] = fe_mul(q0, s[0], P); s[1] = fe_mul(q1, s[1], P);
        }
        Fe q2 = fe_mul(s[2], s[2], P);
        s[2] = fe_mul(q2, s[2], P);
        // MDS: t = s0 + s1 + s2; (t + 2 s0, t - 2 s1, t - 3 s2)
        Fe t = fe_add(fe_add(s[0], s[1], P), s[2], P);
        Fe d0 = fe_add(s[0], s[0], P), d1 = fe_add(s[1], s[1], P), d2 = fe_add(fe_add(s[2], s[2], P), s[2], P);
        s[0] = fe_add(t, d0, P); s[1] = fe_sub(t, d1, P); s[2] = fe_sub(t, d2, P);
    }
}

struct Sponge {
    Fe s[3]; u32 count;
    const u32* ark; const u32* P; const u32* R1;
    __device__ __forceinline__ void init(const u32* ark_, const u32* P_, const u32* R1_) {
        ark = ark_; P = P_; R1 = R1_; count = 0;
        for (int k = 0; k < 3; k++) for (int i = 0; i < 8; i++) s[k].l[i] = 0;
    }
    __device__ __forceinline__ void absorb(const Fe& v) {      // v in Montgomery form
        if ((count & 1) == 0) s[0] = fe_add(s[0], v, P);
        else { s[1] = fe_add(s[1], v, P); hades(s, ark, P); }
        count++;
    }
    __device__ __forceinline__ Fe finish() {                    // poseidon_hash_many padding: a single one
        Fe one = fe_load_const(R1);
        if (count & 1) s[1] = fe_add(s[1], one, P); else s[0] = fe_add(s[0], one, P);
        hades(s, ark, P);
        return s[0];
    }
};

}  // namespace bf
