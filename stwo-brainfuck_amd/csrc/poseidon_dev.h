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
// Montgomery product a * b * 2^-256 mod p (CIOS); inputs < p, output < p.
__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b, const u32* __restrict__ P) {
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { c += (u64)t[j] + (u64)a.l[j] * b.l[i]; t[j] = (u32)c; c >>= 32; }
        c += t[8]; t[8] = (u32)c; t[9] = (u32)(c >> 32);
        // reduction step: m = -t0 (p = 1 mod 2^32); t = (t + m * p) >> 32; p has limbs {1, 0, 0, 0, 0, 0, 17, 2^27}
        u32 m = 0u - t[0];
        c = (u64)t[0] + m;              // low word becomes 0, carry = (t0 != 0)
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 6; j++) { c += t[j]; t[j - 1] = (u32)c; c >>= 32; }
        c += (u64)t[6] + (u64)m * 17u; t[5] = (u32)c; c >>= 32;
        c += (u64)t[7] + (u64)m * 0x08000000u; t[6] = (u32)c; c >>= 32;
        c += t[8]; t[7] = (u32)c; c >>= 32;
        t[8] = t[9] + (u32)c; t[9] = 0;
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    // conditional subtraction (t < 2p)
    Fe s; u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)r.l[i] - P[i] - br; s.l[i] = (u32)d; br = d >> 63; }
    bool ge = t[8] != 0 || br == 0;
    return ge ? s : r;
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
