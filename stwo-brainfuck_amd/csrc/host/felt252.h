// Host-side Stark-field (felt252) arithmetic, Hades permutation and Poseidon hashes for the Poseidon252 Merkle-channel variant
// (BASELINE.json config 5: `Poseidon252MerkleChannel` = stwo core/vcs/poseidon252_merkle.rs + core/channel/poseidon252.rs over
// starknet-crypto 0.6.2, Cargo.lock:821-864). The reference itself fixes Blake2sMerkleChannel (brainfuck_air/mod.rs:56,486-487); this
// variant is an upstream capability — PARITY UNPINNED beyond the public Hades known-answer vector.
// Only the transcript (channel), proof-of-work search and the verifier hash on the host; the Merkle layers are hashed on the GPU
// (poseidon.hip), which shares the generated constant table (poseidon_constants.h, Montgomery form, R = 2^256).
#pragma once
#include "../m31.h"
#include "../poseidon_constants.h"
#include <cstring>
#include <vector>

namespace bf {

// p = 2^251 + 17 * 2^192 + 1. Values are kept in Montgomery form (x * 2^256 mod p) as 4 little-endian u64 limbs unless stated otherwise.
struct Fe252 { u64 l[4]; };

namespace fe252 {
static const u64 P[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull};
inline Fe252 from_u32_limbs(const u32* w) { Fe252 r; for (int i = 0; i < 4; i++) r.l[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32); return r; }
inline bool geq_p(const Fe252& a) { for (int i = 3; i >= 0; i--) { if (a.l[i] != P[i]) return a.l[i] > P[i]; } return true; }
inline Fe252 sub_p(const Fe252& a) { Fe252 r; unsigned __int128 br = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - P[i] - br; r.l[i] = (u64)d; br = (d >> 64) & 1; } return r; }
inline Fe252 add(const Fe252& a, const Fe252& b) {
    Fe252 r; unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
    return geq_p(r) ? sub_p(r) : r;            // a + b < 2p < 2^253: no carry out of 256 bits
}
inline Fe252 sub(const Fe252& a, const Fe252& b) {
    Fe252 r; unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - br; r.l[i] = (u64)d; br = (d >> 64) & 1; }
    if (br) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)r.l[i] + P[i]; r.l[i] = (u64)c; c >>= 64; } }
    return r;
}
// Montgomery product a * b * 2^-256 mod p. p = 1 mod 2^64, so the reduction factor of each step is m = -t0 and m * p touches limbs 0 and 3.
inline Fe252 mul(const Fe252& a, const Fe252& b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; j++) { c += (unsigned __int128)t[j] + (unsigned __int128)a.l[j] * b.l[i]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        const u64 m = 0 - t[0];
        c = (unsigned __int128)t[0] + m; c >>= 64;                       // low limb becomes zero
        c += t[1]; t[0] = (u64)c; c >>= 64;
        c += t[2]; t[1] = (u64)c; c >>= 64;
        c += (unsigned __int128)t[3] + (unsigned __int128)m * P[3]; t[2] = (u64)c; c >>= 64;
        c += t[4]; t[3] = (u64)c; c >>= 64;
        t[4] = t[5] + (u64)c; t[5] = 0;
    }
    Fe252 r = {{t[0], t[1], t[2], t[3]}};
    return (t[4] != 0 || geq_p(r)) ? sub_p(r) : r;
}
inline Fe252 zero() { return Fe252{{0, 0, 0, 0}}; }
inline Fe252 mont_one() { return from_u32_limbs(POSEIDON_R1); }
inline Fe252 to_mont(const Fe252& canonical) { return mul(canonical, from_u32_limbs(POSEIDON_R2)); }
inline Fe252 from_mont(const Fe252& x) { return mul(x, Fe252{{1, 0, 0, 0}}); }
inline Fe252 mont_from_u64(u64 v) { return to_mont(Fe252{{v, 0, 0, 0}}); }

// Hades permutation (state width 3, 4 + 83 + 4 rounds, x^3, MDS [[3,1,1],[1,-1,1],[1,1,-2]]) on Montgomery-form elements.
inline void hades(Fe252 s[3]) {
    for (int r = 0; r < 91; r++) {
        for (int k = 0; k < 3; k++) s[k] = add(s[k], from_u32_limbs(POSEIDON_ARK[3 * r + k]));
        const bool full = r < 4 || r >= 87;
        if (full) { s[0] = mul(mul(s[0], s[0]), s[0]); s[1] = mul(mul(s[1], s[1]), s[1]); }
        s[2] = mul(mul(s[2], s[2]), s[2]);
        Fe252 t = add(add(s[0], s[1]), s[2]);
        Fe252 d0 = add(s[0], s[0]), d1 = add(s[1], s[1]), d2 = add(add(s[2], s[2]), s[2]);
        s[0] = add(t, d0); s[1] = sub(t, d1); s[2] = sub(t, d2);
    }
}
// starknet-crypto poseidon_hash(x, y) = hades([x, y, 2])[0]
inline Fe252 hash2(const Fe252& x, const Fe252& y) { Fe252 s[3] = {x, y, mont_from_u64(2)}; hades(s); return s[0]; }
// starknet-crypto poseidon_hash_many: rate-2 sponge, padded with a single one
inline Fe252 hash_many(const Fe252* v, size_t n) {
    Fe252 s[3] = {zero(), zero(), zero()};
    size_t i = 0;
    for (; i + 1 < n; i += 2) { s[0] = add(s[0], v[i]); s[1] = add(s[1], v[i + 1]); hades(s); }
    if (n & 1) { s[0] = add(s[0], v[n - 1]); s[1] = add(s[1], mont_one()); }
    else s[0] = add(s[0], mont_one());
    hades(s);
    return s[0];
}
// canonical little-endian bytes (the 32-byte record the GPU kernels store: 8 LE u32 limbs) <-> Montgomery form
inline Fe252 from_le_bytes(const u8 b[32]) { Fe252 c; memcpy(c.l, b, 32); return to_mont(c); }
inline void to_le_bytes(const Fe252& x, u8 b[32]) { Fe252 c = from_mont(x); memcpy(b, c.l, 32); }
inline bool canonical_bytes_in_range(const u8 b[32]) { Fe252 c; memcpy(c.l, b, 32); return !geq_p(c); }
// block of up to 8 M31 values packed as w = w * 2^31 + v (zero padded): value k occupies bits [31 (7 - k), 31 (8 - k)); canonical -> Montgomery
inline Fe252 pack_m31_block(const u32* vals, size_t n) {
    Fe252 w = zero();
    for (size_t k = 0; k < 8; k++) {
        const u64 v = k < n ? vals[k] : 0;
        const unsigned sh = 31 * (7 - (unsigned)k), limb = sh >> 6, off = sh & 63;
        w.l[limb] |= v << off;
        if (off > 33 && limb + 1 < 4) w.l[limb + 1] |= v >> (64 - off);
    }
    return to_mont(w);
}
}  // namespace fe252

}  // namespace bf
