// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). PARITY UNPINNED beyond the public Hades known-answer vector.
// Stark-field arithmetic, the Hades permutation and the Poseidon hashes of starknet-crypto 0.6.2 (Cargo.lock:821-864) as used by stwo's
// Poseidon252MerkleHasher (core/vcs/poseidon252_merkle.rs) and Poseidon252Channel (core/channel/poseidon252.rs) — the MerkleChannel of
// BASELINE.json config 5, which the reference itself never instantiates (brainfuck_air/mod.rs:56 fixes Blake2sMerkleChannel).
// Independent of the product's host/felt252.h and poseidon.hip: generic word-serial Montgomery reduction over 4 x 64-bit limbs, round
// constants derived at start-up from their definition sha256("Hades" + i) mod p with this file's own SHA-256. Checked against the
// big-integer Python restatement oracle/poseidon252.py in tests/test_oracle_poseidon.py.
#pragma once
#include "field.h"
#include <cstring>
#include <string>
#include <vector>

namespace orc {

struct Felt {
    u64 w[4];   // Montgomery form x * 2^256 mod p, little-endian limbs
    static const u64* modulus() { static const u64 p[4] = {1ull, 0ull, 0ull, 0x0800000000000011ull}; return p; }   // 2^251 + 17 * 2^192 + 1
    static bool ge_p(const u64 a[4]) { const u64* p = modulus(); for (int i = 3; i >= 0; i--) if (a[i] != p[i]) return a[i] > p[i]; return true; }
    static void sub_p(u64 a[4]) { const u64* p = modulus(); unsigned __int128 b = 0; for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)a[i] - p[i] - b; a[i] = (u64)d; b = (d >> 64) & 1; } }
    Felt operator+(const Felt& o) const {
        Felt r; unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) { c += (unsigned __int128)w[i] + o.w[i]; r.w[i] = (u64)c; c >>= 64; }
        if (ge_p(r.w)) sub_p(r.w);
        return r;
    }
    Felt operator-(const Felt& o) const {
        Felt r; unsigned __int128 b = 0;
        for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)w[i] - o.w[i] - b; r.w[i] = (u64)d; b = (d >> 64) & 1; }
        if (b) { const u64* p = modulus(); unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)r.w[i] + p[i]; r.w[i] = (u64)c; c >>= 64; } }
        return r;
    }
    // REDC of the 512-bit product, one limb at a time with the generic factor m = t[i] * (-p^-1 mod 2^64); here -p^-1 = -1 since p = 1 mod 2^64.
    Felt operator*(const Felt& o) const {
        u64 t[9] = {0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) { c += (unsigned __int128)w[i] * o.w[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
            t[i + 4] = (u64)c;
        }
        const u64* p = modulus();
        for (int i = 0; i < 4; i++) {
            const u64 m = (u64)0 - t[i];
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) { c += (unsigned __int128)m * p[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
            for (int k = i + 4; c && k < 9; k++) { c += t[k]; t[k] = (u64)c; c >>= 64; }
        }
        Felt r = {{t[4], t[5], t[6], t[7]}};
        if (t[8] || ge_p(r.w)) sub_p(r.w);
        return r;
    }
    static Felt raw(u64 a, u64 b, u64 c, u64 d) { Felt f = {{a, b, c, d}}; return f; }
    static Felt r2() {   // 2^512 mod p by 512 modular doublings of 1
        static const Felt v = [] { Felt x = raw(1, 0, 0, 0); for (int i = 0; i < 512; i++) x = x + x; return x; }();   // operator+ is plain modular addition on raw values
        return v;
    }
    static Felt from_canonical(const u64 c[4]) { return raw(c[0], c[1], c[2], c[3]) * r2(); }
    static Felt from_u64(u64 v) { return raw(v, 0, 0, 0) * r2(); }
    void to_canonical(u64 out[4]) const { Felt c = *this * raw(1, 0, 0, 0); memcpy(out, c.w, 32); }
    static Felt from_le_bytes(const u8 b[32]) { u64 c[4]; memcpy(c, b, 32); return from_canonical(c); }
    void to_le_bytes(u8 b[32]) const { u64 c[4]; to_canonical(c); memcpy(b, c, 32); }
};

// ---- SHA-256 (FIPS 180-4), only to derive the round constants from their published definition --------------------------------------
static inline void sha256(const u8* msg, size_t len, u8 out[32]) {
    static const u32 K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
        0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
        0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
        0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::vector<u8> m(msg, msg + len);
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 7; i >= 0; i--) m.push_back((u8)(((u64)len * 8) >> (8 * i)));
    auto rr = [](u32 x, int n) { return (x >> n) | (x << (32 - n)); };
    for (size_t o = 0; o < m.size(); o += 64) {
        u32 w[64];
        for (int i = 0; i < 16; i++) w[i] = ((u32)m[o + 4 * i] << 24) | ((u32)m[o + 4 * i + 1] << 16) | ((u32)m[o + 4 * i + 2] << 8) | m[o + 4 * i + 3];
        for (int i = 16; i < 64; i++) { u32 s0 = rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10); w[i] = w[i - 16] + s0 + w[i - 7] + s1; }
        u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            u32 t1 = hh + (rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            u32 t2 = (rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)(h[i] >> 24); out[4 * i + 1] = (u8)(h[i] >> 16); out[4 * i + 2] = (u8)(h[i] >> 8); out[4 * i + 3] = (u8)h[i]; }
}

// Round constants: sha256("Hades" + decimal(i)) read as a big-endian integer, reduced mod p; 3 per round, 8 + 83 rounds.
static inline const std::vector<Felt>& hades_round_keys() {
    static const std::vector<Felt> keys = [] {
        std::vector<Felt> k;
        for (int i = 0; i < 91 * 3; i++) {
            std::string s = "Hades" + std::to_string(i);
            u8 d[32]; sha256((const u8*)s.data(), s.size(), d);
            // value mod p: fold the 256-bit big-endian integer in byte by byte with modular doubling/addition on raw (non-Montgomery) values
            Felt acc = Felt::raw(0, 0, 0, 0);
            for (int b = 0; b < 32; b++) {
                for (int q = 0; q < 8; q++) acc = acc + acc;
                acc = acc + Felt::raw(d[b], 0, 0, 0);
            }
            k.push_back(Felt::from_canonical(acc.w));
        }
        return k;
    }();
    return keys;
}

static inline void hades_permutation(Felt s[3]) {
    const std::vector<Felt>& ark = hades_round_keys();
    for (int r = 0; r < 91; r++) {
        for (int k = 0; k < 3; k++) s[k] = s[k] + ark[3 * r + k];
        const bool full = r < 4 || r >= 87;
        for (int k = full ? 0 : 2; k < 3; k++) s[k] = s[k] * s[k] * s[k];
        // MixLayer, MDS [[3,1,1],[1,-1,1],[1,1,-2]]
        Felt t = s[0] + s[1] + s[2];
        Felt a = t + s[0] + s[0], b = t - s[1] - s[1], c = t - s[2] - s[2] - s[2];
        s[0] = a; s[1] = b; s[2] = c;
    }
}
static inline Felt poseidon_hash(const Felt& x, const Felt& y) { Felt s[3] = {x, y, Felt::from_u64(2)}; hades_permutation(s); return s[0]; }
static inline Felt poseidon_hash_many(const std::vector<Felt>& v) {
    Felt s[3] = {Felt::raw(0, 0, 0, 0), Felt::raw(0, 0, 0, 0), Felt::raw(0, 0, 0, 0)};
    const Felt one = Felt::from_u64(1);
    size_t n = v.size();
    for (size_t i = 0; i + 1 < n; i += 2) { s[0] = s[0] + v[i]; s[1] = s[1] + v[i + 1]; hades_permutation(s); }
    if (n % 2 == 1) s[0] = s[0] + v[n - 1];
    s[n % 2] = s[n % 2] + one;
    hades_permutation(s);
    return s[0];
}
// cur * 2^31 + y for M31 values y (stwo folds column values and channel felts this way)
static inline Felt fold_m31(const Felt& cur, u32 y) { return cur * Felt::from_u64(u64(1) << 31) + Felt::from_u64(y); }

}  // namespace orc
