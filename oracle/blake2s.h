// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header).
// Blake2s-256 (RFC 7693; pinned by the RFC's "abc" and empty-string KATs in tests/) and the Fiat–Shamir channel of
// stwo@31e8dbc `core/channel/blake2s.rs`, `core/vcs/blake2_hash.rs`, `core/vcs/blake2s_ref.rs` (PARITY UNPINNED for the
// channel byte layouts). Reference call sites: Blake2sChannel::default() crates/brainfuck_prover/src/brainfuck_air/mod.rs:485,
// mix_u64 crates/brainfuck_prover/src/components/mod.rs:133, mix_felts components/mod.rs:82, draw_felts via
// LookupElements::draw (brainfuck_air/mod.rs:158-164).
#pragma once
#include "field.h"
#include "poseidon252.h"
#include <cstring>
#include <vector>

namespace orc {

static const u32 BLAKE2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const u8 BLAKE2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static inline u32 rotr32(u32 x, int r) { return (x >> r) | (x << (32 - r)); }

// Raw compression function F (stwo blake2s_ref::compress(h, m, t0, t1, f0, f1)).
static inline void blake2s_compress(u32 h[8], const u32 m[16], u32 t0, u32 t1, u32 f0, u32 f1) {
    u32 v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = BLAKE2S_IV[i]; }
    v[12] ^= t0; v[13] ^= t1; v[14] ^= f0; v[15] ^= f1;
#define ORC_G(a, b, c, d, x, y)                   \
    v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 12); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);  \
    v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const u8* s = BLAKE2S_SIGMA[r];
        ORC_G(0, 4, 8, 12, m[s[0]], m[s[1]]) ORC_G(1, 5, 9, 13, m[s[2]], m[s[3]])
        ORC_G(2, 6, 10, 14, m[s[4]], m[s[5]]) ORC_G(3, 7, 11, 15, m[s[6]], m[s[7]])
        ORC_G(0, 5, 10, 15, m[s[8]], m[s[9]]) ORC_G(1, 6, 11, 12, m[s[10]], m[s[11]])
        ORC_G(2, 7, 8, 13, m[s[12]], m[s[13]]) ORC_G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef ORC_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

struct Hash32 {
    u8 b[32];
    bool operator==(const Hash32& o) const { return memcmp(b, o.b, 32) == 0; }
    bool operator!=(const Hash32& o) const { return !(*this == o); }
};

// Streaming Blake2s-256 (no key), as the `blake2` crate's Blake2s256 used by stwo's Blake2sHasher.
struct Blake2s {
    u32 h[8];
    u8 buf[64];
    size_t buflen = 0;
    u64 t = 0;
    Blake2s() {
        for (int i = 0; i < 8; i++) h[i] = BLAKE2S_IV[i];
        h[0] ^= 0x01010020u;  // digest_length = 32, fanout = depth = 1
    }
    void block(const u8* p, bool last) {
        u32 m[16];
        memcpy(m, p, 64);  // little-endian host
        blake2s_compress(h, m, (u32)t, (u32)(t >> 32), last ? 0xFFFFFFFFu : 0, 0);
    }
    void update(const void* data, size_t len) {
        const u8* p = (const u8*)data;
        while (len) {
            if (buflen == 64) { t += 64; block(buf, false); buflen = 0; }
            size_t take = std::min(len, size_t(64) - buflen);
            memcpy(buf + buflen, p, take);
            buflen += take; p += take; len -= take;
        }
    }
    Hash32 finalize() {
        t += buflen;
        memset(buf + buflen, 0, 64 - buflen);
        block(buf, true);
        Hash32 out;
        memcpy(out.b, h, 32);
        return out;
    }
    static Hash32 hash(const void* data, size_t len) { Blake2s s; s.update(data, len); return s.finalize(); }
};

// stwo core/channel/blake2s.rs Blake2sChannel, or — conventions().merkle_channel == 1 at construction — core/channel/poseidon252.rs
// Poseidon252Channel, whose felt252 digest is kept as its canonical 32 little-endian bytes.
struct Channel {
    Hash32 digest;
    u32 n_challenges = 0, n_sent = 0;
    bool poseidon = false;
    Channel() : poseidon(conventions().merkle_channel == 1) { memset(digest.b, 0, 32); }
    void update_digest(const Hash32& d) { digest = d; n_challenges++; n_sent = 0; }
    void update_felt(const Felt& f) { Hash32 d; f.to_le_bytes(d.b); update_digest(d); }
    Felt digest_felt() const { return Felt::from_le_bytes(digest.b); }
    // Blake2sMerkleChannel::mix_root: digest = H(digest || root); Poseidon252MerkleChannel::mix_root: digest = poseidon_hash(digest, root)
    void mix_root(const Hash32& root) {
        if (poseidon) { update_felt(poseidon_hash(digest_felt(), Felt::from_le_bytes(root.b))); return; }
        Blake2s s; s.update(digest.b, 32); s.update(root.b, 32); update_digest(s.finalize());
    }
    void mix_felts(const QM31* f, size_t n) {
        if (poseidon) {   // poseidon_hash_many([digest] ++ one felt per chunk of 2 secure felts, folded cur = cur * 2^31 + coordinate)
            std::vector<Felt> res; res.push_back(digest_felt());
            for (size_t i = 0; i < n; i += 2) {
                Felt cur = Felt::raw(0, 0, 0, 0);
                for (size_t k = i; k < std::min(n, i + 2); k++) { auto a = f[k].to_u32(); for (int q = 0; q < 4; q++) cur = fold_m31(cur, a[q]); }
                res.push_back(cur);
            }
            update_felt(poseidon_hash_many(res));
            return;
        }
        Blake2s s; s.update(digest.b, 32);
        for (size_t i = 0; i < n; i++) { auto a = f[i].to_u32(); s.update(a.data(), 16); }
        update_digest(s.finalize());
    }
    // mix_u64: raw compression of [n_lo, n_hi, 0...] with h = digest words, t = f = 0.
    void mix_u64(u64 nonce) {
        if (poseidon) { update_felt(poseidon_hash(digest_felt(), Felt::from_u64(nonce))); return; }
        if (conventions().mix_u64 == 1) {
            u8 in[64]; memcpy(in, digest.b, 32); memset(in + 32, 0, 32); memcpy(in + 32, &nonce, 8);
            update_digest(Blake2s::hash(in, 64));
            return;
        }
        u32 h[8]; memcpy(h, digest.b, 32);
        u32 m[16] = {0}; m[0] = (u32)nonce; m[1] = (u32)(nonce >> 32);
        blake2s_compress(h, m, 0, 0, 0, 0);
        Hash32 d; memcpy(d.b, h, 32);
        update_digest(d);
    }
    // Poseidon252Channel::draw_felt252 as canonical integer limbs
    void draw_felt252(u64 out[4]) { Felt r = poseidon_hash(digest_felt(), Felt::from_u64(n_sent)); n_sent++; r.to_canonical(out); }
    std::vector<u8> draw_random_bytes() {
        if (poseidon) {   // 31 times: byte = cur mod 2^8, cur = cur div 2^8
            u64 c[4]; draw_felt252(c);
            std::vector<u8> out(31);
            for (int i = 0; i < 31; i++) out[i] = (u8)(c[i / 8] >> (8 * (i % 8)));
            return out;
        }
        u8 in[64]; memcpy(in, digest.b, 32); memset(in + 32, 0, 32);
        memcpy(in + 32, &n_sent, 4);  // counter as LE bytes, zero padded to 32
        n_sent++;
        Hash32 r = Blake2s::hash(in, 64);
        return std::vector<u8>(r.b, r.b + 32);
    }
    void draw_base_felts(M31 out[8]) {
        if (poseidon) {   // 8 times: limb = cur mod 2^31, cur = cur div 2^31; BaseField::reduce(limb)
            u64 c[4]; draw_felt252(c);
            for (int i = 0; i < 8; i++) {
                u32 limb = (u32)(c[0] & 0x7fffffffull);
                out[i] = M31(m31_reduce(limb));
                for (int k = 0; k < 4; k++) c[k] = (c[k] >> 31) | (k + 1 < 4 ? c[k + 1] << 33 : 0);
            }
            return;
        }
        for (;;) {
            std::vector<u8> r = draw_random_bytes();
            u32 w[8]; memcpy(w, r.data(), 32);
            bool ok = true;
            for (int i = 0; i < 8; i++) if (w[i] >= 2 * P) ok = false;
            if (!ok) continue;
            for (int i = 0; i < 8; i++) out[i] = M31(m31_reduce(w[i]));
            return;
        }
    }
    QM31 draw_felt() { M31 f[8]; draw_base_felts(f); return QM31::from_m31_array(f); }
    std::vector<QM31> draw_felts(size_t n) {
        std::vector<QM31> out;
        M31 f[8]; int have = 0, pos = 0;
        while (out.size() < n) {
            M31 c[4];
            for (int k = 0; k < 4; k++) {
                if (pos == have) { draw_base_felts(f); have = 8; pos = 0; }
                c[k] = f[pos++];
            }
            out.push_back(QM31::from_m31_array(c));
        }
        return out;
    }
    // trailing zeros of the first 16 digest bytes read as a LE u128 (Poseidon: of digest.to_bytes_be())
    u32 trailing_zeros() const {
        if (poseidon) {
            u8 be[32]; for (int i = 0; i < 32; i++) be[i] = digest.b[31 - i];
            unsigned __int128 v = 0; for (int i = 15; i >= 0; i--) v = (v << 8) | be[i];
            if (!v) return 128;
            u32 tz = 0; while (!(v & 1)) { v >>= 1; tz++; }
            return tz;
        }
        u64 lo, hi; memcpy(&lo, digest.b, 8); memcpy(&hi, digest.b + 8, 8);
        if (lo) return (u32)__builtin_ctzll(lo);
        if (hi) return 64 + (u32)__builtin_ctzll(hi);
        return 128;
    }
};

// GrindOps for CpuBackend: smallest nonce such that mix_u64(nonce) leaves >= pow_bits trailing zeros.
static inline u64 grind(const Channel& ch, u32 pow_bits) {
    for (u64 nonce = 0;; nonce++) { Channel c = ch; c.mix_u64(nonce); if (c.trailing_zeros() >= pow_bits) return nonce; }
}

}  // namespace orc
