// Host-side Blake2s-256 and the Fiat–Shamir channel (the transcript is tiny and serial, so it stays on the host; only roots,
// sampled values and the FRI last layer cross PCIe). Mirrors stwo `Blake2sChannel` / `Blake2sMerkleChannel` as used at
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:485 (default()), :581/:721 (claim mixing via components/mod.rs:82,133),
// :591 (draw_felts through LookupElements::draw) and inside prover::prove (:732).
#pragma once
#include "../m31.h"
#include "felt252.h"
#include <cstring>
#include <vector>
#include <algorithm>

namespace bf {

struct Hash32 { u8 b[32]; bool operator==(const Hash32& o) const { return memcmp(b, o.b, 32) == 0; } };

namespace b2s {
static const u32 IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const u8 SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
inline u32 ror(u32 x, int r) { return (x >> r) | (x << (32 - r)); }
inline void compress(u32 h[8], const u32 m[16], u32 t0, u32 t1, u32 f0, u32 f1) {
    u32 v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t0; v[13] ^= t1; v[14] ^= f0; v[15] ^= f1;
    auto G = [&](int a, int b, int c, int d, u32 x, u32 y) {
        v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 12);
        v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 8); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 7);
    };
    for (int r = 0; r < 10; r++) {
        const u8* s = SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
// one-shot hash of a byte string
inline Hash32 hash(const u8* data, size_t len) {
    u32 h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010020u;
    u64 t = 0;
    u32 m[16];
    while (len > 64) { memcpy(m, data, 64); t += 64; compress(h, m, (u32)t, (u32)(t >> 32), 0, 0); data += 64; len -= 64; }
    u8 last[64] = {0};
    memcpy(last, data, len);
    memcpy(m, last, 64);
    t += len;
    compress(h, m, (u32)t, (u32)(t >> 32), 0xFFFFFFFFu, 0);
    Hash32 out; memcpy(out.b, h, 32);
    return out;
}
}  // namespace b2s

// Blake2sMerkleHasher::hash_node on the host (verifier; the prover hashes on the GPU, merkle.hip). conv = Conventions::merkle_node_hash.
// Poseidon252MerkleHasher::hash_node: poseidon_hash_many([left, right]? ++ blocks of 8 M31 packed as w = w * 2^31 + v, zero padded)
inline Hash32 host_hash_node_poseidon(const Hash32* l, const Hash32* r, const u32* vals, size_t n) {
    std::vector<Fe252> v;
    if (l) { v.push_back(fe252::from_le_bytes(l->b)); v.push_back(fe252::from_le_bytes(r->b)); }
    for (size_t o = 0; o < n; o += 8) v.push_back(fe252::pack_m31_block(vals + o, std::min<size_t>(8, n - o)));
    Hash32 out; fe252::to_le_bytes(fe252::hash_many(v.data(), v.size()), out.b);
    return out;
}
inline Hash32 host_hash_node(const Hash32* l, const Hash32* r, const u32* vals, size_t n, const Conventions& cv) {
    if (cv.merkle_channel == 1) return host_hash_node_poseidon(l, r, vals, n);
    if (cv.merkle_node_hash == 0) {   // zero state, one raw compression per 64-byte block: children first, then the column words zero padded to 16
        u32 st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[16];
        if (l) { memcpy(m, l->b, 32); memcpy(m + 8, r->b, 32); b2s::compress(st, m, 0, 0, 0, 0); }
        for (size_t o = 0; o < n; o += 16) {
            size_t take = std::min<size_t>(16, n - o);
            memset(m, 0, sizeof m); memcpy(m, vals + o, 4 * take);
            b2s::compress(st, m, 0, 0, 0, 0);
        }
        Hash32 out; memcpy(out.b, st, 32); return out;
    }
    std::vector<u8> buf((l ? 64 : 0) + 4 * n);
    if (l) { memcpy(buf.data(), l->b, 32); memcpy(buf.data() + 32, r->b, 32); }
    if (n) memcpy(buf.data() + (l ? 64 : 0), vals, 4 * n);
    return b2s::hash(buf.data(), buf.size());
}

// Blake2sChannel (kind 0) or Poseidon252Channel (kind 1: digest = felt252 kept as its canonical 32 little-endian bytes; stwo
// core/channel/poseidon252.rs). One type, so that prover and verifier code is written once.
struct Channel {
    Hash32 digest; u32 n_sent = 0;
    u32 mix_u64_conv = 0;   // Conventions::mix_u64 (Blake2s channel only)
    u32 kind = 0;           // Conventions::merkle_channel
    Channel() { memset(digest.b, 0, 32); }
    explicit Channel(const Conventions& cv) : mix_u64_conv(cv.mix_u64), kind(cv.merkle_channel) { memset(digest.b, 0, 32); }
    void update(const Hash32& d) { digest = d; n_sent = 0; }
    void update_fe(const Fe252& x) { Hash32 d; fe252::to_le_bytes(x, d.b); update(d); }
    Fe252 digest_fe() const { return fe252::from_le_bytes(digest.b); }
    // MerkleChannel::mix_root: Blake2s(digest || root) / poseidon_hash(digest, root)
    void mix_root(const Hash32& root) {
        if (kind == 1) { update_fe(fe252::hash2(digest_fe(), fe252::from_le_bytes(root.b))); return; }
        u8 buf[64]; memcpy(buf, digest.b, 32); memcpy(buf + 32, root.b, 32); update(b2s::hash(buf, 64));
    }
    void mix_felts(const Q31* f, size_t n) {
        if (kind == 1) {
            // poseidon_hash_many([digest] ++ one felt252 per chunk of 2 secure felts: fold of their M31 coordinates, cur = cur * 2^31 + y)
            std::vector<Fe252> res; res.push_back(digest_fe());
            for (size_t i = 0; i < n; i += 2) {
                Fe252 w = fe252::zero();      // canonical integer < 2^248 while folding
                for (size_t k = i; k < std::min(n, i + 2); k++) {
                    const u32 y[4] = {f[k].a.a, f[k].a.b, f[k].b.a, f[k].b.b};
                    for (int q = 0; q < 4; q++) {
                        w.l[3] = (w.l[3] << 31) | (w.l[2] >> 33); w.l[2] = (w.l[2] << 31) | (w.l[1] >> 33); w.l[1] = (w.l[1] << 31) | (w.l[0] >> 33); w.l[0] = (w.l[0] << 31) | y[q];
                    }
                }
                res.push_back(fe252::to_mont(w));
            }
            update_fe(fe252::hash_many(res.data(), res.size()));
            return;
        }
        std::vector<u8> buf(32 + 16 * n);
        memcpy(buf.data(), digest.b, 32);
        for (size_t i = 0; i < n; i++) { u32 w[4] = {f[i].a.a, f[i].a.b, f[i].b.a, f[i].b.b}; memcpy(buf.data() + 32 + 16 * i, w, 16); }
        update(b2s::hash(buf.data(), buf.size()));
    }
    void mix_u64(u64 v) {
        if (kind == 1) { update_fe(fe252::hash2(digest_fe(), fe252::mont_from_u64(v))); return; }
        if (mix_u64_conv == 1) { u8 buf[64] = {0}; memcpy(buf, digest.b, 32); memcpy(buf + 32, &v, 8); update(b2s::hash(buf, 64)); return; }
        u32 h[8]; memcpy(h, digest.b, 32);
        u32 m[16] = {0}; m[0] = (u32)v; m[1] = (u32)(v >> 32);
        b2s::compress(h, m, 0, 0, 0, 0);
        Hash32 d; memcpy(d.b, h, 32); update(d);
    }
    // Poseidon252Channel::draw_felt252: poseidon_hash(digest, n_sent), canonical little-endian bytes
    Hash32 draw_felt252() { Hash32 r; fe252::to_le_bytes(fe252::hash2(digest_fe(), fe252::mont_from_u64(n_sent)), r.b); n_sent++; return r; }
    // Blake2s: 32 bytes; Poseidon252: the 31 low bytes of a drawn felt (repeated floor_div by 2^8)
    std::vector<u8> draw_random_bytes() {
        if (kind == 1) { Hash32 r = draw_felt252(); return std::vector<u8>(r.b, r.b + 31); }
        u8 buf[64] = {0}; memcpy(buf, digest.b, 32); memcpy(buf + 32, &n_sent, 4); n_sent++;
        Hash32 r = b2s::hash(buf, 64);
        return std::vector<u8>(r.b, r.b + 32);
    }
    void draw_base_felts(u32 out[8]) {
        if (kind == 1) {   // 8 limbs of 31 bits of one drawn felt252 (repeated floor_div by 2^31), each reduced mod P (no rejection)
            Hash32 r = draw_felt252();
            u64 l[4]; memcpy(l, r.b, 32);
            for (int i = 0; i < 8; i++) {
                const unsigned sh = 31 * i, limb = sh >> 6, off = sh & 63;
                u64 v = l[limb] >> off;
                if (off > 33 && limb + 1 < 4) v |= l[limb + 1] << (64 - off);
                const u32 x = (u32)(v & 0x7fffffffu);
                out[i] = x == P31 ? 0u : x;
            }
            return;
        }
        for (;;) {
            std::vector<u8> r = draw_random_bytes();
            u32 w[8]; memcpy(w, r.data(), 32);
            bool ok = true;
            for (int i = 0; i < 8; i++) ok = ok && w[i] < 2 * P31;
            if (!ok) continue;
            for (int i = 0; i < 8; i++) out[i] = w[i] >= P31 ? w[i] - P31 : w[i];
            return;
        }
    }
    Q31 draw_felt() { u32 f[8]; draw_base_felts(f); return q_make(f[0], f[1], f[2], f[3]); }
    void draw_two_felts(Q31& a, Q31& b) { u32 f[8]; draw_base_felts(f); a = q_make(f[0], f[1], f[2], f[3]); b = q_make(f[4], f[5], f[6], f[7]); }
    u32 trailing_zeros() const {
        if (kind == 1) {   // u128::from_le_bytes(first 16 bytes of digest.to_bytes_be()).trailing_zeros(): big-endian byte k = little-endian byte 31 - k
            for (int k = 0; k < 16; k++) { u8 b = digest.b[31 - k]; if (b) return 8 * k + (u32)__builtin_ctz(b); }
            return 128;
        }
        u64 lo, hi; memcpy(&lo, digest.b, 8); memcpy(&hi, digest.b + 8, 8);
        return lo ? (u32)__builtin_ctzll(lo) : hi ? 64 + (u32)__builtin_ctzll(hi) : 128;
    }
};

}  // namespace bf
