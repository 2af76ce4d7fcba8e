// Blake2s mixed-degree Merkle layers for gfx950 — SURVEY.md §8 row a4.
// Replaces stwo `MerkleOps<Blake2sMerkleHasher>::commit_on_layer` reached from tree_builder.commit(channel),
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:500,583,723 (and the composition / FRI layer trees inside prover::prove, :732).
// node(i) absorbs left(32B) || right(32B) || LE-u32 value of every column of this layer's size at row i, 64 bytes per compression.
// Conventions::merkle_node_hash (m31.h) selects how: 0 (default) = stwo's Blake2sMerkleHasher::hash_node of the period — zero initial state,
// raw compress(state, block, 0,0,0,0) per block, column words zero padded to 16; 1 = RFC 7693 Blake2s-256 of the byte string (parameter
// block, byte counter, final flag). Both cost the same number of compressions; the kernels take the choice as an all-ones/zero mask `rfc`.
//
// One lane hashes one node. The message is streamed 64 bytes at a time through a fully unrolled compression (the sigma
// schedule is compile-time, so the 16 message words and the 16 state words stay in VGPRs; rotations by 16/8 lower to
// v_perm/v_alignbit). This kernel is integer-VALU bound (~1.0 k ops per 64-byte block), not HBM bound: see DESIGN.md.
// Column reads are coalesced (lane i reads cell i of each column; replicated columns read cell i >> 4), child hashes are read as
// 4 x 16 B per lane, hashes are stored as 2 x 16 B per lane in AoS [node][8 x u32] order (the order decommitment needs).
#include "kernels.h"
#include <stdexcept>

namespace bf {

__device__ __constant__ const u32 B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

__device__ __forceinline__ u32 rotr(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }

#define B2S_G(a, b, c, d, x, y) \
    a = a + b + (x); d = rotr(d ^ a, 16); c = c + d; b = rotr(b ^ c, 12); \
    a = a + b + (y); d = rotr(d ^ a, 8);  c = c + d; b = rotr(b ^ c, 7);

#define B2S_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2S_G(v0, v4, v8, v12, m[s0], m[s1]) B2S_G(v1, v5, v9, v13, m[s2], m[s3])           \
    B2S_G(v2, v6, v10, v14, m[s4], m[s5]) B2S_G(v3, v7, v11, v15, m[s6], m[s7])         \
    B2S_G(v0, v5, v10, v15, m[s8], m[s9]) B2S_G(v1, v6, v11, v12, m[s10], m[s11])       \
    B2S_G(v2, v7, v8, v13, m[s12], m[s13]) B2S_G(v3, v4, v9, v14, m[s14], m[s15])

__device__ __forceinline__ void blake2s_compress(u32 h[8], const u32 m[16], u32 t0, u32 f0) {
    u32 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
    u32 v12 = 0x510E527Fu ^ t0, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu ^ f0, v15 = 0x5BE0CD19u;
    B2S_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2S_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2S_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2S_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2S_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2S_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2S_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2S_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2S_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2S_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// One Merkle layer of 2^log nodes. prev == nullptr for the deepest layer. Requires has_prev || ncols > 0 ... or hashes the empty string.
// Replication-aware: when every input of a layer is replicated (row-granular columns and/or a replicated child layer), nodes
// i and i' with i >> out_shift == i' >> out_shift hash identical messages, so only 2^(log - out_shift) nodes are computed and
// stored; readers index `node >> shift`. Hash values are exactly those of the full layer.
// write-once data that the next launch (not this one) reads: non-temporal stores keep it from displacing the inputs in L2
__device__ __forceinline__ void store_hash(uint4* __restrict__ out, u32 st, const u32 (&h)[8]) {
    bf_u32x4 lo4 = {h[0], h[1], h[2], h[3]}, hi4 = {h[4], h[5], h[6], h[7]};
    __builtin_nontemporal_store(lo4, reinterpret_cast<bf_u32x4*>(out + 2 * (size_t)st));
    __builtin_nontemporal_store(hi4, reinterpret_cast<bf_u32x4*>(out + 2 * (size_t)st + 1));
}
// initial chaining value: zeros (stwo convention) or IV ^ parameter block (RFC 7693); rfc = 0 / 0xFFFFFFFF
__device__ __forceinline__ void node_init(u32 (&h)[8], u32 rfc) {
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = B2S_IV[k] & rfc;
    h[0] ^= 0x01010020u & rfc;
}
// One message block of column words: m[w] = column (c0 + w) at row i, zero beyond the last column. The descriptors are wave-uniform and
// come through scalar loads (s_load, one wait for all of them), then the N cells are in flight together (one wait): two memory latencies
// per block. The loop this replaces paid a descriptor -> cell round trip per group of four columns (k_merkle_layer) or per COLUMN (the
// one-workgroup kernels of the small end, where nothing else hides the latency: 75 us for the main-trace subtree against 23 us for a
// column-less one, profiles/r04_2p22_timeline.txt). A partial group re-reads the last column (valid addresses, no branches).
template <int N>
__device__ __forceinline__ void load_col_words(u32 (&m)[16], const ColDesc* __restrict__ cols, u32 c0, u32 ncols, u32 i) {
    const u32 lastc = ncols - 1;
    ColDesc d[N];
#pragma unroll
    for (u32 w = 0; w < N; w++) d[w] = ld_constant(cols + min(c0 + w, lastc));
    u32 v[N];
#pragma unroll
    for (u32 w = 0; w < N; w++) v[w] = ld_col(d[w], i);
#pragma unroll
    for (u32 w = 0; w < 16; w++) m[w] = (w < N && c0 + w < ncols) ? v[w < N ? w : 0] : 0u;
}
// c0 < ncols
__device__ __forceinline__ void load_col_block(u32 (&m)[16], const ColDesc* __restrict__ cols, u32 c0, u32 ncols, u32 i) {
    const u32 left = ncols - c0;
    if (left > 12) load_col_words<16>(m, cols, c0, ncols, i);
    else if (left > 8) load_col_words<12>(m, cols, c0, ncols, i);
    else if (left > 4) load_col_words<8>(m, cols, c0, ncols, i);
    else load_col_words<4>(m, cols, c0, ncols, i);
}
// Hash of node i: the two child hashes (a, b = left, c, d = right) when has_children, then the LE-u32 values of the ncols columns at row i.
__device__ __forceinline__ void merkle_node_hash(u32 (&h)[8], bool has_children, uint4 a, uint4 b, uint4 c, uint4 d, const ColDesc* __restrict__ cols, u32 ncols, u32 i, u32 rfc) {
    node_init(h, rfc);
    const u32 total_bytes = (has_children ? 64u : 0u) + 4u * ncols;
    u32 m[16];
    u32 done = 0;   // bytes compressed so far
    u32 c0 = 0;     // next column to absorb
    if (has_children) {
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
        m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
        done = 64;
        bool last = total_bytes == 64;
        blake2s_compress(h, m, done & rfc, last ? rfc : 0u);
        if (last) return;
    }
    // remaining message: column values, 16 words per block (zero padded)
    // (requesting the next block's words before the current block is compressed — two message buffers, 66 VGPRs — measured slower:
    // 64-column leaves 84.5 -> 82.4 % of the compression peak, r04)
    for (;;) {
        if (c0 < ncols) load_col_block(m, cols, c0, ncols, i);
        else {
#pragma unroll
            for (u32 w = 0; w < 16; w++) m[w] = 0;          // a node with neither children nor columns: one empty block
        }
        u32 take = min(64u, total_bytes - done);
        done += take; c0 += 16;
        bool last = done == total_bytes;
        blake2s_compress(h, m, done & rfc, last ? rfc : 0u);
        if (last) break;
    }
}
__device__ __forceinline__ void merkle_node(u32 st, uint4* __restrict__ out, const uint4* __restrict__ prev, const ColDesc* __restrict__ cols, u32 ncols,
                                            u32 out_shift, u32 prev_shift, u32 rfc) {
    const u32 i = st << out_shift;          // representative node of this stored slot
    u32 h[8];
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a;
    if (prev) {
        const size_t cl = ((size_t)2 * i) >> prev_shift, cr = ((size_t)2 * i + 1) >> prev_shift;   // stored slots of the two children
        a = prev[2 * cl]; b = prev[2 * cl + 1]; c = prev[2 * cr]; d = prev[2 * cr + 1];
    }
    merkle_node_hash(h, prev != nullptr, a, b, c, d, cols, ncols, i, rfc);
    store_hash(out, st, h);
}
// Grid-stride over the stored nodes: large layers give every lane several nodes, which amortises wave launch and the kernel prologue.
// [first, first + n_stored) is the range of stored nodes this launch computes (the whole layer, or one rank's share of it).
// Leaves over at most 4 columns (composition and FRI trees: ~200 M of a proof's 674 M compressions) get their own loop in which the
// inputs of a lane's next node are fetched while the current node is compressed: with one compression per node the load latency is
// otherwise exposed once per node and wave (85 -> 89 % of the compression peak). The same for column-less inner nodes measured worse.
__global__ void __launch_bounds__(256) k_merkle_layer(uint4* __restrict__ out, const uint4* __restrict__ prev, const ColDesc* __restrict__ cols, u32 ncols, u32 n_stored,
                                                      u32 out_shift, u32 prev_shift, u32 first, u32 rfc) {
    const u32 stride = gridDim.x * blockDim.x;
    u32 st = blockIdx.x * blockDim.x + threadIdx.x;
    if (st >= n_stored) return;
    if (!prev && ncols >= 1 && ncols <= 4) {
        const u32 lastc = ncols - 1;
        const ColDesc d0 = cols[0], d1 = cols[min(1u, lastc)], d2 = cols[min(2u, lastc)], d3 = cols[min(3u, lastc)];
        u32 i = (first + st) << out_shift;
        u32 n0 = ld_col(d0, i), n1 = ld_col(d1, i), n2 = ld_col(d2, i), n3 = ld_col(d3, i);
        for (;;) {
            const u32 cur = first + st;
            u32 m[16];
#pragma unroll
            for (int k = 4; k < 16; k++) m[k] = 0;
            m[0] = n0; m[1] = ncols > 1 ? n1 : 0u; m[2] = ncols > 2 ? n2 : 0u; m[3] = ncols > 3 ? n3 : 0u;
            st += stride;
            const bool more = st < n_stored;
            if (more) { i = (first + st) << out_shift; n0 = ld_col(d0, i); n1 = ld_col(d1, i); n2 = ld_col(d2, i); n3 = ld_col(d3, i); }
            u32 h[8];
            node_init(h, rfc);
            blake2s_compress(h, m, (4u * ncols) & rfc, rfc);
            store_hash(out, cur, h);
            if (!more) return;
        }
    }
    for (; st < n_stored; st += stride) merkle_node(first + st, out, prev, cols, ncols, out_shift, prev_shift, rfc);
}

// Blake2sChannel stepped on the device for the FRI commit phase (FriProver::commit: mix_root(layer root) then draw_felt per layer):
// removes the device -> host -> device round trip between consecutive layers. One lane; two compressions plus rare redraws.
// chan = digest[8] || n_sent. alpha_out = alpha[4] || alpha^2[4]. root_out receives a copy of the root (the roots of all layers are
// collected in one array and read back once).
__device__ void channel_step(u32* __restrict__ chan, const u32* __restrict__ root, u32* __restrict__ alpha_out, u32* __restrict__ root_out, u32* alpha_lds = nullptr) {
    u32 h[8], m[16], digest[8];
    // mix_root: digest = Blake2s(digest || root), n_sent = 0
    for (int k = 0; k < 8; k++) { m[k] = chan[k]; m[8 + k] = root[k]; root_out[k] = root[k]; h[k] = B2S_IV[k]; }   // root_out: contiguous copy for one read-back
    h[0] ^= 0x01010020u;
    blake2s_compress(h, m, 64, 0xFFFFFFFFu);
    for (int k = 0; k < 8; k++) digest[k] = h[k];
    // draw_felt: Blake2s(digest || n_sent as LE u32 || zero padding to 64 bytes), redrawn until all 8 words are < 2P
    u32 n_sent = 0;
    for (;;) {
        for (int k = 0; k < 8; k++) { m[k] = digest[k]; m[8 + k] = 0; h[k] = B2S_IV[k]; }
        m[8] = n_sent++;
        h[0] ^= 0x01010020u;
        blake2s_compress(h, m, 64, 0xFFFFFFFFu);
        bool ok = true;
        for (int k = 0; k < 8; k++) ok = ok && h[k] < 2u * P31;
        if (ok) break;
    }
    Q31 alpha = q_make(h[0] >= P31 ? h[0] - P31 : h[0], h[1] >= P31 ? h[1] - P31 : h[1], h[2] >= P31 ? h[2] - P31 : h[2], h[3] >= P31 ? h[3] - P31 : h[3]);
    Q31 sq = q_mul(alpha, alpha);
    alpha_out[0] = alpha.a.a; alpha_out[1] = alpha.a.b; alpha_out[2] = alpha.b.a; alpha_out[3] = alpha.b.b;
    alpha_out[4] = sq.a.a; alpha_out[5] = sq.a.b; alpha_out[6] = sq.b.a; alpha_out[7] = sq.b.b;
    if (alpha_lds) { for (int k = 0; k < 8; k++) alpha_lds[k] = alpha_out[k]; }
    for (int k = 0; k < 8; k++) chan[k] = digest[k];
    chan[8] = n_sent;
}
__global__ void k_channel_mix_root_draw(u32* __restrict__ chan, const u32* __restrict__ root, u32* __restrict__ alpha_out, u32* __restrict__ root_out) {
    if (threadIdx.x || blockIdx.x) return;
    channel_step(chan, root, alpha_out, root_out);
}

// ---- the small end of a tree in two launches ---------------------------------------------------------------------------------------
// Below 2^18 nodes a layer launch is pure latency (~5 us for < 2 us of work) and the layers form a dependent chain. With the tree's layout
// as a kernel argument (MerkleTreeDesc: layer pointers, per-level column lists; 528 bytes of kernarg — no staging copy to wait for) two kernels cover it:
//   k_merkle_subtree: levels [hi .. 9], hi <= 17 — one workgroup (256 lanes) per node of level 9 hashes that node's subtree: its 2^(hi-9) nodes
//                     of level hi (children from level hi + 1 in HBM, or none: leaves), then level by level through LDS;
//   k_merkle_top:     levels [8 .. 0] (or [min(max_log, 9) .. 0] of a tree too small for the subtree kernel) by a single workgroup (children of its first level from HBM), columns included, then
//                     (FRI commit phase) the channel step on the root.
// Every level is also written to HBM: the decommitment reads hashes from there. Un-replicated levels only (node i stored at i).
//
// These kernels are latency chains: one compression per tree level, ~1.7 us each when one lane runs the 977 dependent instructions. Narrow
// levels (and the channel's two hashes) are therefore hashed by a QUAD of lanes per compression: lane i of the quad owns column i of the
// 4 x 4 Blake2s state (a, b, c, d = v[i], v[4+i], v[8+i], v[12+i]), the diagonal step rotates b, c, d across the quad with DPP quad_perm
// (no LDS, no extra instruction when the compiler folds the DPP modifier into the consumer), every lane holds the whole message and picks
// its two words per half-round by lane parity (3 v_cndmask each). ~490 instructions instead of 977 on the dependent chain.
// Each kernel keeps ONE single-lane and ONE quad compression site (a compression is ~8 KiB / ~4 KiB of code; the instruction cache is cold
// at every launch of a short kernel).
__device__ __forceinline__ void hash_to_lds(uint4* __restrict__ s, u32 j, const u32 (&h)[8]) { s[2 * j] = make_uint4(h[0], h[1], h[2], h[3]); s[2 * j + 1] = make_uint4(h[4], h[5], h[6], h[7]); }
__device__ __forceinline__ void hash_to_hbm(uint4* __restrict__ out, u32 i, const u32 (&h)[8]) { out[2 * (size_t)i] = make_uint4(h[0], h[1], h[2], h[3]); out[2 * (size_t)i + 1] = make_uint4(h[4], h[5], h[6], h[7]); }
#define BF_QDPP(x, ctrl) ((u32)__builtin_amdgcn_mov_dpp((int)(x), (ctrl), 0xf, 0xf, true))
#define BF_QG(mx, my) \
    a = a + b + (mx); d = rotr(d ^ a, 16); c = c + d; b = rotr(b ^ c, 12); \
    a = a + b + (my); d = rotr(d ^ a, 8);  c = c + d; b = rotr(b ^ c, 7);
#define BF_QROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) { \
    u32 mx = qsel(m[s0], m[s2], m[s4], m[s6]), my = qsel(m[s1], m[s3], m[s5], m[s7]); \
    BF_QG(mx, my) \
    b = BF_QDPP(b, 0x39); c = BF_QDPP(c, 0x4E); d = BF_QDPP(d, 0x93);      /* lane i takes b of lane i+1, c of i+2, d of i+3: the diagonals */ \
    mx = qsel(m[s8], m[s10], m[s12], m[s14]); my = qsel(m[s9], m[s11], m[s13], m[s15]); \
    BF_QG(mx, my) \
    b = BF_QDPP(b, 0x93); c = BF_QDPP(c, 0x4E); d = BF_QDPP(d, 0x39); }
// ha = h[qi], hb = h[4 + qi] on entry and on return (qi = lane & 3); m, t0, f0 identical in the 4 lanes; all 4 lanes of the quad active.
__device__ __forceinline__ void blake2s_compress_quad(u32& ha, u32& hb, const u32 (&m)[16], u32 t0, u32 f0, u32 qi) {
    const bool odd = qi & 1, up = qi & 2;
    auto qsel = [&](u32 x0, u32 x1, u32 x2, u32 x3) -> u32 { const u32 lo = odd ? x1 : x0, hi = odd ? x3 : x2; return up ? hi : lo; };
    u32 a = ha, b = hb;
    u32 c = qsel(0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
    u32 d = qsel(0x510E527Fu ^ t0, 0x9B05688Cu, 0x1F83D9ABu ^ f0, 0x5BE0CD19u);
    BF_QROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    BF_QROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    BF_QROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    BF_QROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    BF_QROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    BF_QROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    BF_QROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    BF_QROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    BF_QROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    BF_QROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    ha ^= a ^ c; hb ^= b ^ d;
}
// initial chaining words of lane qi: (IV ^ parameter block)[qi], [4 + qi]
__device__ __forceinline__ void quad_iv(u32 qi, u32& lo, u32& hi) {
    const bool odd = qi & 1, up = qi & 2;
    const u32 l0 = odd ? 0xBB67AE85u : (0x6A09E667u ^ 0x01010020u), l1 = odd ? 0xA54FF53Au : 0x3C6EF372u;
    const u32 h0 = odd ? 0x9B05688Cu : 0x510E527Fu, h1 = odd ? 0x5BE0CD19u : 0x1F83D9ABu;
    lo = up ? l1 : l0; hi = up ? h1 : h0;
}
__device__ __forceinline__ void kids_to_m(u32 (&m)[16], uint4 a, uint4 b, uint4 c, uint4 d) {
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
}
// Node hash by one lane with ONE compression site: message blocks = [the two child hashes]? then the column words 16 at a time.
// leaf4 != nullptr: the node's four column values are given (FRI layer leaves from LDS) instead of column descriptors.
__device__ __forceinline__ void node_hash_lean(u32 (&h)[8], bool has, uint4 ka, uint4 kb, uint4 kc, uint4 kd, const ColDesc* __restrict__ cols, u32 ncols, u32 i, u32 rfc,
                                               const u32* leaf4 = nullptr) {
    node_init(h, rfc);
    const u32 nblk = (has ? 1u : 0u) + (ncols + 15) / 16 + ((!has && ncols == 0) ? 1u : 0u);
    u32 done = 0;
#pragma unroll 1
    for (u32 blk = 0; blk < nblk; blk++) {
        u32 m[16];
        if (blk == 0 && has) { kids_to_m(m, ka, kb, kc, kd); done = 64; }
        else if (leaf4) {
#pragma unroll
            for (int w = 0; w < 16; w++) m[w] = w < 4 ? leaf4[w] : 0u;
            done += 16;
        } else {
            const u32 c0 = 16 * (blk - (has ? 1u : 0u));
            if (c0 < ncols) load_col_block(m, cols, c0, ncols, i);
            else {
#pragma unroll
                for (u32 w = 0; w < 16; w++) m[w] = 0;      // the empty message of a column-less leaf
            }
            done += min(64u, 4u * (ncols - min(ncols, c0)));
        }
        const bool last = blk + 1 == nblk;
        blake2s_compress(h, m, done & rfc, last ? rfc : 0u);
    }
}

// The same node hash by a QUAD of lanes (ha = h[qi], hb = h[4 + qi] on return), any number of columns: one quad compression site in a loop.
// Every lane of the quad builds the whole message block (the 4 lanes read the same addresses).
__device__ __forceinline__ void node_hash_quad(u32& ha, u32& hb, bool has, uint4 ka, uint4 kb, uint4 kc, uint4 kd, const ColDesc* __restrict__ cols, u32 ncols, u32 i, u32 rfc, u32 qi) {
    quad_iv(qi, ha, hb); ha &= rfc; hb &= rfc;
    const u32 nblk = (has ? 1u : 0u) + (ncols + 15) / 16 + ((!has && ncols == 0) ? 1u : 0u);
    u32 done = 0;
#pragma unroll 1
    for (u32 blk = 0; blk < nblk; blk++) {
        u32 m[16];
        if (blk == 0 && has) { kids_to_m(m, ka, kb, kc, kd); done = 64; }
        else {
            const u32 c0 = 16 * (blk - (has ? 1u : 0u));
            if (c0 < ncols) load_col_block(m, cols, c0, ncols, i);
            else {
#pragma unroll
                for (u32 w = 0; w < 16; w++) m[w] = 0;      // the empty message of a column-less leaf
            }
            done += min(64u, 4u * (ncols - min(ncols, c0)));
        }
        const bool last = blk + 1 == nblk;
        blake2s_compress_quad(ha, hb, m, done & rfc, last ? rfc : 0u, qi);
    }
}

// General form: workgroup b (= blockIdx.x + wg0) owns the nodes [b << (lg - lo), (b + 1) << (lg - lo)) of every level lg in [stop, hi] — the
// subtree(s) under what would be node b of level lo. The small end of a tree: lo = stop = 9, wg0 = 0, 512 workgroups. The bottom of a shard
// group's share-wise band (r04): stop = the band's lowest level, lo = stop - r (a workgroup takes 2^r roots: 256 nodes of level hi), wg0 = the
// rank's first workgroup — the band's last four levels, 64..512 workgroups each as single launches, in one launch. At most 512 nodes per level
// and workgroup (LDS).
__global__ void __launch_bounds__(256) k_merkle_subtree(const MerkleTreeDesc td, u32 hi, u32 rfc, u32 lo, u32 stop, u32 wg0) {
    __shared__ uint4 s_lv[2][2 * 256];
    const u32 b = blockIdx.x + wg0, t = threadIdx.x, qi = t & 3, qn = t >> 2;
    for (u32 lg = hi; lg >= stop; lg--) {
        const u32 n = 1u << (lg - lo);
        const ColDesc* cols = td.cols + td.col_off[lg];
        const u32 ncols = td.col_off[lg - 1] - td.col_off[lg];      // col_off is indexed by level; levels are laid out descending
        const bool first = lg == hi, has = !first || hi < td.max_log;
        const uint4* prev = first && has ? td.layers[hi + 1] : nullptr;
        const u32 ps = first && has ? td.shifts[hi + 1] : 0u;
        const uint4* src = s_lv[(lg + 1) & 1];
        const bool quad = 4 * n <= blockDim.x;
        const u32 j = quad ? qn : t;
        if (j < n) {
            const u32 i = (b << (lg - lo)) + j;
            uint4 ka = make_uint4(0, 0, 0, 0), kb = ka, kc = ka, kd = ka;
            if (has) {
                if (first) { const size_t cl = ((size_t)2 * i) >> ps, cr = ((size_t)2 * i + 1) >> ps; ka = prev[2 * cl]; kb = prev[2 * cl + 1]; kc = prev[2 * cr]; kd = prev[2 * cr + 1]; }
                else { ka = src[4 * j]; kb = src[4 * j + 1]; kc = src[4 * j + 2]; kd = src[4 * j + 3]; }
            }
            if (quad) {
                u32 ha, hb;
                node_hash_quad(ha, hb, has, ka, kb, kc, kd, cols, ncols, i, rfc, qi);
                u32* o = reinterpret_cast<u32*>(td.layers[lg]) + 8 * (size_t)i; o[qi] = ha; o[4 + qi] = hb;
                u32* l = reinterpret_cast<u32*>(s_lv[lg & 1]) + 8 * j; l[qi] = ha; l[4 + qi] = hb;
            } else {
                u32 h[8];
                node_hash_lean(h, has, ka, kb, kc, kd, cols, ncols, i, rfc);
                hash_to_hbm(td.layers[lg], i, h);
                hash_to_lds(s_lv[lg & 1], j, h);
            }
        }
        __syncthreads();
    }
}

// chan != nullptr: Blake2sChannel::mix_root(root) and draw_felt() follow the root as two more quad steps (levels -1 and -2 of the loop).
// stamp_out != nullptr (with root_out in pinned host memory): behind the root the kernel writes stamp_value there — the host polls that word
// instead of an event behind the kernel (ctx.h: wait_stamp).
__global__ void __launch_bounds__(256) k_merkle_top(const MerkleTreeDesc td, u32 top_hi, u32* chan, u32* alpha_out, u32* root_out, u32 rfc, u32* stamp_out, u32 stamp_value) {
    // The levels form a dependent chain (one compression of latency each): a level's nodes stay in LDS for the next level (two buffers,
    // alternating) besides going to HBM for the decommitment, so only the first level pays a global-memory round trip.
    __shared__ uint4 s_lv[2][2 * 512];
    __shared__ u32 s_ch[16];              // [0, 8): channel digest; [8, 16): the draw's output
    const u32 t = threadIdx.x, qi = t & 3, qn = t >> 2;
    if (chan && t < 8) s_ch[t] = chan[t];
    u32 n_sent = 0;
    for (int lg = (int)top_hi;;) {
        const bool tree = lg >= 0;
        const ColDesc* cols = tree ? td.cols + td.col_off[lg] : nullptr;
        const u32 ncols = tree ? (lg > 0 ? td.col_off[lg - 1] : td.n_cols) - td.col_off[lg] : 0u;
        const bool first = lg == (int)top_hi, has = !first || top_hi < td.max_log;
        const uint4* prev = tree && first && has ? td.layers[lg + 1] : nullptr;
        const uint4* src = s_lv[(lg + 1) & 1];
        const u32 n = tree ? 1u << lg : 1u;
        const bool quad = !tree || n <= 64;
        if (!quad) {
            for (u32 i = t; i < n; i += blockDim.x) {
                u32 h[8];
                uint4 ka = make_uint4(0, 0, 0, 0), kb = ka, kc = ka, kd = ka;
                if (has) {
                    if (first) { ka = prev[4 * i]; kb = prev[4 * i + 1]; kc = prev[4 * i + 2]; kd = prev[4 * i + 3]; }
                    else { ka = src[4 * i]; kb = src[4 * i + 1]; kc = src[4 * i + 2]; kd = src[4 * i + 3]; }
                }
                node_hash_lean(h, has, ka, kb, kc, kd, cols, ncols, i, rfc);
                hash_to_hbm(td.layers[lg], i, h);
                hash_to_lds(s_lv[lg & 1], i, h);
            }
        } else if (tree) {
            if (qn < n) {
                u32 ha, hb;
                uint4 ka = make_uint4(0, 0, 0, 0), kb = ka, kc = ka, kd = ka;
                if (has) {
                    if (first) { ka = prev[4 * qn]; kb = prev[4 * qn + 1]; kc = prev[4 * qn + 2]; kd = prev[4 * qn + 3]; }
                    else { ka = src[4 * qn]; kb = src[4 * qn + 1]; kc = src[4 * qn + 2]; kd = src[4 * qn + 3]; }
                }
                node_hash_quad(ha, hb, has, ka, kb, kc, kd, cols, ncols, qn, rfc, qi);
                u32* o = reinterpret_cast<u32*>(td.layers[lg]) + 8 * qn; o[qi] = ha; o[4 + qi] = hb;
                u32* l = reinterpret_cast<u32*>(s_lv[lg & 1]) + 8 * qn; l[qi] = ha; l[4 + qi] = hb;
                if (lg == 0 && !chan && root_out) { root_out[qi] = ha; root_out[4 + qi] = hb; if (stamp_out) __threadfence_system(); }     // the root, straight to where the host reads it
            }
        } else if (qn < n) {
            u32 m[16], ha, hb, t0 = 64u, f0 = 0xFFFFFFFFu;
            quad_iv(qi, ha, hb);
            {
#pragma unroll
                for (int k = 0; k < 8; k++) m[k] = s_ch[k];
                if (lg == -1) {         // mix_root: Blake2s(digest || root)
                    const u32* root = reinterpret_cast<const u32*>(s_lv[0]);
#pragma unroll
                    for (int k = 0; k < 8; k++) m[8 + k] = root[k];
                    root_out[qi] = m[8 + qi]; root_out[4 + qi] = m[12 + qi];     // contiguous copy of the roots for one read-back
                } else {                // draw: Blake2s(digest || n_sent as LE u32 || zero padding to 64 bytes)
#pragma unroll
                    for (int k = 9; k < 16; k++) m[k] = 0;
                    m[8] = n_sent;
                }
            }
            blake2s_compress_quad(ha, hb, m, t0, f0, qi);
            if (lg == -1) { s_ch[qi] = ha; s_ch[4 + qi] = hb; }
            else { s_ch[8 + qi] = ha; s_ch[12 + qi] = hb; }
        }
        __syncthreads();
        if (lg == -2) {
            bool ok = true;      // redrawn until all 8 words are < 2P
#pragma unroll
            for (int k = 0; k < 8; k++) ok = ok && s_ch[8 + k] < 2u * P31;
            n_sent++;
            if (ok) break;
            __syncthreads();     // every lane has read the rejected draw before it is overwritten
        } else {
            lg--;
            if (lg < 0 && !chan) {
                if (stamp_out && t == 0) __hip_atomic_store(stamp_out, stamp_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);     // behind the barrier: the four root stores are out
                return;
            }
        }
    }
    if (t == 0) {
        u32 w[4];
        for (int k = 0; k < 4; k++) w[k] = s_ch[8 + k] >= P31 ? s_ch[8 + k] - P31 : s_ch[8 + k];
        const Q31 alpha = q_make(w[0], w[1], w[2], w[3]), sq = q_mul(alpha, alpha);
        alpha_out[0] = alpha.a.a; alpha_out[1] = alpha.a.b; alpha_out[2] = alpha.b.a; alpha_out[3] = alpha.b.b;
        alpha_out[4] = sq.a.a; alpha_out[5] = sq.a.b; alpha_out[6] = sq.b.a; alpha_out[7] = sq.b.b;
        for (int k = 0; k < 8; k++) chan[k] = s_ch[k];
        chan[8] = n_sent;
    }
}

// ---- FRI commit phase below 2^10 rows: one launch ------------------------------------------------------------------------------------
// Steps of one layer of 2^log rows: leaves (kind 0, level log), inner levels log-1 .. 0 (kind 1), mix_root (kind 2), draw (kind 3), fold.
__global__ void __launch_bounds__(256) k_fri_tail(const FriTailArgs* __restrict__ ap) {
    __shared__ u32 s_ev[2][4][1024];
    __shared__ uint4 s_h[2][2 * 1024];
    __shared__ u32 s_ch[16];
    const FriTailArgs& a = *ap;
    const u32 t = threadIdx.x, rfc = a.rfc, qi = t & 3, qn = t >> 2;
    for (u32 i = t; i < (1u << a.top_log); i += 256)
        for (int w = 0; w < 4; w++) s_ev[0][w][i] = a.layer[0].ev[w][i];
    if (t < 8) s_ch[t] = a.chan[t];
    __syncthreads();
    for (u32 k = 0; k < a.n_layers; k++) {
        const u32 log = a.top_log - k;
        const FriTailLayer& L = a.layer[k];
        u32 (*ev)[1024] = s_ev[k & 1];
        u32 n_sent = 0;
        for (int lg = (int)log, kind = 0;;) {
            const u32 n = kind <= 1 ? 1u << lg : 1u;
            const uint4* src = s_h[(lg + 1) & 1];
            if (n > 64) {
                for (u32 i = t; i < n; i += 256) {
                    u32 h[8];
                    u32 leaf[4] = {0, 0, 0, 0};
                    uint4 ka = make_uint4(0, 0, 0, 0), kb = ka, kc = ka, kd = ka;
                    if (kind == 0) { leaf[0] = ev[0][i]; leaf[1] = ev[1][i]; leaf[2] = ev[2][i]; leaf[3] = ev[3][i]; }
                    else { ka = src[4 * i]; kb = src[4 * i + 1]; kc = src[4 * i + 2]; kd = src[4 * i + 3]; }
                    node_hash_lean(h, kind == 1, ka, kb, kc, kd, nullptr, kind == 0 ? 4u : 0u, i, rfc, leaf);
                    hash_to_hbm(L.tree[lg], i, h);
                    hash_to_lds(s_h[lg & 1], i, h);
                }
            } else if (qn < n) {
                u32 m[16], ha, hb, t0 = 64u, f0 = 0xFFFFFFFFu;
                quad_iv(qi, ha, hb);
                if (kind == 0) {
#pragma unroll
                    for (int w = 4; w < 16; w++) m[w] = 0;
                    m[0] = ev[0][qn]; m[1] = ev[1][qn]; m[2] = ev[2][qn]; m[3] = ev[3][qn];
                    ha &= rfc; hb &= rfc; t0 = 16u & rfc; f0 = rfc;
                } else if (kind == 1) {
                    kids_to_m(m, src[4 * qn], src[4 * qn + 1], src[4 * qn + 2], src[4 * qn + 3]);
                    ha &= rfc; hb &= rfc; t0 &= rfc; f0 = rfc;
                } else {
#pragma unroll
                    for (int w = 0; w < 8; w++) m[w] = s_ch[w];
                    if (kind == 2) {
                        const u32* root = reinterpret_cast<const u32*>(s_h[0]);
#pragma unroll
                        for (int w = 0; w < 8; w++) m[8 + w] = root[w];
                        u32* ro = a.roots + 8 * (a.root_idx + k);
                        ro[qi] = m[8 + qi]; ro[4 + qi] = m[12 + qi];
                    } else {
#pragma unroll
                        for (int w = 9; w < 16; w++) m[w] = 0;
                        m[8] = n_sent;
                    }
                }
                blake2s_compress_quad(ha, hb, m, t0, f0, qi);
                if (kind <= 1) {
                    u32* o = reinterpret_cast<u32*>(L.tree[lg]) + 8 * qn; o[qi] = ha; o[4 + qi] = hb;
                    u32* l = reinterpret_cast<u32*>(s_h[lg & 1]) + 8 * qn; l[qi] = ha; l[4 + qi] = hb;
                } else if (kind == 2) { s_ch[qi] = ha; s_ch[4 + qi] = hb; }
                else { s_ch[8 + qi] = ha; s_ch[12 + qi] = hb; }
            }
            __syncthreads();
            if (kind == 3) {
                bool ok = true;
#pragma unroll
                for (int w = 0; w < 8; w++) ok = ok && s_ch[8 + w] < 2u * P31;
                n_sent++;
                if (ok) break;
                __syncthreads();
            } else if (kind == 2) kind = 3;
            else if (lg == 0) kind = 2;
            else { lg--; kind = 1; }
        }
        // alpha, alpha^2 (every lane computes them; lane 0 publishes them and the channel state)
        u32 w4[4];
#pragma unroll
        for (int w = 0; w < 4; w++) w4[w] = s_ch[8 + w] >= P31 ? s_ch[8 + w] - P31 : s_ch[8 + w];
        const Q31 alpha = q_make(w4[0], w4[1], w4[2], w4[3]), alpha_sq = q_mul(alpha, alpha);
        const QConst k_alpha = q_const(alpha), k_alpha_sq = q_const(alpha_sq);       // the folds multiply by these two constants (m31.h: q_mul_const)
        if (t == 0) {
            u32* ao = a.alpha + 8 * (a.alpha_idx + k);
            ao[0] = alpha.a.a; ao[1] = alpha.a.b; ao[2] = alpha.b.a; ao[3] = alpha.b.b;
            ao[4] = alpha_sq.a.a; ao[5] = alpha_sq.a.b; ao[6] = alpha_sq.b.a; ao[7] = alpha_sq.b.b;
            if (k + 1 == a.n_layers) { for (int w = 0; w < 8; w++) a.chan[w] = s_ch[w]; a.chan[8] = n_sent; }
        }
        // fold into the next layer
        u32 (*nx)[1024] = s_ev[(k + 1) & 1];
        u32* const* out = k + 1 < a.n_layers ? a.layer[k + 1].ev : a.ev_last;
        for (u32 i = t; i < (1u << (log - 1)); i += 256) {
            const u32 xinv = a.itw[a.tw_total - (1u << log) + i];
            const Q31 fx = q_make(ev[0][2 * i], ev[1][2 * i], ev[2][2 * i], ev[3][2 * i]), fn = q_make(ev[0][2 * i + 1], ev[1][2 * i + 1], ev[2][2 * i + 1], ev[3][2 * i + 1]);
            Q31 r = q_add(q_add(fx, fn), q_mul_const(q_mulm(q_sub(fx, fn), xinv), k_alpha));
            if (L.quot[0]) {
                const u32* t1 = a.itw + (a.tw_total - (1u << (log - 1)));
                const u32 cx = t1[(i >> 2) * 2], cy = t1[(i >> 2) * 2 + 1], sel = i & 3;
                const u32 yinv = sel == 0 ? cy : sel == 1 ? m_neg(cy) : sel == 2 ? m_neg(cx) : cx;
                const Q31 fp = q_make(L.quot[0][2 * i], L.quot[1][2 * i], L.quot[2][2 * i], L.quot[3][2 * i]);
                const Q31 fq = q_make(L.quot[0][2 * i + 1], L.quot[1][2 * i + 1], L.quot[2][2 * i + 1], L.quot[3][2 * i + 1]);
                const Q31 fprime = q_add(q_mul_const(q_mulm(q_sub(fp, fq), yinv), k_alpha), q_add(fp, fq));
                r = q_add(q_mul_const(r, k_alpha_sq), fprime);
            }
            nx[0][i] = r.a.a; nx[1][i] = r.a.b; nx[2][i] = r.b.a; nx[3][i] = r.b.b;
            out[0][i] = r.a.a; out[1][i] = r.a.b; out[2][i] = r.b.a; out[3][i] = r.b.b;
        }
        __syncthreads();
    }
}
// ---- one FRI inner layer of 2^11 .. 2^16 rows as ONE launch ----------------------------------------------------------------------------
// fold (fold_line of the previous layer + fold-in of the quotient of that size), leaves, the tree and the channel step: a workgroup folds
// and hashes 256 rows and reduces them to one node in LDS; the workgroup that finishes LAST (a ticket counter in HBM, release / acquire fences
// at device scope) hashes the remaining levels above the workgroup roots and steps the channel. Three launches (fold, subtree, top) and two
// kernel start-ups less on the latency chain of every such layer: 27-49 us per layer instead of ~61. The hand-over costs a device-scope release per
// workgroup, which on this part writes the XCD's L2 back (the eight L2s are not coherent with each other): the kernel's time grows with the number
// of workgroups (8: 27 us, 256: 49 us, 512: 73 us — slower than three launches), hence the 2^16-row limit; the same structure for the small end of
// the big trees (512 workgroups) measured 70 us against 56-69 for k_merkle_subtree + k_merkle_top and was not kept (profiles/r03_small_end_fusion.txt).
__global__ void __launch_bounds__(256) k_fri_layer(FriLayerArgs a) {
    __shared__ uint4 s_h[2][2 * 512];
    __shared__ u32 s_ch[16];
    __shared__ u32 s_ticket;
    const u32 t = threadIdx.x, qi = t & 3, qn = t >> 2, rfc = a.rfc, log = a.log;
    const u32 nb = gridDim.x;                               // 2^(log - 8) workgroups
    const u32 root_lv = log - 8;                            // level of the workgroup roots
    // ---- fold: row i of this layer from rows 2i, 2i + 1 of the previous one (2^(log + 1) rows) ----
    u32 leaf[4];
    {
        const u32 i = blockIdx.x * 256 + t, slog = log + 1;
        const QConst k_alpha = q_const(q_make(a.alpha8[0], a.alpha8[1], a.alpha8[2], a.alpha8[3]));
        const u32 xinv = a.itw[a.tw_total - (1u << slog) + i];
        const uint2 a0 = reinterpret_cast<const uint2*>(a.src[0])[i], a1 = reinterpret_cast<const uint2*>(a.src[1])[i], a2 = reinterpret_cast<const uint2*>(a.src[2])[i], a3 = reinterpret_cast<const uint2*>(a.src[3])[i];
        const Q31 fx = q_make(a0.x, a1.x, a2.x, a3.x), fn = q_make(a0.y, a1.y, a2.y, a3.y);
        Q31 r = q_add(q_add(fx, fn), q_mul_const(q_mulm(q_sub(fx, fn), xinv), k_alpha));
        if (a.quot[0]) {
            const QConst k_alpha_sq = q_const(q_make(a.alpha8[4], a.alpha8[5], a.alpha8[6], a.alpha8[7]));
            const u32* t1 = a.itw + (a.tw_total - (1u << (slog - 1)));
            const u32 cx = t1[(i >> 2) * 2], cy = t1[(i >> 2) * 2 + 1], sel = i & 3;
            const u32 yinv = sel == 0 ? cy : sel == 1 ? m_neg(cy) : sel == 2 ? m_neg(cx) : cx;
            const uint2 b0 = reinterpret_cast<const uint2*>(a.quot[0])[i], b1 = reinterpret_cast<const uint2*>(a.quot[1])[i], b2 = reinterpret_cast<const uint2*>(a.quot[2])[i], b3 = reinterpret_cast<const uint2*>(a.quot[3])[i];
            const Q31 fp = q_make(b0.x, b1.x, b2.x, b3.x), fq = q_make(b0.y, b1.y, b2.y, b3.y);
            const Q31 fprime = q_add(q_mul_const(q_mulm(q_sub(fp, fq), yinv), k_alpha), q_add(fp, fq));
            r = q_add(q_mul_const(r, k_alpha_sq), fprime);
        }
        leaf[0] = r.a.a; leaf[1] = r.a.b; leaf[2] = r.b.a; leaf[3] = r.b.b;
        a.dst[0][i] = leaf[0]; a.dst[1][i] = leaf[1]; a.dst[2][i] = leaf[2]; a.dst[3][i] = leaf[3];
    }
    // ---- levels: kind 0 = leaves (level log, 256 per workgroup), 1 = inner; `wide` = nodes of the level this workgroup hashes ----
    bool last_block = false;
    u32 n_sent = 0;
    for (int lg = (int)log, kind = 0;;) {
        const bool in_block = !last_block;
        const u32 n = kind >= 2 ? 1u : in_block ? 1u << (lg - (int)root_lv) : 1u << lg;        // nodes of this step
        const u32 node0 = in_block && kind <= 1 ? blockIdx.x << (lg - (int)root_lv) : 0u;     // first node (global index) of this workgroup at level lg
        const uint4* src = s_h[(lg + 1) & 1];
        const bool from_hbm = last_block && kind == 1 && lg + 1 == (int)root_lv;                // the last workgroup's first level reads every workgroup's root
        if (kind <= 1 && n > 64) {
            for (u32 j = t; j < n; j += 256) {
                u32 h[8];
                uint4 ka = make_uint4(0, 0, 0, 0), kb = ka, kc = ka, kd = ka;
                if (kind == 1) {
                    if (from_hbm) { const uint4* p = a.tree[lg + 1]; ka = p[4 * j]; kb = p[4 * j + 1]; kc = p[4 * j + 2]; kd = p[4 * j + 3]; }
                    else { ka = src[4 * j]; kb = src[4 * j + 1]; kc = src[4 * j + 2]; kd = src[4 * j + 3]; }
                }
                node_hash_lean(h, kind == 1, ka, kb, kc, kd, nullptr, kind == 0 ? 4u : 0u, node0 + j, rfc, leaf);
                hash_to_hbm(a.tree[lg], node0 + j, h);
                hash_to_lds(s_h[lg & 1], j, h);
            }
        } else if (qn < n) {
            u32 m[16], ha, hb, t0 = 64u, f0 = 0xFFFFFFFFu;
            quad_iv(qi, ha, hb);
            if (kind == 1) {
                if (from_hbm) { const uint4* p = a.tree[lg + 1]; kids_to_m(m, p[4 * qn], p[4 * qn + 1], p[4 * qn + 2], p[4 * qn + 3]); }
                else kids_to_m(m, src[4 * qn], src[4 * qn + 1], src[4 * qn + 2], src[4 * qn + 3]);
                ha &= rfc; hb &= rfc; t0 &= rfc; f0 = rfc;
            } else {
#pragma unroll
                for (int w = 0; w < 8; w++) m[w] = s_ch[w];
                if (kind == 2) {
                    const u32* root = reinterpret_cast<const u32*>(s_h[0]);
#pragma unroll
                    for (int w = 0; w < 8; w++) m[8 + w] = root[w];
                    a.root_out[qi] = m[8 + qi]; a.root_out[4 + qi] = m[12 + qi];
                } else {
#pragma unroll
                    for (int w = 9; w < 16; w++) m[w] = 0;
                    m[8] = n_sent;
                }
            }
            blake2s_compress_quad(ha, hb, m, t0, f0, qi);
            if (kind == 1) {
                u32* o = reinterpret_cast<u32*>(a.tree[lg]) + 8 * (size_t)(node0 + qn); o[qi] = ha; o[4 + qi] = hb;
                u32* l = reinterpret_cast<u32*>(s_h[lg & 1]) + 8 * qn; l[qi] = ha; l[4 + qi] = hb;
            } else if (kind == 2) { s_ch[qi] = ha; s_ch[4 + qi] = hb; }
            else { s_ch[8 + qi] = ha; s_ch[12 + qi] = hb; }
        }
        __syncthreads();
        if (kind == 3) {
            bool ok = true;
#pragma unroll
            for (int w = 0; w < 8; w++) ok = ok && s_ch[8 + w] < 2u * P31;
            n_sent++;
            if (ok) break;
            __syncthreads();
        } else if (kind == 2) kind = 3;
        else if (in_block && lg == (int)root_lv) {
            // this workgroup's root is in HBM: release it, take a ticket; every workgroup but the last one is done
            __threadfence();
            if (t == 0) s_ticket = atomicAdd(a.counter, 1u);
            __syncthreads();
            if (s_ticket != nb - 1) return;
            __threadfence();                       // acquire: the other workgroups' roots
            last_block = true;
            if (t == 0) *a.counter = 0;            // ready for the next use of this counter
            if (t < 8) s_ch[t] = a.chan[t];
            if (root_lv == 0) kind = 2; else { lg--; kind = 1; }
            __syncthreads();
        } else if (lg == 0) kind = 2;
        else { lg--; kind = 1; }
    }
    if (t == 0) {
        u32 w4[4];
        for (int w = 0; w < 4; w++) w4[w] = s_ch[8 + w] >= P31 ? s_ch[8 + w] - P31 : s_ch[8 + w];
        const Q31 alpha = q_make(w4[0], w4[1], w4[2], w4[3]), sq = q_mul(alpha, alpha);
        a.alpha_out[0] = alpha.a.a; a.alpha_out[1] = alpha.a.b; a.alpha_out[2] = alpha.b.a; a.alpha_out[3] = alpha.b.b;
        a.alpha_out[4] = sq.a.a; a.alpha_out[5] = sq.a.b; a.alpha_out[6] = sq.b.a; a.alpha_out[7] = sq.b.b;
        for (int w = 0; w < 8; w++) a.chan[w] = s_ch[w];
        a.chan[8] = n_sent;
    }
}
void fri_layer(hipStream_t stream, const FriLayerArgs& a) {
    if (a.log < 11 || a.log > 16) throw std::runtime_error("fri_layer: 2^11 .. 2^16 rows");
    const double nodes = (double)((2u << a.log) - 1);
    ProfScope ps(stream, "k_fri_layer", 48.0 * nodes + 48.0 * (double)(1u << a.log), nodes);
    hipLaunchKernelGGL(k_fri_layer, dim3(1u << (a.log - 8)), dim3(256), 0, stream, a);
}

void fri_tail(hipStream_t stream, const FriTailArgs* d_args, double bytes, double compressions) {
    ProfScope ps(stream, "k_fri_tail", bytes, compressions);
    hipLaunchKernelGGL(k_fri_tail, dim3(1), dim3(256), 0, stream, d_args);
}

#ifndef MERKLE_NODES_PER_LANE
#define MERKLE_NODES_PER_LANE 4
#endif
// col_bytes = bytes of column storage this layer reads (for the roofline accounting only)
void merkle_layer(hipStream_t stream, void* out, const void* prev, const ColDesc* d_cols, u32 ncols, u32 log, double col_bytes, u32 out_shift, u32 prev_shift,
                  u32 node_conv, u32 first, u32 count) {
    const u32 total = (1u << log) >> out_shift;
    const u32 n = count ? count : total;                 // count == 0: the whole layer
    u32 threads = n < 256 ? (n < 64 ? 64 : n) : 256;
    const double frac = (double)n / (double)total;
    // compressions per node: one for the two children, one per started group of 16 column words (both conventions)
    const double comp_per_node = (prev ? 1.0 : 0.0) + (double)((ncols + 15) / 16) + ((!prev && ncols == 0) ? 1.0 : 0.0);
    ProfScope ps(stream, "k_merkle_layer", ((prev ? 64.0 * total : 0.0) + 32.0 * total + col_bytes) * frac, comp_per_node * n, /*dominant=*/true);
    u32 blocks = (n + threads - 1) / threads;
    if (blocks >= (1u << 14)) blocks /= MERKLE_NODES_PER_LANE;   // >= 2^22 nodes: several nodes per lane (measured: 2..16 equivalent, 4 kept)
    hipLaunchKernelGGL(k_merkle_layer, dim3(blocks), dim3(threads), 0, stream, (uint4*)out, (const uint4*)prev, d_cols, ncols, n, out_shift, prev_shift, count ? first : 0u, node_conv ? 0xFFFFFFFFu : 0u);
}
void merkle_subtree(hipStream_t stream, const MerkleTreeDesc& tree, u32 hi, u32 node_conv, double bytes, double compressions) {
    ProfScope ps(stream, "k_merkle_subtree", bytes, compressions);
    hipLaunchKernelGGL(k_merkle_subtree, dim3(1u << MERKLE_SUBTREE_ROOT_LEVEL), dim3(256), 0, stream, tree, hi, node_conv ? 0xFFFFFFFFu : 0u,
                       (u32)MERKLE_SUBTREE_ROOT_LEVEL, (u32)MERKLE_SUBTREE_ROOT_LEVEL, 0u);
}
// levels [stop, hi] of a rank's share: workgroups [wg0, wg0 + n_wg), each over 2^(hi - lo) nodes of level hi (<= 512)
void merkle_subtree_share(hipStream_t stream, const MerkleTreeDesc& tree, u32 hi, u32 stop, u32 lo, u32 wg0, u32 n_wg, u32 node_conv, double bytes, double compressions) {
    if (hi < stop || stop < lo || hi - lo > 9 || !n_wg) throw std::runtime_error("merkle_subtree_share: bad level range");
    ProfScope ps(stream, "k_merkle_subtree", bytes, compressions);
    hipLaunchKernelGGL(k_merkle_subtree, dim3(n_wg), dim3(256), 0, stream, tree, hi, node_conv ? 0xFFFFFFFFu : 0u, lo, stop, wg0);
}
void merkle_top(hipStream_t stream, const MerkleTreeDesc& tree, u32 top_hi, u32 node_conv, u32* d_chan, u32* d_alpha8, u32* d_root_copy, double bytes, double compressions,
                u32* d_stamp, u32 stamp_value) {
    ProfScope ps(stream, "k_merkle_top", bytes, compressions);
    hipLaunchKernelGGL(k_merkle_top, dim3(1), dim3(256), 0, stream, tree, top_hi, d_chan, d_alpha8, d_root_copy, node_conv ? 0xFFFFFFFFu : 0u, d_stamp, stamp_value);
}

void channel_mix_root_draw(hipStream_t stream, u32* d_chan, const u32* d_root, u32* d_alpha8, u32* d_root_copy) {
    hipLaunchKernelGGL(k_channel_mix_root_draw, dim3(1), dim3(64), 0, stream, d_chan, d_root, d_alpha8, d_root_copy);
}

// Proof-of-work search (GrindOps::grind): smallest nonce whose mix_u64 digest has >= pow_bits trailing zero bits
// (trailing_zeros of the first 16 digest bytes as LE u128). Each launch scans `span` nonces from `base`; the minimum hit is kept.
// hashed = Conventions::mix_u64 == 1: digest' = Blake2s-256(digest || LE64(nonce) zero padded to 32 bytes) instead of the raw compression.
__global__ void k_grind(const u32* __restrict__ digest, u64 base, u32 pow_bits, unsigned long long* __restrict__ best, u32 hashed) {
    u64 nonce = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 h[8], m[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = 0;
    if (hashed) {
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = digest[k];
        m[8] = (u32)nonce; m[9] = (u32)(nonce >> 32);
        node_init(h, 0xFFFFFFFFu);
        blake2s_compress(h, m, 64, 0xFFFFFFFFu);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = digest[k];
        m[0] = (u32)nonce; m[1] = (u32)(nonce >> 32);
        blake2s_compress(h, m, 0, 0);
    }
    u32 tz = h[0] ? __ffs(h[0]) - 1 : h[1] ? 32 + __ffs(h[1]) - 1 : h[2] ? 64 + __ffs(h[2]) - 1 : h[3] ? 96 + __ffs(h[3]) - 1 : 128;
    if (tz >= pow_bits) atomicMin(best, (unsigned long long)nonce);
}
void grind_span(hipStream_t stream, const u32* d_digest, u64 base, u32 span, u32 pow_bits, unsigned long long* d_best, u32 mix_u64_conv) {
    hipLaunchKernelGGL(k_grind, dim3(span / 256), dim3(256), 0, stream, d_digest, base, pow_bits, d_best, mix_u64_conv);
}

// Diagnostic only (bfhip_clock_probe; never part of a proof): the shader clock this device SUSTAINS under the proof's dominant load. A
// register-only loop of the same compression the Merkle kernels run, one chain per lane like k_merkle_layer; lane 0 of every workgroup stamps
// the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its loop: clock = d(memtime) / d(memrealtime) x
// 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6). The stamps go to a buffer nothing else reads. Devices of one model differ by up to
// 12 % on such loops (same section, item 5): a bench line's VALU fraction against the nominal 2.4 GHz cannot tell a slow device from a slow
// kernel — this number can.
__global__ void __launch_bounds__(256) k_clock_probe(uint4* __restrict__ stamps, u32* __restrict__ sink, u32 iters) {
    u32 h[8], m[16];
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = tid * 0x9E3779B9u + i;
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = (tid ^ 0x5BD1E995u) * (2u * i + 1u);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (u32 it = 0; it < iters; it++) { blake2s_compress(h, m, it, 0); m[it & 15] ^= h[0]; }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= h[i];
    sink[tid] = acc;
    if (threadIdx.x == 0) stamps[blockIdx.x] = uint4{(u32)(c1 - c0), (u32)((c1 - c0) >> 32), (u32)(r1 - r0), (u32)((r1 - r0) >> 32)};
}
// Sidecar of bfhip_clock_probe_mix: ONE wave that sleeps beside the real Merkle kernel (launched on the context's other stream) and stamps the shader-cycle counter
// against the 100 MHz counter when it starts and when the host raises `*stop` (pinned memory) — or after max_ticks of the 100 MHz counter, whichever comes first. The
// clock domain is the chip's, so the quotient is the clock the device holds under the REAL kernel's mix of VALU and memory traffic, with no stamp in the kernel itself.
__global__ void k_clock_sampler(uint4* __restrict__ out, const volatile u32* __restrict__ stop, unsigned long long max_ticks) {
    if (threadIdx.x) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    u32 stopped = 0;
    while (r1 - r0 < max_ticks) {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
        if (*stop) { stopped = 1; break; }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = uint4{(u32)(c1 - c0), (u32)((c1 - c0) >> 32), (u32)(r1 - r0), (u32)((r1 - r0) >> 32)};
    out[1] = uint4{stopped, 0u, 0u, 0u};
}
void clock_sampler_launch(hipStream_t stream, uint4* d_out, const u32* d_stop_alias, unsigned long long max_ticks) {
    hipLaunchKernelGGL(k_clock_sampler, dim3(1), dim3(64), 0, stream, d_out, d_stop_alias, max_ticks);
}
// pseudo-random fill of a hash layer (the probe hashes "random data": the clock a device holds depends on the operands)
__global__ void k_fill_mix(u32* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32 x = (u32)i * 0x9E3779B9u + 0x7F4A7C15u; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15; p[i] = x; }
}
void fill_mix(hipStream_t stream, u32* p, size_t n) { hipLaunchKernelGGL(k_fill_mix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n); }
void clock_probe_launch(hipStream_t stream, uint4* d_stamps, u32* d_sink, u32 blocks, u32 iters) {
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, stream, d_stamps, d_sink, iters);
}

}  // namespace bf
