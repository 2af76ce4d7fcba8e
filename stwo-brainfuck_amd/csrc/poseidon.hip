// Poseidon252 Merkle layers for gfx950 — SURVEY.md §8(f)3 / BASELINE.json config 5 (`MerkleOps<Poseidon252MerkleHasher>`).
// The reference never uses this hasher (no `Poseidon` token in crates/); it is an upstream stwo capability over
// starknet-crypto's `poseidon_hash_many`. node(i) = poseidon_hash_many([left, right]? ++ blocks), a block = 8 M31 values of the
// layer's columns packed as w = w * 2^31 + v (zero padded to a multiple of 8).
//
// Arithmetic: the Stark field p = 2^251 + 17 * 2^192 + 1, Montgomery form with R = 2^261: products on 9 limbs of 29 bits (one u64
// accumulator per column, no carry instructions; p = 1 (mod 2^29) and m * p touches only columns i, i + 6, i + 8), the linear layer on
// 8 x 32-bit words (poseidon_dev.h). One Hades permutation (width 3, x^3 S-box,
// 4 full + 83 partial + 4 full rounds, MDS [[3,1,1],[1,-1,1],[1,1,-2]]) is 214 field multiplications ≈ 90 k lane-ops: this hasher
// is ~90x more VALU work per 64 hashed bytes than Blake2s — firmly VALU-bound. The permutation is pinned by the public Hades([0,0,0])
// known-answer vector (tests/test_gpu_poseidon.py); the node layout is recalled from stwo (unpinned).
#include "kernels.h"
#include "poseidon_constants.h"
#include "poseidon_dev.h"
#include <vector>
#include <stdexcept>
#include <mutex>

namespace bf {

// Replication-aware and range-aware like k_merkle_layer (merkle.hip): node i is stored at slot i >> out_shift, its children at
// (2i) >> prev_shift and (2i + 1) >> prev_shift; [first, first + n_stored) is the range of stored slots this launch computes (the whole
// layer, or one rank's share of it in a shard group). consts: the PoseidonConsts block (poseidon_dev.h).
__global__ void __launch_bounds__(128, 4) k_merkle_layer_poseidon(u32* __restrict__ out, const u32* __restrict__ prev, const ColDesc* __restrict__ cols, u32 ncols, u32 n_stored,
                                                              u32 out_shift, u32 prev_shift, u32 first, const u32* __restrict__ consts) {
    u32 st = blockIdx.x * blockDim.x + threadIdx.x;
    if (st >= n_stored) return;
    st += first;
    const u32 i = st << out_shift;          // representative node of this stored slot
    const PoseidonConsts pc(consts);
    Sponge sp(pc);
    if (prev) {
        const size_t cl = ((size_t)2 * i) >> prev_shift, cr = ((size_t)2 * i + 1) >> prev_shift;
        Fe l, r;
        for (int k = 0; k < 8; k++) { l.l[k] = prev[8 * cl + k]; r.l[k] = prev[8 * cr + k]; }
        sp.absorb(f9_from_canonical(l, pc));
        sp.absorb(f9_from_canonical(r, pc));
    }
    for (u32 c0 = 0; c0 < ncols; c0 += 8) {
        // w = sum_k v_k * 2^(31 * (7 - k)): value k occupies bits [31 (7-k), 31 (8-k))  (< 2^248 < p, no reduction needed)
        Fe w; for (int k = 0; k < 8; k++) w.l[k] = 0;
        for (u32 k = 0; k < 8; k++) {
            u32 c = c0 + k;
            u32 v = 0;
            if (c < ncols) { ColDesc cd = cols[c]; v = ld_col(cd, i); }
            u32 sh = 31 * (7 - k), limb = sh >> 5, off = sh & 31;
            w.l[limb] |= v << off;
            if (off > 1 && limb + 1 < 8) w.l[limb + 1] |= v >> (32 - off);
        }
        sp.absorb(f9_from_canonical(w, pc));
    }
    const Fe h = f9_to_canonical(sp.finish(), pc);
    for (int k = 0; k < 8; k++) out[(size_t)8 * st + k] = h.l[k];
}

// Test hook: one Hades permutation of 3 canonical field elements (24 words in, 24 words out).
__global__ void k_hades_once(const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ consts) {
    if (threadIdx.x || blockIdx.x) return;
    const PoseidonConsts pc(consts);
    F9 s[3];
    for (int k = 0; k < 3; k++) { Fe x; for (int i = 0; i < 8; i++) x.l[i] = in[8 * k + i]; s[k] = f9_from_canonical(x, pc); }
    hades(s, pc.table);
    for (int k = 0; k < 3; k++) { const Fe y = f9_to_canonical(s[k], pc); for (int i = 0; i < 8; i++) out[8 * k + i] = y.l[i]; }
}

static u32* g_poseidon_consts[64] = {nullptr};   // per device; tiny (8.8 KB each), kept for the life of the process
static std::mutex g_poseidon_mutex;
static const u32* poseidon_consts() {
    std::lock_guard<std::mutex> guard(g_poseidon_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) throw std::runtime_error("hipGetDevice(poseidon constants)");
    if (!g_poseidon_consts[dev]) {
        std::vector<u32> h;
        h.insert(h.end(), POSEIDON_P, POSEIDON_P + 8); h.insert(h.end(), POSEIDON_DEV_R1, POSEIDON_DEV_R1 + 9); h.insert(h.end(), POSEIDON_DEV_R2, POSEIDON_DEV_R2 + 9);
        for (int r = 0; r < 92; r++) h.insert(h.end(), POSEIDON_DEV_ROUNDS[r], POSEIDON_DEV_ROUNDS[r] + F9_ROUND_WORDS);
        if (h.size() != POSEIDON_CONSTS_WORDS) throw std::runtime_error("poseidon constants: layout mismatch");
        u32* p = nullptr;
        if (hipMalloc((void**)&p, h.size() * sizeof(u32)) != hipSuccess) throw std::runtime_error("hipMalloc(poseidon constants)");
        if (hipMemcpy(p, h.data(), h.size() * sizeof(u32), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); throw std::runtime_error("hipMemcpy(poseidon constants)"); }
        g_poseidon_consts[dev] = p;
    }
    return g_poseidon_consts[dev];
}

void merkle_layer_poseidon(hipStream_t stream, void* out, const void* prev, const ColDesc* d_cols, u32 ncols, u32 log, u32 out_shift, u32 prev_shift, u32 first, u32 count) {
    const u32 total = (1u << log) >> out_shift;
    const u32 n = count ? count : total;
    u32 threads = n < 128 ? (n < 64 ? 64 : n) : 128;
    // Hades permutations per node: poseidon_hash_many over (2 children +) ceil(ncols / 8) blocks = floor(elements / 2) + 1
    const u32 elems = (prev ? 2u : 0u) + (ncols + 7) / 8;
    ProfScope ps(stream, "k_merkle_layer_poseidon", ((prev ? 64.0 : 0.0) + 32.0 + 4.0 * ncols) * n, (double)(elems / 2 + 1) * n, /*dominant=*/true);
    hipLaunchKernelGGL(k_merkle_layer_poseidon, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, (u32*)out, (const u32*)prev, d_cols, ncols, n, out_shift, prev_shift,
                       count ? first : 0u, poseidon_consts());
}
void hades_once(hipStream_t stream, const u32* d_in24, u32* d_out24) {
    hipLaunchKernelGGL(k_hades_once, dim3(1), dim3(64), 0, stream, d_in24, d_out24, poseidon_consts());
}

}  // namespace bf
