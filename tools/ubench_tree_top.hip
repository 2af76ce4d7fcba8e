// Micro-benchmark of the latency chain at the small end of a Merkle tree (k_merkle_subtree: levels 17..9, k_merkle_top: levels 8..0 +
// channel step): where do the ~42 + ~30 us per tree go? Times each kernel alone on a synthetic column-less tree,
//   warm  = back-to-back launches of the same kernel (instruction cache and descriptors hot),
//   cold  = after a large launch of a DIFFERENT kernel (k_merkle_layer over 2^22 nodes) — the situation inside a proof,
// and k_merkle_top for every first level 0..8, so that per-level cost and fixed cost separate.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stwo-brainfuck_amd/csrc -o /tmp/ubench_tree_top tools/ubench_tree_top.hip && /tmp/ubench_tree_top
#include "../stwo-brainfuck_amd/csrc/merkle.hip"
#include "../stwo-brainfuck_amd/csrc/prof.hip"
#include <cstdio>
#include <vector>
using namespace bf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const u32 max_log = 18;
    hipStream_t s; CK(hipStreamCreate(&s));
    MerkleTreeDesc td{};
    std::vector<uint4*> layers(max_log + 2);
    for (u32 lg = 0; lg <= max_log; lg++) { CK(hipMalloc((void**)&layers[lg], (size_t)32 << lg)); CK(hipMemset(layers[lg], 0x5a, (size_t)32 << lg)); td.layers[lg] = layers[lg]; td.shifts[lg] = 0; td.col_off[lg] = 0; }
    td.cols = nullptr; td.n_cols = 0; td.max_log = max_log;
    uint4* big; CK(hipMalloc((void**)&big, (size_t)32 << 24)); CK(hipMemset(big, 1, (size_t)32 << 24));   // level 22 (128 MiB) followed by its children, level 23 (256 MiB)
    u32* chan; CK(hipMalloc((void**)&chan, 64 * 4)); CK(hipMemset(chan, 0, 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto thrash = [&]() { merkle_layer(s, big, big + ((size_t)2 << 22), nullptr, 0, 22, 0.0, 0, 0, 0); };
    auto timed = [&](auto launch, bool cold, int reps) -> double {
        double tot = 0;
        for (int r = 0; r < reps; r++) {
            if (cold) thrash();
            (void)hipEventRecord(e0, s); launch(); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); tot += ms;
        }
        return tot / reps * 1e3;
    };
    // warm-up (code objects)
    merkle_subtree(s, td, 17, 0, 0, 0); merkle_top(s, td, 8, 0, chan, chan + 16, chan + 32, 0, 0); thrash(); CK(hipStreamSynchronize(s));
    printf("kernel                          warm us   cold us   (HIP events around one launch; includes ~2-3 us of event overhead)\n");
    for (u32 hi : {17u, 15u, 13u, 11u}) {
        auto f = [&]() { merkle_subtree(s, td, hi, 0, 0, 0); };
        printf("k_merkle_subtree hi=%2u (-> 9)  %8.1f  %8.1f\n", hi, timed(f, false, 20), timed(f, true, 20));
    }
    for (int top_hi = 8; top_hi >= 0; top_hi -= 2) {
        auto f = [&]() { merkle_top(s, td, (u32)top_hi, 0, nullptr, nullptr, nullptr, 0, 0); };
        printf("k_merkle_top top_hi=%d (-> 0)   %8.1f  %8.1f\n", top_hi, timed(f, false, 20), timed(f, true, 20));
    }
    {
        auto f = [&]() { merkle_top(s, td, 8, 0, chan, chan + 16, chan + 32, 0, 0); };
        printf("k_merkle_top 8 + channel step  %8.1f  %8.1f\n", timed(f, false, 20), timed(f, true, 20));
        auto g = [&]() { hipLaunchKernelGGL(k_channel_mix_root_draw, dim3(1), dim3(64), 0, s, chan, (const u32*)layers[0], chan + 16, chan + 32); };
        printf("k_channel_mix_root_draw        %8.1f  %8.1f\n", timed(g, false, 20), timed(g, true, 20));
        auto e = [&]() {};
        printf("empty (event pair only)        %8.1f  %8.1f\n", timed(e, false, 20), timed(e, true, 20));
    }
    return 0;
}
