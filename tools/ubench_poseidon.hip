// Micro-benchmarks behind the bound analysis of k_merkle_layer_poseidon: (1) issue rate of v_mad_u64_u32, the instruction a
// 256-bit Montgomery product is made of; (2) Hades permutations per second of the kernel's own field code with the state in registers
// (no memory traffic) — the compute ceiling of the Poseidon252 Merkle layer.   hipcc -O3 --offload-arch=gfx950 -I stwo-brainfuck_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "poseidon_constants.h"
#include "poseidon_dev.h"
using namespace bf;

__global__ void __launch_bounds__(256) k_mad(u32* out, int iters) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3;
    u32 x = tid * 2654435761u | 1, y = tid * 40503u + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {   // 4 independent chains: d = x * y + d (64-bit accumulate)
            a0 = (u64)x * y + a0; a1 = (u64)y * (u32)a0 + a1; a2 = (u64)x * (u32)a1 + a2; a3 = (u64)y * (u32)a2 + a3;
        }
    }
    out[tid] = (u32)(a0 ^ a1 ^ a2 ^ a3) ^ (u32)((a0 ^ a1 ^ a2 ^ a3) >> 32);
}
__global__ void __launch_bounds__(128) k_hades(u32* out, const u32* __restrict__ consts, int iters) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const PoseidonConsts pc(consts);
    F9 s[3];
    for (int k = 0; k < 3; k++) for (int i = 0; i < 9; i++) s[k].l[i] = (tid * 31 + 9 * k + i) & (i == 8 ? 0x000fffffu : M29);
    for (int it = 0; it < iters; it++) hades(s, pc.table);
    u32 acc = 0; for (int k = 0; k < 3; k++) for (int i = 0; i < 9; i++) acc ^= s[k].l[i];
    out[tid] = acc;
}
int main() {
    u32* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    { int blocks = 4096, iters = 1024;
      k_mad<<<blocks, 256>>>(d, 4); hipDeviceSynchronize();
      hipEventRecord(e0); k_mad<<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      double n = (double)blocks * 256 * iters * 64;
      printf("v_mad_u64_u32: %.2f T/s chip-wide (%.2f lanes per clock per CU at 2.4 GHz)\n", n / ms / 1e9, n / (ms * 1e-3) / 256 / 2.4e9); }
    { std::vector<u32> h; h.insert(h.end(), POSEIDON_P, POSEIDON_P + 8); h.insert(h.end(), POSEIDON_DEV_R1, POSEIDON_DEV_R1 + 9); h.insert(h.end(), POSEIDON_DEV_R2, POSEIDON_DEV_R2 + 9);
      for (int r = 0; r < 92; r++) h.insert(h.end(), POSEIDON_DEV_ROUNDS[r], POSEIDON_DEV_ROUNDS[r] + F9_ROUND_WORDS);
      u32* c; hipMalloc(&c, h.size() * 4); hipMemcpy(c, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      int blocks = 2048, iters = 8;
      k_hades<<<blocks, 128>>>(d, c, 1); hipDeviceSynchronize();
      hipEventRecord(e0); k_hades<<<blocks, 128>>>(d, c, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      double n = (double)blocks * 128 * iters;
      printf("Hades permutations (registers only): %.1f M/s  (214 Montgomery products each -> %.2f G products/s)\n", n / ms / 1e3, n * 214 / ms / 1e6); }
    return 0;
}
