// Micro-benchmark: M31 butterfly throughput in registers on gfx950 (which mulmod formulation is cheapest?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
#define P31 0x7fffffffu
__device__ __forceinline__ u32 m_add(u32 a, u32 b) { u32 s = a + b; u32 t = s - P31; return t < s ? t : s; }
__device__ __forceinline__ u32 m_sub(u32 a, u32 b) { u32 s = a - b; u32 t = s + P31; return t < s ? t : s; }
__device__ __forceinline__ u32 mul_a(u32 a, u32 b) { u64 x = (u64)a * b; u32 lo = (u32)x & P31, hi = (u32)(x >> 31); u32 s = lo + hi; u32 t = s - P31; return t < s ? t : s; }
// twiddle pre-doubled (b2 = 2b): hi word of a*b2 is (a*b)>>31 directly, lo word >> 1 is (a*b) & P
__device__ __forceinline__ u32 mul_b(u32 a, u32 b2) { u32 hi = __umulhi(a, b2); u32 lo = (a * b2) >> 1; u32 s = lo + hi; u32 t = s - P31; return t < s ? t : s; }
// 16-bit split with 24-bit multipliers is not applicable (operands are 31 bits); fp64 variant:
__device__ __forceinline__ u32 mul_c(u32 a, u32 b) {
    double p = (double)a * (double)b;                 // rounded
    double q = __builtin_floor(p * (1.0 / 2147483647.0));
    double r = __builtin_fma(-q, 2147483647.0, p);    // not exact for 62-bit products: shown for rate only
    return (u32)(long long)r;
}
template <int V>
__global__ void __launch_bounds__(256) k(u32* out, int iters) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 v[8], t[4];
    for (int i = 0; i < 8; i++) v[i] = (tid * 2654435761u + i * 40503u) & P31;
    for (int i = 0; i < 4; i++) t[i] = (tid * 97u + i * 7919u + 12345u) & P31;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u32 a = v[i], b = v[i + 4];
            u32 w = V == 0 ? mul_a(b, t[i]) : V == 1 ? mul_b(b, t[i]) : V == 2 ? mul_c(b, t[i]) : b;
            v[i] = m_add(a, w); v[i + 4] = m_sub(a, w);
        }
        u32 x = v[0]; for (int i = 0; i < 7; i++) v[i] = v[i + 1]; v[7] = x;
    }
    u32 acc = 0; for (int i = 0; i < 8; i++) acc ^= v[i];
    out[tid] = acc;
}
template <int V> void run(const char* name) {
    int blocks = 4096, iters = 2048;
    u32* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(d, 8); hipDeviceSynchronize();
    hipEventRecord(e0); k<V><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * iters * 4;
    printf("%-28s %.1f G butterflies/s\n", name, n / ms / 1e6);
    hipFree(d);
}
int main() { run<3>("add/sub only (no mul)"); run<0>("u64 product + fold"); run<1>("mulhi + mullo (2t twiddle)"); run<2>("fp64 (rate probe)"); return 0; }
