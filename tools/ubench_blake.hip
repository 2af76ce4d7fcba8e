// Micro-benchmark: Blake2s compression throughput on gfx950 (registers only, no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
__device__ __forceinline__ u32 rotr(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }
#define G(a, b, c, d, x, y) a = a + b + (x); d = rotr(d ^ a, 16); c = c + d; b = rotr(b ^ c, 12); a = a + b + (y); d = rotr(d ^ a, 8); c = c + d; b = rotr(b ^ c, 7);
#define ROUND(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
  G(v0,v4,v8,v12,m[s0],m[s1]) G(v1,v5,v9,v13,m[s2],m[s3]) G(v2,v6,v10,v14,m[s4],m[s5]) G(v3,v7,v11,v15,m[s6],m[s7]) \
  G(v0,v5,v10,v15,m[s8],m[s9]) G(v1,v6,v11,v12,m[s10],m[s11]) G(v2,v7,v8,v13,m[s12],m[s13]) G(v3,v4,v9,v14,m[s14],m[s15])
__device__ __forceinline__ void compress(u32 h[8], const u32 m[16], u32 t0, u32 f0) {
  u32 v0=h[0],v1=h[1],v2=h[2],v3=h[3],v4=h[4],v5=h[5],v6=h[6],v7=h[7];
  u32 v8=0x6A09E667u,v9=0xBB67AE85u,v10=0x3C6EF372u,v11=0xA54FF53Au,v12=0x510E527Fu^t0,v13=0x9B05688Cu,v14=0x1F83D9ABu^f0,v15=0x5BE0CD19u;
  ROUND(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15) ROUND(14,10,4,8,9,15,13,6,1,12,0,2,11,7,5,3) ROUND(11,8,12,0,5,2,15,13,10,14,3,6,7,1,9,4)
  ROUND(7,9,3,1,13,12,11,14,2,6,5,10,4,0,15,8) ROUND(9,0,5,7,2,4,10,15,14,1,11,12,6,8,3,13) ROUND(2,12,6,10,0,11,8,3,4,13,7,5,15,14,1,9)
  ROUND(12,5,1,15,14,13,4,10,0,7,6,3,9,2,8,11) ROUND(13,11,7,14,12,1,3,9,5,0,15,4,8,6,2,10) ROUND(6,15,14,9,11,3,0,8,12,2,13,7,1,4,10,5)
  ROUND(10,2,8,4,7,6,1,5,15,11,9,14,3,12,13,0)
  h[0]^=v0^v8;h[1]^=v1^v9;h[2]^=v2^v10;h[3]^=v3^v11;h[4]^=v4^v12;h[5]^=v5^v13;h[6]^=v6^v14;h[7]^=v7^v15;
}
template <int NH>
__global__ void __launch_bounds__(256) k_bench(u32* out, int iters) {
  u32 h[NH][8], m[NH][16];
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int q = 0; q < NH; q++) { for (int i = 0; i < 8; i++) h[q][i] = tid * 31 + i + q; for (int i = 0; i < 16; i++) m[q][i] = tid + i * 7 + q; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < NH; q++) { compress(h[q], m[q], it, 0); m[q][it & 15] ^= h[q][0]; }
  }
  u32 acc = 0; for (int q = 0; q < NH; q++) for (int i = 0; i < 8; i++) acc ^= h[q][i];
  out[tid] = acc;
}
template <int NH> void run(const char* name, int blocks, int iters) {
  u32* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_bench<NH><<<blocks, 256>>>(d, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0); k_bench<NH><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)blocks * 256 * iters * NH;
  printf("%s blocks=%d: %.2f G compress/s (%.2f T ops/s @977 ops)\n", name, blocks, n / ms / 1e6, n * 977 / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int blocks : {256 * 4, 256 * 8, 256 * 16}) { run<1>("1 hash/thread", blocks, 256); run<2>("2 hash/thread", blocks, 256); }
  return 0;
}
