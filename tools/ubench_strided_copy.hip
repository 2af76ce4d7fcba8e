// Micro-benchmark: HBM throughput of the strided-tile access pattern of k_fft_strided7 (rows of CHUNK bytes at stride 2^lo words),
// copy only (no butterflies). Answers: is 128-byte row granularity the limiter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
// tile: 128 rows x (CH words); grid walks (H, Lhi) exactly like the FFT kernel. in-place read-modify-write.
template <int CHW>  // chunk words: 32 (128 B) or 64 (256 B) or 128 (512 B)
__global__ void __launch_bounds__(256) k_copy(u32* __restrict__ p, u32 lo, u32 log) {
    constexpr int LANES_PER_ROW = CHW / 4;
    constexpr int ROWS_PER_PASS = 256 / LANES_PER_ROW;
    u32 t = threadIdx.x, tile = blockIdx.x;
    u32 c = __builtin_ctz(CHW);
    u32 n_lhi_log = lo - c;
    u32 H = tile >> n_lhi_log, Lhi = tile & ((1u << n_lhi_log) - 1);
    u32 base = (H << (lo + 7)) | (Lhi << c);
    u32 l4 = 4 * (t % LANES_PER_ROW), r0 = t / LANES_PER_ROW;
    uint4 v[128 / ROWS_PER_PASS];
#pragma unroll
    for (int q = 0; q < 128 / ROWS_PER_PASS; q++) v[q] = *reinterpret_cast<uint4*>(p + (base | ((r0 + q * ROWS_PER_PASS) << lo) | l4));
#pragma unroll
    for (int q = 0; q < 128 / ROWS_PER_PASS; q++) { v[q].x += 1; *reinterpret_cast<uint4*>(p + (base | ((r0 + q * ROWS_PER_PASS) << lo) | l4)) = v[q]; }
}
__global__ void __launch_bounds__(256) k_copy_contig(uint4* __restrict__ p, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = p[i + q * 256];
#pragma unroll
    for (int q = 0; q < 4; q++) { v[q].x += 1; p[i + q * 256] = v[q]; }
}
template <int CHW> void run(u32* d, u32 log, u32 lo, int ncols) {
    size_t n = (size_t)1 << log;
    u32 tiles = (u32)(n / (128 * CHW));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int c = 0; c < ncols; c++) k_copy<CHW><<<tiles, 256>>>(d + c * n, lo, log);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int c = 0; c < ncols; c++) k_copy<CHW><<<tiles, 256>>>(d + c * n, lo, log);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rows of %4d B at stride 2^%u words: %.0f GB/s moved\n", CHW * 4, lo, 8.0 * n * ncols / ms / 1e6);
}
int main() {
    const u32 log = 24; const int ncols = 32;
    u32* d; hipMalloc(&d, ((size_t)4 << log) * ncols); hipMemset(d, 0, ((size_t)4 << log) * ncols);
    {
        size_t n4 = ((size_t)1 << log) * ncols / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_copy_contig<<<n4 / 1024, 256>>>((uint4*)d, n4); hipDeviceSynchronize();
        hipEventRecord(e0); k_copy_contig<<<n4 / 1024, 256>>>((uint4*)d, n4); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("contiguous read+write: %.0f GB/s moved\n", 32.0 * n4 / ms / 1e6);
    }
    for (u32 lo : {12u, 17u}) { run<32>(d, log, lo, ncols); run<64>(d, log, lo, ncols); run<128>(d, log, lo, ncols); }
    return 0;
}
