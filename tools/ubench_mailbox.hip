// Micro-benchmark of one Fiat-Shamir round trip (GPU result -> host -> challenge-dependent table -> next kernel), the ~30 us idle gaps of a
// proof (profiles/r04_2p22_timeline_gaps.txt). Variants, each timed as wall time per round trip over a chain of dependent round trips:
//   V0  today's order: kernel writes its result into pinned memory, host polls an event, builds the table, hipMemcpyAsync into device
//       memory, launches the consumer.
//   V1  mailbox: the consumer (and a one-workgroup kernel in front of it that waits for a flag in pinned memory and then copies the table
//       from the pinned ring into device memory) is enqueued BEFORE the host waits; the host polls the event, writes the table, sets the flag.
//   V2  as V1, the host polls a sequence stamp the producer writes behind its result instead of an event.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/ubench_mailbox tools/ubench_mailbox.hip && /tmp/ubench_mailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_produce(const u32* __restrict__ in, u32* result_pinned, u32 seq) {
    // stands for the top of a Merkle tree: one workgroup, result (8 words) straight into pinned memory, then a stamp
    const u32 t = threadIdx.x;
    u32 v = in[t & 63] + seq;
    for (int i = 0; i < 200; i++) v = v * 1664525u + 1013904223u;     // ~1 us of dependent work
    if (t < 8) result_pinned[t] = v + t;
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(result_pinned + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_consume(const u32* __restrict__ table, u32* out, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[i & 255] + i;
}
__global__ void __launch_bounds__(256) k_mailbox(const u32* flag, u32 expect, const u32* __restrict__ src_pinned, u32* __restrict__ dst, u32 n_words, u32* err) {
    __shared__ u32 ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        u32 good = 1;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != expect) {
            if (wall_clock64() - t0 > 20000000ull) { good = 0; break; }          // 0.2 s at 100 MHz: give up, never hang the queue
            __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
        if (!good) *err = expect;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < n_words; i += blockDim.x) dst[i] = __builtin_nontemporal_load(src_pinned + i);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    u32 *h_res, *d_res_alias, *h_ring, *d_ring_alias, *d_table, *d_in, *d_out, *h_flag, *d_flag_alias, *d_err;
    CK(hipHostMalloc((void**)&h_res, 4096)); CK(hipHostGetDevicePointer((void**)&d_res_alias, h_res, 0));
    CK(hipHostMalloc((void**)&h_ring, 1 << 16)); CK(hipHostGetDevicePointer((void**)&d_ring_alias, h_ring, 0));
    CK(hipHostMalloc((void**)&h_flag, 4096)); CK(hipHostGetDevicePointer((void**)&d_flag_alias, h_flag, 0));
    CK(hipMalloc((void**)&d_table, 1 << 16)); CK(hipMalloc((void**)&d_in, 4096)); CK(hipMalloc((void**)&d_out, 4 << 20)); CK(hipMalloc((void**)&d_err, 64));
    CK(hipMemset(d_in, 1, 4096)); CK(hipMemset(d_err, 0, 64)); memset(h_res, 0, 4096); memset(h_flag, 0, 4096);
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const u32 n_out = 1 << 20, table_words = 1024;     // 4 KiB table
    auto host_work = [&](u32 seq) { for (u32 i = 0; i < table_words; i++) h_ring[i] = h_res[i & 7] * 2654435761u + seq + i; };   // ~1 us
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int reps = 400;
    u32 seq = 0;
    for (int variant = 0; variant < 3; variant++) {
        for (int pass = 0; pass < 2; pass++) {       // pass 0 warms up
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int r = 0; r < reps; r++) {
                seq++;
                hipLaunchKernelGGL(k_produce, dim3(1), dim3(256), 0, s, d_in, d_res_alias, seq);
                if (variant == 0) {
                    CK(hipEventRecord(ev, s));
                    while (hipEventQuery(ev) == hipErrorNotReady) {}
                    host_work(seq);
                    CK(hipMemcpyAsync(d_table, h_ring, table_words * 4, hipMemcpyHostToDevice, s));
                    hipLaunchKernelGGL(k_consume, dim3(n_out / 256), dim3(256), 0, s, d_table, d_out, n_out);
                } else {
                    if (variant == 1) CK(hipEventRecord(ev, s));
                    hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(256), 0, s, d_flag_alias, seq, d_ring_alias, d_table, table_words, d_err);
                    hipLaunchKernelGGL(k_consume, dim3(n_out / 256), dim3(256), 0, s, d_table, d_out, n_out);
                    if (variant == 1) { while (hipEventQuery(ev) == hipErrorNotReady) {} }
                    else { while (__atomic_load_n(h_res + 8, __ATOMIC_ACQUIRE) != seq) {} }
                    host_work(seq);
                    __atomic_store_n(h_flag, seq, __ATOMIC_RELEASE);
                }
            }
            CK(hipStreamSynchronize(s));
            const double dt = now() - t0;
            if (pass) printf("V%d  %.2f us per round trip (produce ~1-2 us + consume ~2 us of kernel time included)\n", variant, dt / reps * 1e6);
        }
    }
    u32 err[16]; CK(hipMemcpy(err, d_err, 64, hipMemcpyDeviceToHost));
    printf("mailbox timeouts: %u\n", err[0]);
    // floor: the same two kernels back to back with no host in between
    {
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_produce, dim3(1), dim3(256), 0, s, d_in, d_res_alias, ++seq);
            hipLaunchKernelGGL(k_consume, dim3(n_out / 256), dim3(256), 0, s, d_table, d_out, n_out);
        }
        CK(hipStreamSynchronize(s));
        printf("floor (no host round trip)  %.2f us per pair\n", (now() - t0) / reps * 1e6);
    }
    return 0;
}
