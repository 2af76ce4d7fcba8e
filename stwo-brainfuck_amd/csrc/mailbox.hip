// Mailboxes: the launches that FOLLOW a Fiat-Shamir point are enqueued BEFORE the host knows the challenge (r04). Used by default for proofs
// with LOG_MAX_ROWS <= 21 — measured, it pays on small proofs only —; BFHIP_MAILBOX=1 / 0 forces it (ctx.h: mailbox_mode).
//
// At a Fiat-Shamir point the host must see a result of the GPU (a root, the sampled values), step the channel, and hand the next kernels the
// challenge-dependent part of their parameter tables. Done in that order — wait, compute, copy, launch — the GPU idles for the host's wake-up,
// the launch latency of a copy and of the first kernel: ~30 us, five times per proof (profiles/r04_2p22_timeline_gaps.txt). But the launches
// themselves (grids, pointers, the structure of every table) depend on the trace's shape only. So the next phase is enqueued right behind the
// previous one, with ONE small kernel in front of it:
//
//   k_mailbox   one workgroup; lane 0 polls a flag word in pinned host memory until it carries this proof's number, then the workgroup copies
//               the phase's parameter blocks from the pinned staging ring to their device copy (what the hipMemcpyAsync of a staging batch
//               would have done). The blocks were written to the ring when the phase was enqueued, structurally complete; the host patches
//               the challenge-dependent words in place and sets the flag (Mailbox::post, a release store).
//   k_post_stamp  one lane; writes this proof's number into a pinned word behind a phase whose result the host waits for. Polling that word
//               costs the host ~2 us; polling an event behind the same kernel ~8 us (tools/ubench_mailbox.hip: 13.9 vs 20.4 us per round trip). The top
//               kernel of a tree writes the stamp itself, behind the root (merkle.hip: k_merkle_top).
//
// The queue can never hang on a host that died: after ten seconds (BFHIP_MAILBOX_TIMEOUT_MS) without the flag the kernel gives up, reports
// through a pinned error word, and copies what is in the ring — structurally valid blocks with stale challenge words, so the kernels behind it
// compute garbage on valid addresses; the proof fails with an error when the host reads the word. A Mailbox that goes out of scope un-posted
// (an exception between enqueue and challenge) posts itself for the same reason.
#include "ctx.h"

namespace bf {

__global__ void __launch_bounds__(256) k_mailbox(const u32* flag, u32 expect, const uint4* __restrict__ src_pinned, uint4* __restrict__ dst, u32 n16, u32* err_pinned, u32 timeout_ticks_hi) {      // err_pinned[0]: gave up; [1]: ticks waited
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64(), limit = (unsigned long long)timeout_ticks_hi << 16;      // 100 MHz ticks
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != expect) {
            if (wall_clock64() - t0 > limit) { __hip_atomic_store(err_pinned, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        err_pinned[1] = (u32)(wall_clock64() - t0);        // 100 MHz ticks spent waiting for the host (BFHIP_TRACE_HOST prints them)
    }
    __syncthreads();
    // the ring is fine-grained (uncached) host memory and lane 0's acquire precedes the barrier: plain 16-byte loads see the host's stores
    for (u32 i = threadIdx.x; i < n16; i += blockDim.x) { const bf_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf_u32x4*>(src_pinned) + i); dst[i] = make_uint4(v.x, v.y, v.z, v.w); }
}
__global__ void k_post_stamp(u32* stamp_pinned, u32 value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(stamp_pinned, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void mailbox_launch(hipStream_t s, const u32* d_flag, u32 expect, const void* src_pinned_alias, void* dst, size_t bytes, u32* d_err, double timeout_seconds) {
    const u32 n16 = (u32)((bytes + 15) / 16);
    const unsigned long long ticks = (unsigned long long)(timeout_seconds * 1e8);
    hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(256), 0, s, d_flag, expect, reinterpret_cast<const uint4*>(src_pinned_alias), reinterpret_cast<uint4*>(dst), n16, d_err, (u32)((ticks >> 16) + 1));
}
void post_stamp_launch(hipStream_t s, u32* d_stamp, u32 value) { hipLaunchKernelGGL(k_post_stamp, dim3(1), dim3(64), 0, s, d_stamp, value); }

}  // namespace bf
