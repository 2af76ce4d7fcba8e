// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). Used by bench.py's `cpu_baseline.simd_bound` leg and by tests/, never by the product.
//
// What could the reference's CPU path reach on THIS host? The reference proves with stwo's SimdBackend + rayon (`--features parallel`,
// README.md:23-36; Cargo.toml:50-53; the time it prints: bin/brainfuck_prover.rs:137-139) and cannot be built in this image (no cargo, stwo not
// vendored). The scalar OpenMP port (oracle/prover.h) says nothing about it. This file measures the two inner loops a SimdBackend-shaped
// prover cannot do without, at the rate the host's vector units sustain when EVERY core runs them out of registers / L1:
//
//   blake2s_x16   the 16-lane Blake2s compression (stwo `compress16`: one u32x16 per state / message word, the Merkle and channel hash) —
//                 AVX-512 when the host has it, else two AVX2 halves per 16 lanes;
//   m31_butterfly the radix-2 circle-FFT butterfly on 16 packed M31 lanes (a + b t, a - b t with the 32x32 -> 64-bit multiply and the
//                 Mersenne reduction stwo's PackedM31 uses).
//
// From the rates: T_bound(workload) = compressions / (compressions per s) + butterflies / (butterflies per s). Everything else the
// reference does (constraint evaluation, quotients, the logUp columns, memory traffic, rayon's scheduling) only adds time, and its
// SimdBackend does not know the 16x replication of the trace columns that the HIP path exploits. So T_bound is a LOWER bound of the reference's
// proving time on this host with SimdBackend-shaped code, and GPU-time / T_bound is a lower bound of the speedup: if even that ratio
// exceeds 10 the north-star's ">= 10x the parallel CPU prover" is met whatever the real reference does; if not, the bound does not decide it.
#include <cstdint>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
#include <immintrin.h>
#include "simd_kernels.h"

namespace {

// `iters` chained compressions of 16 lanes (the output feeds the next message: nothing can be hoisted); returns a checksum
__attribute__((target("avx512f,avx512bw,avx512vl"))) uint32_t blake_loop_avx512(uint64_t iters, uint32_t seed) {
    v16u h[8], m[16];
    for (int i = 0; i < 8; i++) for (int l = 0; l < 16; l++) h[i][l] = IV[i] ^ (seed + 131u * l + i);
    for (int i = 0; i < 16; i++) for (int l = 0; l < 16; l++) m[i][l] = seed * 2654435761u + 977u * l + i;
    for (uint64_t k = 0; k < iters; k++) {
        compress_avx512(h, m, 64, 0);
        m[0] ^= h[0]; m[5] ^= h[3]; m[10] ^= h[5]; m[15] ^= h[7];          // the next message depends on this digest
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) for (int l = 0; l < 16; l++) acc ^= h[i][l];
    return acc;
}
__attribute__((target("avx2"))) uint32_t blake_loop_avx2(uint64_t iters, uint32_t seed) {
    v8u h[2][8], m[2][16];       // two halves = 16 lanes per iteration, like the other path
    for (int p = 0; p < 2; p++) {
        for (int i = 0; i < 8; i++) for (int l = 0; l < 8; l++) h[p][i][l] = IV[i] ^ (seed + 131u * (8 * p + l) + i);
        for (int i = 0; i < 16; i++) for (int l = 0; l < 8; l++) m[p][i][l] = seed * 2654435761u + 977u * (8 * p + l) + i;
    }
    for (uint64_t k = 0; k < iters; k++)
        for (int p = 0; p < 2; p++) { compress_avx2(h[p], m[p], 64, 0); m[p][0] ^= h[p][0]; m[p][5] ^= h[p][3]; m[p][10] ^= h[p][5]; m[p][15] ^= h[p][7]; }
    uint32_t acc = 0;
    for (int p = 0; p < 2; p++) for (int i = 0; i < 8; i++) for (int l = 0; l < 8; l++) acc ^= h[p][i][l];
    return acc;
}
// the same compression on one scalar lane: pins the vector paths to RFC 7693 through oracle/blake2s.h's known answers (tests/test_oracle_math.py)
uint32_t blake_scalar_lane0(uint64_t iters, uint32_t seed, int which_word) {
    typedef uint32_t lane1 __attribute__((vector_size(4)));
    lane1 h[8], m[16];
    for (int i = 0; i < 8; i++) h[i][0] = IV[i] ^ (seed + i);
    for (int i = 0; i < 16; i++) m[i][0] = seed * 2654435761u + i;
    const uint32_t t0 = 64, f0 = 0;
    for (uint64_t k = 0; k < iters; k++) {
        { COMPRESS_BODY(lane1) }
        m[0] ^= h[0]; m[5] ^= h[3]; m[10] ^= h[5]; m[15] ^= h[7];
    }
    return h[which_word][0];
}

// ---- M31 butterflies on 16 packed lanes ------------------------------------------------------------------------------------------------
// (a, b) -> (a + b t, a - b t) mod p = 2^31 - 1 on canonical lanes; the product as even/odd 32x32 -> 64-bit multiplies, reduced by
// x mod p = (x & p) + (x >> 31) twice folded — the shape of stwo's PackedM31 multiplication.
__attribute__((target("avx512f,avx512bw,avx512vl"))) uint32_t butterfly_loop_avx512(uint64_t iters, uint32_t seed) {
    const __m512i P = _mm512_set1_epi32(0x7fffffff);
    alignas(64) uint32_t init[16];
    __m512i a[4], b[4], t;
    for (int k = 0; k < 4; k++) {
        for (int l = 0; l < 16; l++) init[l] = (seed * 2654435761u + 7919u * l + 104729u * k) & 0x3fffffff;
        a[k] = _mm512_load_si512(init);
        for (int l = 0; l < 16; l++) init[l] = (seed * 40503u + 31337u * l + 65537u * k) & 0x3fffffff;
        b[k] = _mm512_load_si512(init);
    }
    for (int l = 0; l < 16; l++) init[l] = (seed + 3u * l + 5u) & 0x3fffffff;
    t = _mm512_load_si512(init);
    for (uint64_t it = 0; it < iters; it++)
        for (int k = 0; k < 4; k++) {               // four independent butterflies in flight
            const __m512i bt = m31_mul_avx512(b[k], t);
            __m512i s = _mm512_add_epi32(a[k], bt);
            s = _mm512_min_epu32(s, _mm512_sub_epi32(s, P));
            __m512i d = _mm512_sub_epi32(a[k], bt);
            d = _mm512_min_epu32(d, _mm512_add_epi32(d, P));
            a[k] = s; b[k] = d;
        }
    __m512i acc = _mm512_setzero_si512();
    for (int k = 0; k < 4; k++) acc = _mm512_xor_si512(acc, _mm512_xor_si512(a[k], b[k]));
    _mm512_store_si512(init, acc);
    uint32_t r = 0;
    for (int l = 0; l < 16; l++) r ^= init[l];
    return r;
}
__attribute__((target("avx2"), always_inline)) inline __m256i m31_mul_avx2(__m256i a, __m256i b) {
    const __m256i P = _mm256_set1_epi32(0x7fffffff), P64 = _mm256_set1_epi64x(0x7fffffff);
    const __m256i pe = _mm256_mul_epu32(a, b);
    const __m256i po = _mm256_mul_epu32(_mm256_srli_epi64(a, 32), _mm256_srli_epi64(b, 32));
    const __m256i lo = _mm256_or_si256(_mm256_and_si256(pe, P64), _mm256_slli_epi64(_mm256_and_si256(po, P64), 32));
    const __m256i hi = _mm256_or_si256(_mm256_srli_epi64(pe, 31), _mm256_slli_epi64(_mm256_srli_epi64(po, 31), 32));
    __m256i s = _mm256_add_epi32(lo, hi);
    return _mm256_min_epu32(s, _mm256_sub_epi32(s, P));
}
__attribute__((target("avx2"))) uint32_t butterfly_loop_avx2(uint64_t iters, uint32_t seed) {
    const __m256i P = _mm256_set1_epi32(0x7fffffff);
    alignas(32) uint32_t init[8];
    __m256i a[8], b[8], t;                              // 8 x 8 lanes = four 16-lane butterflies per iteration, like the other path
    for (int k = 0; k < 8; k++) {
        for (int l = 0; l < 8; l++) init[l] = (seed * 2654435761u + 7919u * l + 104729u * k) & 0x3fffffff;
        a[k] = _mm256_load_si256((const __m256i*)init);
        for (int l = 0; l < 8; l++) init[l] = (seed * 40503u + 31337u * l + 65537u * k) & 0x3fffffff;
        b[k] = _mm256_load_si256((const __m256i*)init);
    }
    for (int l = 0; l < 8; l++) init[l] = (seed + 3u * l + 5u) & 0x3fffffff;
    t = _mm256_load_si256((const __m256i*)init);
    for (uint64_t it = 0; it < iters; it++)
        for (int k = 0; k < 8; k++) {
            const __m256i bt = m31_mul_avx2(b[k], t);
            __m256i s = _mm256_add_epi32(a[k], bt);
            s = _mm256_min_epu32(s, _mm256_sub_epi32(s, P));
            __m256i d = _mm256_sub_epi32(a[k], bt);
            d = _mm256_min_epu32(d, _mm256_add_epi32(d, P));
            a[k] = s; b[k] = d;
        }
    __m256i acc = _mm256_setzero_si256();
    for (int k = 0; k < 8; k++) acc = _mm256_xor_si256(acc, _mm256_xor_si256(a[k], b[k]));
    _mm256_store_si256((__m256i*)init, acc);
    uint32_t r = 0;
    for (int l = 0; l < 8; l++) r ^= init[l];
    return r;
}
// scalar model of the butterfly loop's lane `lane` (k-th butterfly): checks the vector arithmetic against plain modular arithmetic
uint32_t butterfly_scalar(uint64_t iters, uint32_t seed, int k, int lane, bool want_b) {
    const uint64_t P = 0x7fffffffu;
    uint64_t a = (seed * 2654435761u + 7919u * (uint32_t)lane + 104729u * (uint32_t)k) & 0x3fffffff;
    uint64_t b = (seed * 40503u + 31337u * (uint32_t)lane + 65537u * (uint32_t)k) & 0x3fffffff;
    const uint64_t t = (seed + 3u * (uint32_t)lane + 5u) & 0x3fffffff;
    for (uint64_t it = 0; it < iters; it++) {
        const uint64_t bt = (b * t) % P;
        const uint64_t s = (a + bt) % P, d = (a + P - bt) % P;
        a = s; b = d;
    }
    return (uint32_t)(want_b ? b : a);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// runs `loop(iters, seed)` on `threads` threads for about `seconds`; returns calls of the loop body per second over all threads
template <class F>
double rate(F loop, int threads, double seconds, uint64_t chunk, std::atomic<uint32_t>& sink) {
    std::vector<std::thread> th;
    std::vector<uint64_t> done(threads, 0);
    std::atomic<bool> stop{false};
    std::atomic<int> started{0};
    const double t_begin = now_s();
    for (int i = 0; i < threads; i++)
        th.emplace_back([&, i]() {
            started++;
            while (started.load() < threads) std::this_thread::yield();
            uint32_t acc = 0;
            while (!stop.load(std::memory_order_relaxed)) { acc ^= loop(chunk, 0x9E3779B9u * (uint32_t)(i + 1) + (uint32_t)done[i]); done[i] += chunk; }
            sink ^= acc;
        });
    while (started.load() < threads) std::this_thread::yield();
    const double t0 = now_s();
    (void)t_begin;
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& t : th) t.join();
    const double dt = now_s() - t0;
    uint64_t total = 0;
    for (auto d : done) total += d;
    return (double)total / dt;
}

}  // namespace

extern "C" {

// out[0] = 16-lane Blake2s compressions per second (= 16 x the loop rate) over `threads` threads, out[1] = M31 butterflies per second
// (16 lanes x 4 per iteration), out[2] = 512 (AVX-512 path) / 256 (AVX2 path) / 0 (neither: rates are 0), out[3] = threads used.
int orc_simd_bound(int threads, double seconds_each, double out[4]) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    __builtin_cpu_init();
    const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    const bool avx2 = __builtin_cpu_supports("avx2");
    std::atomic<uint32_t> sink{0};
    out[0] = out[1] = 0.0; out[2] = avx512 ? 512.0 : avx2 ? 256.0 : 0.0; out[3] = (double)threads;
    if (avx512) {
        out[0] = 16.0 * rate(blake_loop_avx512, threads, seconds_each, 20000, sink);
        out[1] = 64.0 * rate(butterfly_loop_avx512, threads, seconds_each, 200000, sink);
    } else if (avx2) {
        out[0] = 16.0 * rate(blake_loop_avx2, threads, seconds_each, 20000, sink);
        out[1] = 64.0 * rate(butterfly_loop_avx2, threads, seconds_each, 200000, sink);
    }
    return (int)(sink.load() & 1u) * 0;      // the checksum keeps the loops alive
}

// Test hooks: the vector loops against their scalar models (tests/test_oracle_math.py). which: 0 = AVX-512 path, 1 = AVX2 path.
// Returns 0 when equal, 1 when different, -1 when the host lacks the instruction set.
int orc_simd_bound_selfcheck(int which) {
    __builtin_cpu_init();
    const uint64_t iters = 37;
    const uint32_t seed = 0xC0FFEEu;
    if (which == 0) {
        if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl"))) return -1;
        uint32_t r = 0;
        for (int k = 0; k < 4; k++) for (int l = 0; l < 16; l++) r ^= butterfly_scalar(iters, seed, k, l, false) ^ butterfly_scalar(iters, seed, k, l, true);
        if (r != butterfly_loop_avx512(iters, seed)) return 1;
        return blake_loop_avx512(iters, seed) == blake_loop_avx2(iters, seed) ? 0 : 1;      // the two vector widths agree lane for lane
    }
    if (!__builtin_cpu_supports("avx2")) return -1;
    uint32_t r = 0;
    for (int k = 0; k < 8; k++) for (int l = 0; l < 8; l++) r ^= butterfly_scalar(iters, seed, k, l, false) ^ butterfly_scalar(iters, seed, k, l, true);
    if (r != butterfly_loop_avx2(iters, seed)) return 1;
    return 0;
}
// One chained scalar compression sequence (the body the vector paths share), word `which_word` of the state after `iters` steps.
uint32_t orc_simd_bound_blake_scalar(uint64_t iters, uint32_t seed, int which_word) { return blake_scalar_lane0(iters, seed, which_word); }

}  // extern "C"
