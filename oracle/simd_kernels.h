// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). The vector kernels shared by oracle/simd_bound.cpp (the host-vector-unit bound) and
// oracle/simd_port.cpp (the SIMD mode of the CPU port): the Blake2s compression of RFC 7693 section 3.2 on a vector of u32 lanes (one lane =
// one independent message — the shape of stwo's `compress16`) and the packed M31 product. Every function carries its own target attribute;
// callers check the host's instruction sets at run time.
#pragma once
#include <cstdint>
#include <immintrin.h>

namespace {

typedef uint32_t v16u __attribute__((vector_size(64)));
typedef uint32_t v8u __attribute__((vector_size(32)));

const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
// ---- Blake2s on V = a vector of u32 lanes: the compression of RFC 7693 section 3.2, one lane = one independent message --------------------
#define ROTR(x, r) (((x) >> (r)) | ((x) << (32 - (r))))
#define G(a, b, c, d, x, y) \
    a = a + b + (x); d = ROTR(d ^ a, 16); c = c + d; b = ROTR(b ^ c, 12); \
    a = a + b + (y); d = ROTR(d ^ a, 8);  c = c + d; b = ROTR(b ^ c, 7);
#define ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                          \
    G(v0, v4, v8, v12, m[s0], m[s1]) G(v1, v5, v9, v13, m[s2], m[s3]) G(v2, v6, v10, v14, m[s4], m[s5]) G(v3, v7, v11, v15, m[s6], m[s7]) \
    G(v0, v5, v10, v15, m[s8], m[s9]) G(v1, v6, v11, v12, m[s10], m[s11]) G(v2, v7, v8, v13, m[s12], m[s13]) G(v3, v4, v9, v14, m[s14], m[s15])
// the sigma schedule is spelled out so that every message index is a compile-time constant and the 16 state words stay in registers
#define COMPRESS_BODY(V)                                                                                                     \
    V v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];                                \
    V v8 = (V){} + IV[0], v9 = (V){} + IV[1], v10 = (V){} + IV[2], v11 = (V){} + IV[3];                                       \
    V v12 = (V){} + (IV[4] ^ t0), v13 = (V){} + IV[5], v14 = (V){} + (IV[6] ^ f0), v15 = (V){} + IV[7];                        \
    ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)                                                              \
    ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)                                                              \
    ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)                                                              \
    ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)                                                              \
    ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)                                                              \
    ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)                                                              \
    ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)                                                              \
    ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)                                                              \
    ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)                                                              \
    ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)                                                              \
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;

__attribute__((target("avx512f,avx512bw,avx512vl"), always_inline)) inline void compress_avx512(v16u h[8], const v16u m[16], uint32_t t0, uint32_t f0) { COMPRESS_BODY(v16u) }
__attribute__((target("avx2"), always_inline)) inline void compress_avx2(v8u h[8], const v8u m[16], uint32_t t0, uint32_t f0) { COMPRESS_BODY(v8u) }

__attribute__((target("avx512f,avx512bw,avx512vl"), always_inline)) inline __m512i m31_mul_avx512(__m512i a, __m512i b) {
    const __m512i P = _mm512_set1_epi32(0x7fffffff), P64 = _mm512_set1_epi64(0x7fffffff);
    const __m512i pe = _mm512_mul_epu32(a, b);                                             // even lanes: 62-bit products
    const __m512i po = _mm512_mul_epu32(_mm512_srli_epi64(a, 32), _mm512_srli_epi64(b, 32));
    // x = lo + 2^31 hi with lo, hi < 2^31, and 2^31 = 1 (mod p): x = lo + hi (mod p); even products in the low, odd ones in the high halves
    const __m512i lo = _mm512_or_si512(_mm512_and_si512(pe, P64), _mm512_slli_epi64(_mm512_and_si512(po, P64), 32));
    const __m512i hi = _mm512_or_si512(_mm512_srli_epi64(pe, 31), _mm512_slli_epi64(_mm512_srli_epi64(po, 31), 32));
    __m512i s = _mm512_add_epi32(lo, hi);                                                 // < 2^32
    s = _mm512_min_epu32(s, _mm512_sub_epi32(s, P));                                       // one conditional subtraction
    return s;
}

}  // namespace
