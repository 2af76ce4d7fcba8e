// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). PARITY UNPINNED (the same restatement as merkle.h / circle.h, other instructions).
//
// The SIMD mode of the CPU port (orc_set_simd(1)): bench.py's `cpu_baseline` stand-in for the reference's parallel CPU path — stwo's
// SimdBackend + rayon (`cargo build --features parallel --release`, README.md:23-36; Cargo.toml:50-53; the time it prints:
// bin/brainfuck_prover.rs:137-139), which cannot be built in this image. In this mode the two loops that dominate a SimdBackend-shaped
// prover run on the host's AVX-512 units, 16 u32 lanes per instruction like stwo's PackedM31 / compress16:
//
//   merkle_layer_blake2s   MerkleProver::commit's layer loop (merkle.h; mod.rs:500,583,723): 16 nodes per compression call, lanes = nodes;
//   circle_fft / _ifft     CpuBackend::evaluate / ::interpolate (circle.h; mod.rs:497,550-562,690-702): 16 butterflies per instruction, the
//                          four smallest strides through in-register permutations, and the large strides THREE layers at a time so that a
//                          column makes a third of the passes over memory.
//
// Values are canonical after every operation, exactly as in the scalar code, so every intermediate word — and the proof — is identical to the
// scalar port's (tests/test_oracle_prove.py checks SIMD == scalar on the reference's programs). The scalar path stays the checker's default.
#include "simd_port.h"
#include "simd_kernels.h"
#include <atomic>
#include <cstring>
#include <vector>

#define AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq")))

namespace orc { namespace simd {

static std::atomic<int> g_enabled{0};
bool available() {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
}
void set_enabled(bool on) { g_enabled.store(on && available() ? 1 : 0); }
bool enabled() { return g_enabled.load(std::memory_order_relaxed) != 0; }

// ---- Merkle layer -------------------------------------------------------------------------------------------------------------------
// Blake2sMerkleHasher::hash_node in the stwo-compress convention (merkle.h): state = 0; compress(state, left || right) if there is a deeper
// layer; compress(state, 16 column words, zero padded) per chunk. Nodes [0, n_nodes), n_nodes a multiple of 16.
AVX512 static void merkle_block16(const uint32_t* prev32, const uint32_t* const* cols, size_t n_cols, size_t i, uint32_t* out32) {
    v16u h[8], m[16];
    for (int w = 0; w < 8; w++) h[w] = (v16u){};
    if (prev32) {
        const __m512i lane16 = _mm512_setr_epi32(0, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240);
        for (int k = 0; k < 16; k++) m[k] = (v16u)_mm512_i32gather_epi32(lane16, (const int*)(prev32 + 16 * i + k), 4);     // word k of (left || right) of node i + lane
        compress_avx512(h, m, 0, 0);
    }
    for (size_t o = 0; o < n_cols; o += 16) {
        for (size_t k = 0; k < 16; k++) m[k] = o + k < n_cols ? (v16u)_mm512_loadu_si512((const void*)(cols[o + k] + i)) : (v16u){};
        compress_avx512(h, m, 0, 0);
    }
    const __m512i lane8 = _mm512_setr_epi32(0, 8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120);
    for (int w = 0; w < 8; w++) _mm512_i32scatter_epi32((int*)(out32 + 8 * i + w), lane8, (__m512i)h[w], 4);
}
void merkle_layer_blake2s(const uint8_t* prev, const uint32_t* const* cols, size_t n_cols, size_t n_nodes, uint8_t* out) {
    const uint32_t* prev32 = reinterpret_cast<const uint32_t*>(prev);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(out);
    const long blocks = (long)(n_nodes / 16);
#pragma omp parallel for schedule(static) if (blocks >= 64)
    for (long b = 0; b < blocks; b++) merkle_block16(prev32, cols, n_cols, (size_t)b * 16, out32);
}

// ---- packed M31 ---------------------------------------------------------------------------------------------------------------------
AVX512 static inline __m512i add_m(__m512i a, __m512i b) { const __m512i P = _mm512_set1_epi32(0x7fffffff); __m512i s = _mm512_add_epi32(a, b); return _mm512_min_epu32(s, _mm512_sub_epi32(s, P)); }
AVX512 static inline __m512i sub_m(__m512i a, __m512i b) { const __m512i P = _mm512_set1_epi32(0x7fffffff); __m512i d = _mm512_sub_epi32(a, b); return _mm512_min_epu32(d, _mm512_add_epi32(d, P)); }
AVX512 static inline __m512i neg_m(__m512i a) { const __m512i P = _mm512_set1_epi32(0x7fffffff); __m512i d = _mm512_sub_epi32(P, a); return _mm512_min_epu32(d, _mm512_sub_epi32(d, P)); }   // -0 = 0
// forward: (a, b) -> (a + b t, a - b t); inverse: (a, b) -> (a + b, (a - b) t)
AVX512 static inline void bf(__m512i& a, __m512i& b, __m512i t) { const __m512i bt = m31_mul_avx512(b, t); const __m512i s = add_m(a, bt); b = sub_m(a, bt); a = s; }
AVX512 static inline void ibf(__m512i& a, __m512i& b, __m512i t) { const __m512i s = add_m(a, b); b = m31_mul_avx512(sub_m(a, b), t); a = s; }

// In-register layers of a 32-element block (two vectors): stride S = 1 << ls in {1, 2, 4, 8}. lo/hi gather the pair-firsts / pair-seconds.
struct Perm { __m512i lo, hi, a, b, tw; };
AVX512 static Perm make_perm(int ls, int tw_shift) {
    alignas(64) int lo[16], hi[16], pa[16], pb[16], tw[16];
    const int S = 1 << ls;
    for (int k = 0; k < 16; k++) {
        const int e = ((k >> ls) << (ls + 1)) | (k & (S - 1));
        lo[k] = e; hi[k] = e | S;
        tw[k] = e >> tw_shift;
    }
    for (int e = 0; e < 32; e++) {
        const int k = ((e >> (ls + 1)) << ls) | (e & (S - 1));
        (e < 16 ? pa : pb)[e & 15] = k | ((e & S) ? 16 : 0);
    }
    Perm p;
    p.lo = _mm512_load_si512(lo); p.hi = _mm512_load_si512(hi); p.a = _mm512_load_si512(pa); p.b = _mm512_load_si512(pb); p.tw = _mm512_load_si512(tw);
    return p;
}
// twiddle vector of the circle layer for pairs h0 .. h0 + 15 (h0 a multiple of 16): chunks [x, y] of the first line layer -> [y, -y, -x, x]
AVX512 static inline __m512i circle_tw16(const uint32_t* l0, size_t h0) {
    const __m512i idx = _mm512_setr_epi32(1, 1, 0, 0, 3, 3, 2, 2, 5, 5, 4, 4, 7, 7, 6, 6);
    const __m512i t = _mm512_i32gather_epi32(idx, (const int*)(l0 + h0 / 2), 4);
    return _mm512_mask_mov_epi32(t, (__mmask16)0x6666, neg_m(t));
}
template <bool INV>
AVX512 static inline void small_layer(__m512i& A, __m512i& B, const Perm& p, __m512i t) {
    __m512i lo = _mm512_permutex2var_epi32(A, p.lo, B), hi = _mm512_permutex2var_epi32(A, p.hi, B);
    if (INV) ibf(lo, hi, t); else bf(lo, hi, t);
    A = _mm512_permutex2var_epi32(lo, p.a, hi); B = _mm512_permutex2var_epi32(lo, p.b, hi);
}
// Layers of strides 1 (circle), 2, 4, 8 on every 32-element block of [v + lo, v + hi): tw1/2/3 = line twiddle layers 0/1/2, l0 = layer 0 too.
template <bool INV>
AVX512 static void small_layers(uint32_t* v, size_t lo, size_t hi, const uint32_t* l0, const uint32_t* l1, const uint32_t* l2, unsigned half_log) {
    const Perm p0 = make_perm(0, 0), p1 = make_perm(1, 2), p2 = make_perm(2, 3), p3 = make_perm(3, 4);
    for (size_t base = lo; base < hi; base += 32) {
        __m512i A = _mm512_loadu_si512((const void*)(v + base)), B = _mm512_loadu_si512((const void*)(v + base + 16));
        const __m512i tc = circle_tw16(l0, base / 2);
        // line layer j has stride 2^(j+1) and twiddle index (element >> (j+2)); a column shorter than a layer has no such layer
        const __m512i t1 = _mm512_i32gather_epi32(p1.tw, (const int*)(l0 + (base >> 2)), 4);
        const __m512i t2 = half_log >= 2 ? _mm512_i32gather_epi32(p2.tw, (const int*)(l1 + (base >> 3)), 4) : t1;
        const __m512i t3 = half_log >= 3 ? _mm512_i32gather_epi32(p3.tw, (const int*)(l2 + (base >> 4)), 4) : t1;
        if (INV) {
            small_layer<true>(A, B, p0, tc);
            small_layer<true>(A, B, p1, t1);
            if (half_log >= 2) small_layer<true>(A, B, p2, t2);
            if (half_log >= 3) small_layer<true>(A, B, p3, t3);
        } else {
            if (half_log >= 3) small_layer<false>(A, B, p3, t3);
            if (half_log >= 2) small_layer<false>(A, B, p2, t2);
            small_layer<false>(A, B, p1, t1);
            small_layer<false>(A, B, p0, tc);
        }
        _mm512_storeu_si512((void*)(v + base), A); _mm512_storeu_si512((void*)(v + base + 16), B);
    }
}

// Line layers with stride >= 16, L consecutive layers per pass over [v + lo, v + hi) (a multiple of the largest block of the group):
// layer j (stride 2^(j+1)) pairs idx0 = (h << (j+2)) + l with idx0 + 2^(j+1), twiddle tw_j[h]. The 2^L vectors of one butterfly network are
// v[blk + r * s_lo + o .. + 16) for r < 2^L, with s_lo the smallest stride of the group.
template <bool INV, int L>
AVX512 static void big_layers(uint32_t* v, size_t lo, size_t hi, const uint32_t* const* tw, unsigned j_lo) {
    // layers j_lo .. j_lo + L - 1; strides 2^(j+1)
    const size_t s_lo = size_t(1) << (j_lo + 1), blk_size = s_lo << L;
    constexpr int R = 1 << L;
    for (size_t blk = lo; blk < hi; blk += blk_size) {
        __m512i t[L][R / 2 > 0 ? R / 2 : 1];
        // twiddles of this block: layer j = j_lo + q has 2^(L-1-q) distinct blocks of size 2^(j+2) inside blk
        for (int q = 0; q < L; q++) {
            const unsigned j = j_lo + q;
            const int cnt = 1 << (L - 1 - q);
            for (int c = 0; c < cnt; c++) t[q][c] = _mm512_set1_epi32((int)tw[j][(blk >> (j + 2)) + c]);
        }
        for (size_t o = 0; o < s_lo; o += 16) {
            __m512i x[R];
            for (int r = 0; r < R; r++) x[r] = _mm512_loadu_si512((const void*)(v + blk + (size_t)r * s_lo + o));
            if (INV) {
                for (int q = 0; q < L; q++) {                       // ascending strides
                    const int half = 1 << q;
                    for (int r = 0; r < R; r++) if (!(r & half)) ibf(x[r], x[r | half], t[q][r >> (q + 1)]);
                }
            } else {
                for (int q = L - 1; q >= 0; q--) {                  // descending strides
                    const int half = 1 << q;
                    for (int r = 0; r < R; r++) if (!(r & half)) bf(x[r], x[r | half], t[q][r >> (q + 1)]);
                }
            }
            for (int r = 0; r < R; r++) _mm512_storeu_si512((void*)(v + blk + (size_t)r * s_lo + o), x[r]);
        }
    }
}

// line twiddle layer j of a domain whose half coset has log size half_log: the slice [len - 2 * 2^(half_log-1-j), len - 2^(half_log-1-j))
static inline const uint32_t* line_layer(const uint32_t* buf, size_t len, unsigned half_log, unsigned j) { return buf + (len - (size_t(2) << (half_log - 1 - j))); }

template <bool INV>
AVX512 static void transform(uint32_t* v, unsigned log_size, const uint32_t* twbuf, size_t tw_len) {
    const unsigned half_log = log_size - 1;            // line layers j = 0 .. half_log - 1 (strides 2 .. 2^half_log), plus the circle layer (stride 1)
    const size_t n = size_t(1) << log_size;
    std::vector<const uint32_t*> tw(half_log);
    for (unsigned j = 0; j < half_log; j++) tw[j] = line_layer(twbuf, tw_len, half_log, j);
    const uint32_t* l1 = half_log >= 2 ? tw[1] : tw[0];
    const uint32_t* l2 = half_log >= 3 ? tw[2] : tw[0];
    // big layers: j = 3 .. half_log - 1 in groups of 3 (then 2, then 1)
    std::vector<std::pair<unsigned, int>> groups;
    for (unsigned j = 3; j < half_log;) { const int L = half_log - j >= 3 ? 3 : (int)(half_log - j); groups.push_back({j, L}); j += L; }
    auto run_group = [&](const std::pair<unsigned, int>& g) {
        if (g.second == 3) big_layers<INV, 3>(v, 0, n, tw.data(), g.first);
        else if (g.second == 2) big_layers<INV, 2>(v, 0, n, tw.data(), g.first);
        else big_layers<INV, 1>(v, 0, n, tw.data(), g.first);
    };
    if (INV) {
        small_layers<true>(v, 0, n, tw[0], l1, l2, half_log);
        for (size_t k = 0; k < groups.size(); k++) run_group(groups[k]);
    } else {
        for (size_t k = groups.size(); k-- > 0;) run_group(groups[k]);
        small_layers<false>(v, 0, n, tw[0], l1, l2, half_log);
    }
}

AVX512 static void scale(uint32_t* v, size_t n, uint32_t c) {
    const __m512i cv = _mm512_set1_epi32((int)c);
    for (size_t i = 0; i < n; i += 16) _mm512_storeu_si512((void*)(v + i), m31_mul_avx512(_mm512_loadu_si512((const void*)(v + i)), cv));
}

void circle_fft(uint32_t* values, unsigned log_size, const uint32_t* twiddles, size_t tw_len) { transform<false>(values, log_size, twiddles, tw_len); }
void circle_ifft(uint32_t* values, unsigned log_size, const uint32_t* itwiddles, size_t tw_len, uint32_t n_inverse) {
    transform<true>(values, log_size, itwiddles, tw_len);
    scale(values, size_t(1) << log_size, n_inverse);
}

// ---- FRI quotients (compute_fri_quotients / accumulate_row_quotients, prover.h) on 16 rows per instruction -----------------------------------
struct C16 { __m512i a, b; };                 // packed CM31
struct Q16 { C16 a, b; };                     // packed QM31 = CM31[u] / (u^2 - (2 + i))
AVX512 static inline __m512i bc(uint32_t x) { return _mm512_set1_epi32((int)x); }
AVX512 static inline C16 c_add(C16 x, C16 y) { return {add_m(x.a, y.a), add_m(x.b, y.b)}; }
AVX512 static inline C16 c_sub(C16 x, C16 y) { return {sub_m(x.a, y.a), sub_m(x.b, y.b)}; }
AVX512 static inline C16 c_mul(C16 x, C16 y) { return {sub_m(m31_mul_avx512(x.a, y.a), m31_mul_avx512(x.b, y.b)), add_m(m31_mul_avx512(x.a, y.b), m31_mul_avx512(x.b, y.a))}; }
AVX512 static inline C16 c_mul_R(C16 x) { return {sub_m(add_m(x.a, x.a), x.b), add_m(x.a, add_m(x.b, x.b))}; }     // (a + bi)(2 + i) = (2a - b) + (a + 2b) i
AVX512 static inline Q16 q_add(Q16 x, Q16 y) { return {c_add(x.a, y.a), c_add(x.b, y.b)}; }
AVX512 static inline Q16 q_sub(Q16 x, Q16 y) { return {c_sub(x.a, y.a), c_sub(x.b, y.b)}; }
AVX512 static inline Q16 q_mul(Q16 x, Q16 y) { return {c_add(c_mul(x.a, y.a), c_mul_R(c_mul(x.b, y.b))), c_add(c_mul(x.a, y.b), c_mul(x.b, y.a))}; }
AVX512 static inline Q16 q_mul_m(Q16 x, __m512i y) { return {{m31_mul_avx512(x.a.a, y), m31_mul_avx512(x.a.b, y)}, {m31_mul_avx512(x.b.a, y), m31_mul_avx512(x.b.b, y)}}; }
AVX512 static inline Q16 q_mul_c(Q16 x, C16 y) { return {c_mul(x.a, y), c_mul(x.b, y)}; }
AVX512 static inline Q16 q_bc(const uint32_t w[4]) { return {{bc(w[0]), bc(w[1])}, {bc(w[2]), bc(w[3])}}; }
AVX512 static inline Q16 q_zero() { const __m512i z = _mm512_setzero_si512(); return {{z, z}, {z, z}}; }

AVX512 void quotient_rows(const uint32_t* const* cols, const uint32_t* ys, const QuotientBatch* batches, size_t n_batches, size_t r0, size_t r1, uint32_t* const out[4]) {
    {
        for (size_t row = r0; row < r1; row += 16) {
            const __m512i y = _mm512_loadu_si512((const void*)(ys + (row - r0)));
            Q16 acc = q_zero();
            for (size_t bi = 0; bi < n_batches; bi++) {
                const QuotientBatch& b = batches[bi];
                Q16 num = q_zero();
                for (size_t k = 0; k < b.n_cols; k++) {
                    const uint32_t* lc = b.line_coeffs + 12 * k;                      // lc[0] = alpha a, lc[1] = alpha b, lc[2] = alpha c
                    const __m512i v = _mm512_loadu_si512((const void*)(cols[b.col_index[k]] + row));
                    const Q16 value = q_mul_m(q_bc(lc + 8), v);
                    const Q16 linear = q_add(q_mul_m(q_bc(lc), y), q_bc(lc + 4));
                    num = q_add(num, q_sub(value, linear));
                }
                const C16 deninv = {_mm512_loadu_si512((const void*)(b.deninv_a + (row - r0))), _mm512_loadu_si512((const void*)(b.deninv_b + (row - r0)))};
                acc = q_add(q_mul(acc, q_bc(b.batch_coeff)), q_mul_c(num, deninv));
            }
            _mm512_storeu_si512((void*)(out[0] + row), acc.a.a); _mm512_storeu_si512((void*)(out[1] + row), acc.a.b);
            _mm512_storeu_si512((void*)(out[2] + row), acc.b.a); _mm512_storeu_si512((void*)(out[3] + row), acc.b.b);
        }
    }
}

}}  // namespace orc::simd
