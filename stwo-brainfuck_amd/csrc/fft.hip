// Circle FFT / iFFT over M31 for gfx950 (MI355X) — SURVEY.md §8 rows a1 (twiddles), a2 (interpolate), a3 (evaluate / LDE).
// Replaces stwo `PolyOps::{precompute_twiddles, interpolate_columns, evaluate_polynomials}` as reached from
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:480-484 (twiddles), :497,550-562,690-702 (extend_evals = interpolate),
// :500,583,723 (commit = LDE evaluate).
//
// Layout: a column is a contiguous u32[2^log] in HBM, bit-reversed circle-domain order (the order the reference stores).
// The butterfly network is identical to stwo's CpuBackend (layer 0 = circle layer on adjacent pairs, layer i>=1 = line layer
// at distance 2^i), so results are value-identical. A transform is executed as 1..3 passes; each pass stages a 4096-element
// tile in LDS, runs up to 12 (contiguous tile) or 7 (128-byte-chunk strided tile) layers there, and touches HBM once for read and
// once for write with 16-byte per-lane accesses. Twiddles of the tile are staged in LDS once and reused for every column of
// the batch (grid.y walks column groups), so twiddle traffic is amortised over the batch.
//
// "line mode" (circle = 0) runs the same network without the circle layer: this is exactly what the circle FFT of a column
// whose every value is replicated 16x (the reference broadcasts each table row into 16 SIMD lanes, memory/table.rs:95-104)
// reduces to, on the 16x smaller row-granular column. See DESIGN.md "replicated columns".
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <algorithm>
#include <vector>
#include <mutex>
#include <set>
#include <string>

namespace bf {

static constexpr int TILE_LOG = 12;          // 4096 elements = 16 KiB of LDS
static constexpr int TILE = 1 << TILE_LOG;
static constexpr int CHUNK_LOG = 5;          // 32 x u32 = 128 B contiguous per strided row
static constexpr int STRIDED_K = TILE_LOG - CHUNK_LOG;  // 7 layers per strided pass
static constexpr int FFT_THREADS = 256;
// PassArgs::scale of an inverse transform is 2^-n = a power of two (m31.h: m_inv_pow2): its exponent, for m_mul_pow2 (uniform: scalar unit)
__device__ __forceinline__ u32 scale_shift(u32 scale) { return (u32)__builtin_ctz(scale); }

// ---------------------------------------------------------------------------------------------------------------------
// Twiddle generation (a1). Layered buffer exactly as stwo's slow_precompute_twiddles(Coset::half_odds(R)):
// layer j (j < R) holds the bit-reversed x-coordinates of the first half of the coset doubled j times; last entry = 1.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void point_of_index(u32 idx, const uint2* __restrict__ tlo, const uint2* __restrict__ thi, u32& x, u32& y) {
    uint2 a = tlo[idx & 0xffffu], b = thi[(idx >> 16) & 0x7fffu];
    x = m_sub(m_mul(a.x, b.x), m_mul(a.y, b.y));
    y = m_add(m_mul(a.x, b.y), m_mul(a.y, b.x));
}

__global__ void k_gen_twiddles(u32* __restrict__ tw, u32* __restrict__ itw, u32 R, const uint2* __restrict__ tlo, const uint2* __restrict__ thi) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = 1u << R;
    if (g >= total) return;
    if (g == total - 1) { tw[g] = 1; itw[g] = 1; return; }
    // layer j starts at offset 2^R - 2^(R-j): j = number of leading ones of g within R bits
    u32 j = __clz(~(g << (32 - R)));
    u32 off = total - (1u << (R - j));
    u32 k = g - off;
    u32 bits = R - 1 - j;
    u32 init = (1u << (31 - R - 2)) << j, step = (1u << (31 - R)) << j;
    u32 idx = (init + step * bit_rev(k, bits)) & 0x7fffffffu;
    u32 x, y;
    point_of_index(idx, tlo, thi, x, y);
    tw[g] = x;
    itw[g] = m_inv(x);
}

// ---------------------------------------------------------------------------------------------------------------------
// Preprocessed columns in closed form: interpolate(gen_is_first(n)) for every n in one launch (mod.rs:495-499: gen_is_first::<SimdBackend>
// then tree_builder.extend_evals = interpolate). IsFirst(n) is the indicator of cell 0; under the inverse butterflies
// (v0, v1) -> (v0 + v1, (v0 - v1) t) a one-hot vector stays a tensor product: after layer i cell (1 << i) | l holds cell l times t_i,
// t_i = the layer's inverse twiddle of butterfly block 0. Hence coefficient j = 2^-n * prod over the set bits i of j of t_i — the values the
// transform of the one-hot column produces, with no column traffic and no 19 x (1 + 3) launches.
// t_0 = the circle layer's twiddle of pair 0 (y of the first entry pair of the first line table), t_i = entry 0 of line layer i.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_is_first_coeffs(IsFirstCols a, const u32* __restrict__ itw, u32 tw_total) {
    // one lane writes 16 consecutive coefficients; lanes are flattened over the columns in ascending size
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x + (1u << (a.log_min - 4));
    if (u >= (1u << (a.log_max - 3))) return;
    const u32 n = 35u - __clz(u);                     // column IsFirst(n): u in [2^(n-4), 2^(n-3))
    u32* __restrict__ dst = a.ptr[n - a.log_min];
    if (!dst) return;                                 // another rank of the shard group owns this column
    const u32 jhi = u - (1u << (n - 4));              // coefficient index >> 4
    auto t = [&](u32 i) -> u32 { return i == 0 ? itw[tw_total - (1u << (n - 1)) + 1] : itw[tw_total - (1u << (n - i))]; };
    u32 v[16];
    v[0] = m_inv_pow2(n);
    for (u32 i = 4; i < n; i++) if ((jhi >> (i - 4)) & 1) v[0] = m_mul(v[0], t(i));
#pragma unroll
    for (u32 b = 0; b < 4; b++) {
        const u32 tb = t(b);
#pragma unroll
        for (u32 e = 0; e < (1u << b); e++) v[e + (1u << b)] = m_mul(v[e], tb);
    }
    uint4* o = reinterpret_cast<uint4*>(dst + 16 * (size_t)jhi);
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
void is_first_coeffs(hipStream_t stream, const IsFirstCols& a, const u32* itw, u32 tw_root_log) {
    if (a.log_min < 4 || a.log_max < a.log_min || a.log_max - a.log_min >= 28 || a.log_max > tw_root_log + 1) throw std::runtime_error("is_first_coeffs: unsupported sizes");
    const u32 lanes = (1u << (a.log_max - 3)) - (1u << (a.log_min - 4));
    hipLaunchKernelGGL(k_is_first_coeffs, dim3((lanes + 255) / 256), dim3(256), 0, stream, a, itw, 1u << tw_root_log);
}

// ---------------------------------------------------------------------------------------------------------------------
// One pass = layers [lo, lo+k) of a size-2^log transform, for the columns cols[by * cpb .. +cpb) of one GROUP (columns of one size and
// storage). A launch covers several groups (PassArgs table in HBM, kernels.h): workgroup b belongs to the group g with
// groups[g].block0 <= b < groups[g + 1].block0 and is its (tile, column block) = ((b - block0) % grid_x, (b - block0) / grid_x). The 13
// components of a proof have ~10 distinct sizes: one launch per pass and kernel kind instead of one per size.
// ---------------------------------------------------------------------------------------------------------------------
// kernel kinds of a launch. PassArgs::kind != 0 marks a group that rides in ANOTHER kind's launch: the single-pass transforms of 2^6..2^11 cells (K_PASS) and the tiny
// ones (K_TINY) of a plan join its contiguous-tile launch (r06: a proof issued ~24 separate 5-18 us launches for them, each a latency chain on an otherwise idle GPU)
enum { K_TILE12 = 0, K_STRIDED5 = 1, K_STRIDED6 = 2, K_PASS = 3, K_TINY = 4, K_STRIDED_K8 = 5, K_STRIDED_K9 = 6, K_STRIDED_K10 = 7, K_KINDS = 8 };
struct BlockOfGroup { u32 g, tile, by; };
template <bool INV> __device__ __forceinline__ void fft_tiny_body(const PassArgs& a, const BlockOfGroup& bg);      // defined with k_fft_tiny below
__device__ __forceinline__ BlockOfGroup find_group(const PassArgs* __restrict__ groups, u32 ngroups) {
    u32 g = 0;
    while (g + 1 < ngroups && groups[g + 1].block0 <= blockIdx.x) g++;      // uniform: scalar loads
    const u32 local = blockIdx.x - groups[g].block0, gx = groups[g].grid_x;
    return {g, local % gx, local / gx};
}
__device__ __forceinline__ const u32* layer_table(const PassArgs& a, u32 layer) {
    // line layer `layer` (>= 1 in circle mode, >= 0 in line mode): len = 2^(log-1-layer), table at tw_total - 2*len.
    // In line mode local layer j plays the role of circle-mode layer j+1 of a transform twice the size.
    u32 eff = a.circle ? layer : layer + 1;
    u32 lg = a.circle ? a.log : a.log + 1;
    u32 len = 1u << (lg - 1 - eff);
    return a.tw + (a.tw_total - 2 * len);
}

// s_val, s_tw: TILE words each (s_tw: per-layer twiddle segments of this tile, packed)
template <bool INV>
__device__ __forceinline__ void fft_pass_body(const PassArgs& a, const BlockOfGroup& bg, u32* __restrict__ s_val, u32* __restrict__ s_tw) {
    const u32 t = threadIdx.x;
    const u32 lo = a.lo, k = a.k;
    const u32 c = lo == 0 ? 0 : CHUNK_LOG;
    const u32 tile_log = lo == 0 ? a.tile_log : k + c;
    const u32 tile_n = 1u << tile_log;
    const u32 kt = tile_log - c;   // twiddle span: local layer j needs 2^(kt-1-j) entries, staged at offset 2^kt - 2^(kt-j)
    const u32 tile = bg.tile;
    // element (m, l) of the tile -> global index
    // contiguous: idx = tile * tile_n + m
    // strided   : idx = (H << (lo+k)) | (m << lo) | (Lhi << c) | l   with tile = H * 2^(lo-c) + Lhi
    const u32 n_lhi_log = lo - c;  // only used when lo > 0
    const u32 H = lo ? (tile >> n_lhi_log) : 0, Lhi = lo ? (tile & ((1u << n_lhi_log) - 1)) : 0;
    const u32 base = lo ? ((H << (lo + k)) | (Lhi << c)) : (tile << tile_log);

    // ---- stage twiddles: for local layer j (global lo+j) the tile needs 2^(k-1-j) entries -----------------------------
    // s_tw offset of local layer j: sum_{q<j} 2^(k-1-q) = 2^k - 2^(k-j)
    for (u32 e = t; e < (1u << kt) - (1u << (kt - k)); e += FFT_THREADS) {
        // find layer j with offset <= e
        u32 j = __clz(~((e) << (32 - kt)));  // leading ones of e in kt bits
        u32 off = (1u << kt) - (1u << (kt - j));
        u32 q = e - off;                     // q < 2^(kt-1-j): index of the h-block inside the tile
        u32 gl = lo + j;
        u32 hbase = lo ? (H << (kt - 1 - j)) : (tile << (kt - 1 - j));
        u32 v;
        if (a.circle && gl == 0) {
            // circle layer: pair h uses [y, -y, -x, x][h & 3] of chunk (h >> 2) of the first line table
            const u32* l0 = layer_table(a, 1);
            u32 h = hbase + q;
            u32 x = l0[(h >> 2) * 2], y = l0[(h >> 2) * 2 + 1];
            u32 sel = h & 3;
            v = sel == 0 ? y : sel == 1 ? m_neg(y) : sel == 2 ? m_neg(x) : x;
        } else {
            v = layer_table(a, gl)[hbase + q];
        }
        s_tw[e] = v;
    }

    const u32 col0 = bg.by * a.cols_per_block;
    const u32 col1 = min(a.ncols, col0 + a.cols_per_block);
    for (u32 col = col0; col < col1; col++) {
        g_cu32p src = as_global(a.src[col]);
        g_u32p dst = as_global(a.dst[col]);
        __syncthreads();  // s_tw ready / previous column's stores done reading s_val
        // ---- load tile (16 B per lane) -----------------------------------------------------------------------------
        for (u32 e4 = t * 4; e4 < tile_n; e4 += 4 * FFT_THREADS) {
            u32 gidx = lo ? (base | ((e4 >> c) << lo) | (e4 & ((1u << c) - 1))) : (base + e4);
            uint4 v = ld16(src + (gidx & a.src_mask));
            *reinterpret_cast<uint4*>(&s_val[e4]) = v;
        }
        __syncthreads();
        // ---- butterflies -------------------------------------------------------------------------------------------
        const u32 nb = tile_n >> 1;
        if (INV) {
            for (u32 j = 0; j < k; j++) {
                u32 dist = 1u << (j + c);
                u32 toff = (1u << kt) - (1u << (kt - j));
                for (u32 b = t; b < nb; b += FFT_THREADS) {
                    u32 i0 = ((b >> (j + c)) << (j + c + 1)) | (b & (dist - 1));
                    u32 tw = s_tw[toff + (b >> (j + c))];
                    u32 v0 = s_val[i0], v1 = s_val[i0 + dist];
                    s_val[i0] = m_add(v0, v1);
                    s_val[i0 + dist] = m_mul(m_sub(v0, v1), tw);
                }
                __syncthreads();
            }
        } else {
            for (int j = (int)k - 1; j >= 0; j--) {
                u32 dist = 1u << (j + c);
                u32 toff = (1u << kt) - (1u << (kt - j));
                for (u32 b = t; b < nb; b += FFT_THREADS) {
                    u32 i0 = ((b >> (j + c)) << (j + c + 1)) | (b & (dist - 1));
                    u32 tw = s_tw[toff + (b >> (j + c))];
                    u32 v0 = s_val[i0], v1 = m_mul(s_val[i0 + dist], tw);
                    s_val[i0] = m_add(v0, v1);
                    s_val[i0 + dist] = m_sub(v0, v1);
                }
                __syncthreads();
            }
        }
        // ---- store tile --------------------------------------------------------------------------------------------
        for (u32 e4 = t * 4; e4 < tile_n; e4 += 4 * FFT_THREADS) {
            u32 gidx = lo ? (base | ((e4 >> c) << lo) | (e4 & ((1u << c) - 1))) : (base + e4);
            uint4 v = *reinterpret_cast<uint4*>(&s_val[e4]);
            if (INV && a.scale != 1) { const u32 sk = scale_shift(a.scale); v.x = m_mul_pow2(v.x, sk); v.y = m_mul_pow2(v.y, sk); v.z = m_mul_pow2(v.z, sk); v.w = m_mul_pow2(v.w, sk); }
            st16(dst + gidx, v);
        }
    }
}

template <bool INV>
__global__ void __launch_bounds__(FFT_THREADS) k_fft_pass(const PassArgs* __restrict__ groups, u32 ngroups) {
    const BlockOfGroup bg = find_group(groups, ngroups);
    const PassArgs a = groups[bg.g];
    __shared__ u32 s_val[TILE];
    __shared__ u32 s_tw[TILE];
    fft_pass_body<INV>(a, bg, s_val, s_tw);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast path (transforms of >= 2^12 cells): register-radix butterflies, one LDS round trip per 4 layers.
//   k_fft_tile12 : contiguous 4096-cell tile, layers [0, k), 6 <= k <= 12. 256 lanes x 16 cells. Layers 0,1 and 10,11 are done
//                  in registers on the 16-byte global accesses; layers 2..5 and 6..9 are two radix-16 rounds through LDS.
//   k_fft_strided7: 128 rows x 32 or 64 cells (128- or 256-byte rows at stride 2^lo), layers [lo, lo+7). 32 cells per lane. Layers lo..lo+2 in
//                  registers on the 16-byte accesses, layers lo+3..lo+6 as one radix-16 round with 4-byte coalesced row accesses.
// LDS index padding p(i) = i + 4*(i >> 6) keeps the stride-4 round conflict-free and the 16-byte accesses aligned.
// ---------------------------------------------------------------------------------------------------------------------
// t2 = TWICE the twiddle (the kernels below stage their twiddles doubled in LDS, once per workgroup and batch of columns): m31.h m_mul_pre2
// 11 VALU per butterfly (product 5, a + t 3, a - t 3), and no lazily reduced form is shorter in 32-bit lanes (r05): p = 2^31 - 1
// leaves one spare bit, so (i) m_mul_pre2 needs its multiplicand < 2^31 — for b < 2^32 the fold high + (low >> 1) of b * 2w can reach
// 2^32 + 2^31 and loses the carry —, (ii) a lazy sum a + t <= 2p may be added to once more at most (3p > 2^32) while half of a layer's outputs are the
// next layer's multiplicands, and (iii) folding the addition into the multiply-add (b * 2w + 2a in 64 bits) saves an add but costs a second
// v_mad_u64_u32 (1.6 issue slots) for the subtracted output.
template <bool INV> __device__ __forceinline__ void bfly(u32& x, u32& y, u32 t2) {
    if (INV) { u32 s = m_add(x, y); y = m_mul_pre2(m_sub(x, y), t2); x = s; }
    else { u32 w = m_mul_pre2(y, t2); y = m_sub(x, w); x = m_add(x, w); }
}
// radix-16 over the 4 local bits of v[16]; tw(jl, idx) returns the twiddle of local layer jl for pair index (e >> (jl+1)).
template <bool INV, class TW> __device__ __forceinline__ void radix16(u32 (&v)[16], u32 nlayers, TW tw) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int jl = INV ? s : 3 - s;
        if ((u32)jl < nlayers) {
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int e0 = ((b >> jl) << (jl + 1)) | (b & ((1 << jl) - 1));
                bfly<INV>(v[e0], v[e0 | (1 << jl)], tw(jl, e0 >> (jl + 1)));
            }
        }
    }
}
__device__ __forceinline__ u32 lds_pad(u32 i) { return i + 4 * (i >> 6); }

// KC != 0: k == KC is a compile-time fact (KC = 9, 10, 11, 12: k0 = layers - 7 x strided passes; the contiguous pass of the three-pass transforms of 2^24 / 2^25 / >= 2^26 cells and of
// every two-pass size): every "layer < k" test folds away, and with them the phi copies the compiler otherwise keeps for the skipped-layer paths
// (r05: profiles/r05_fft_isa_mix.txt). KC = 0: k read from the pass descriptor. One kernel, a uniform branch at its top picks the body.
template <bool INV, int KC>
__device__ __forceinline__ void fft_tile12_body(const PassArgs& a, const BlockOfGroup& bg, u32* __restrict__ s_val, u32* __restrict__ s_tw) {
    const u32 t = threadIdx.x, k = KC ? (u32)KC : a.k, tile = bg.tile;
    const u32 base = tile << 12;
    const u32 col0 = bg.by * a.cols_per_block, col1 = min(a.ncols, col0 + a.cols_per_block);
    // The tile of the next column is fetched while the current one is transformed (16 more registers, same occupancy): without it
    // every column starts with an exposed HBM round trip, which costs ~25 % when a block has only a few columns.
    uint4 nxt[4];
    if (col0 < col1) {
        g_cu32p src0 = as_global(a.src[col0]);
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = ld16(src0 + ((base + 1024 * q + 4 * t) & a.src_mask));
    }
    // stage twiddles of local layers j < k: entry (j, q) at 4096 - 2^(12-j) + q, q < 2^(11-j); global h = (tile << (11-j)) + q.
    // 16 entries per lane with a compile-time layer for each slot (8 of layer 0, 4 of layer 1, 2 of layer 2, 1 of layer 3, 1 of the
    // layers 4..11), so that the 16 loads are issued back to back and waited for once — a rolled loop serialises 16 memory round
    // trips per block, which dominates when a batch has only a few columns per block.
    {
        u32 tw_r[16];
        g_cu32p tw = as_global(a.tw);
        auto table = [&](u32 j) -> g_cu32p { return tw + (layer_table(a, j) - a.tw); };
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const u32 j = s < 8 ? 0u : s < 12 ? 1u : s < 14 ? 2u : s < 15 ? 3u : 4u;            // slot -> layer (4 = "4 and above")
            const u32 e = 256u * s + t;
            u32 v = 0;
            if (j < 4) {
                if (j < k) {
                    const u32 q = e - (4096u - (4096u >> j));
                    const u32 h = (tile << (11 - j)) + q;
                    if (j == 0 && a.circle) {
                        g_cu32p l0 = table(1);
                        const u32 x = l0[(h >> 2) * 2], y = l0[(h >> 2) * 2 + 1], sel = h & 3;
                        v = sel == 0 ? y : sel == 1 ? m_neg(y) : sel == 2 ? m_neg(x) : x;
                    } else v = table(j)[h];
                }
            } else if (e < 4096u - (4096u >> k)) {
                const u32 jj = __clz(~(e << 20));
                const u32 q = e - (4096u - (4096u >> jj));
                v = table(jj)[(tile << (11 - jj)) + q];
            }
            tw_r[s] = v;
        }
#pragma unroll
        for (int s = 0; s < 16; s++) s_tw[256u * s + t] = 2u * tw_r[s];     // doubled: bfly
    }
    auto TW = [&](u32 layer, u32 idx) -> u32 { return s_tw[4096u - (4096u >> layer) + idx]; };
    for (u32 col = col0; col < col1; col++) {
        g_u32p dst = as_global(a.dst[col]);
        __syncthreads();
        u32 r[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++) { r[q][0] = nxt[q].x; r[q][1] = nxt[q].y; r[q][2] = nxt[q].z; r[q][3] = nxt[q].w; }
        if (col + 1 < col1) {
            g_cu32p srcn = as_global(a.src[col + 1]);
#pragma unroll
            for (int q = 0; q < 4; q++) nxt[q] = ld16(srcn + ((base + 1024 * q + 4 * t) & a.src_mask));
        }
        // ---- register stage on the 16-byte groups: layers 0,1 (cells 4t..4t+3 of each quarter q) and 10,11 (across q) --------
        auto stage_low = [&]() {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int L = INV ? s : 1 - s;
                if ((u32)L < k) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        u32 i = 1024 * q + 4 * t;
                        if (L == 0) { bfly<INV>(r[q][0], r[q][1], TW(0, i >> 1)); bfly<INV>(r[q][2], r[q][3], TW(0, (i >> 1) + 1)); }
                        else { u32 w = TW(1, i >> 2); bfly<INV>(r[q][0], r[q][2], w); bfly<INV>(r[q][1], r[q][3], w); }
                    }
                }
            }
        };
        auto stage_high = [&]() {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int L = INV ? 10 + s : 11 - s;
                if ((u32)L < k) {
                    if (L == 10) { u32 w0 = TW(10, 0), w1 = TW(10, 1);
#pragma unroll
                        for (int j = 0; j < 4; j++) { bfly<INV>(r[0][j], r[1][j], w0); bfly<INV>(r[2][j], r[3][j], w1); } }
                    else { u32 w = TW(11, 0);
#pragma unroll
                        for (int j = 0; j < 4; j++) { bfly<INV>(r[0][j], r[2][j], w); bfly<INV>(r[1][j], r[3][j], w); } }
                }
            }
        };
        auto round_lds = [&](u32 first_layer) {
            // 16 cells per lane with tile-index bits [first_layer, first_layer+4) = e
            u32 lo_bits = first_layer;                       // 2 or 6
            u32 lo = t & ((1u << lo_bits) - 1), hi = t >> lo_bits;
            u32 idx0 = (hi << (lo_bits + 4)) | lo;
            u32 v[16];
#pragma unroll
            for (int e = 0; e < 16; e++) v[e] = s_val[lds_pad(idx0 | ((u32)e << lo_bits))];
            u32 nl = k > first_layer ? k - first_layer : 0;
            radix16<INV>(v, nl, [&](int jl, int pe) -> u32 { return TW(first_layer + jl, ((hi << 4) >> (jl + 1)) + pe); });
#pragma unroll
            for (int e = 0; e < 16; e++) s_val[lds_pad(idx0 | ((u32)e << lo_bits))] = v[e];
        };
        auto put = [&]() {
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<uint4*>(&s_val[lds_pad(1024 * q + 4 * t)]) = make_uint4(r[q][0], r[q][1], r[q][2], r[q][3]);
        };
        auto get = [&]() {
#pragma unroll
            for (int q = 0; q < 4; q++) { uint4 v = *reinterpret_cast<uint4*>(&s_val[lds_pad(1024 * q + 4 * t)]); r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w; }
        };
        if (INV) {
            stage_low(); put(); __syncthreads();
            round_lds(2); __syncthreads();
            round_lds(6); __syncthreads();
            get(); stage_high();
        } else {
            stage_high(); put(); __syncthreads();
            round_lds(6); __syncthreads();
            round_lds(2); __syncthreads();
            get(); stage_low();
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = make_uint4(r[q][0], r[q][1], r[q][2], r[q][3]);
            if (INV && a.scale != 1) { const u32 sk = scale_shift(a.scale); v.x = m_mul_pow2(v.x, sk); v.y = m_mul_pow2(v.y, sk); v.z = m_mul_pow2(v.z, sk); v.w = m_mul_pow2(v.w, sk); }
            st16(dst + base + 1024 * q + 4 * t, v);
        }
    }
}

template <bool INV>
__global__ void __launch_bounds__(256) k_fft_tile12(const PassArgs* __restrict__ groups, u32 ngroups, u32 generic) {
    const BlockOfGroup bg = find_group(groups, ngroups);
    const PassArgs a = groups[bg.g];
    __shared__ __attribute__((aligned(16))) u32 s_val[4096 + 4 * 64];
    __shared__ u32 s_tw[4096];
    // guests of this launch (fft_plan): small single-pass transforms and tiny ones, in the same LDS (4096 words each suffice)
    if (a.kind == K_PASS) { fft_pass_body<INV>(a, bg, s_val, s_tw); return; }
    if (a.kind == K_TINY) { fft_tiny_body<INV>(a, bg); return; }
    const u32 k = generic ? 0u : a.k;                  // generic: A/B knob (BFHIP_FFT_TILE12_GENERIC=1), same bytes
    if (k == 12) fft_tile12_body<INV, 12>(a, bg, s_val, s_tw);
    else if (k == 11) fft_tile12_body<INV, 11>(a, bg, s_val, s_tw);
    else if (k == 10) fft_tile12_body<INV, 10>(a, bg, s_val, s_tw);
    else if (k == 9) fft_tile12_body<INV, 9>(a, bg, s_val, s_tw);
    else fft_tile12_body<INV, 0>(a, bg, s_val, s_tw);
}

// CL = log2 of the cells per row: 5 (128-byte rows, 128 lanes) or 6 (256-byte rows, 256 lanes: longer contiguous bursts per DRAM page at
// the same per-lane work; needs lo >= 6). Per lane always 32 cells: 8 rows x 4 cells in the 16-byte phase, 2 x 16 in the 4-byte phase.
template <bool INV, int CL>
__global__ void __launch_bounds__(4 << CL, 3) k_fft_strided7(const PassArgs* __restrict__ groups, u32 ngroups) {
    const BlockOfGroup bg = find_group(groups, ngroups);
    const PassArgs a = groups[bg.g];
    constexpr u32 C = 1u << CL, NT = 4u << CL;
    __shared__ __attribute__((aligned(16))) u32 s_val[128 * C];
    __shared__ u32 s_tw[128];
    const u32 t = threadIdx.x, lo = a.lo, tile = bg.tile;
    const u32 n_lhi_log = lo - CL;
    const u32 H = tile >> n_lhi_log, Lhi = tile & ((1u << n_lhi_log) - 1);
    const u32 base = (H << (lo + 7)) | (Lhi << CL);
    // twiddles: local layer j (global lo + j) entry q < 2^(6-j) at 128 - 2^(7-j) + q; global h = (H << (6-j)) + q
    if (t < 127) {
        u32 j = __clz(~(t << 25));
        u32 q = t - (128u - (128u >> j));
        s_tw[t] = 2u * layer_table(a, lo + j)[(H << (6 - j)) + q];          // doubled: bfly
    }
    auto TW = [&](u32 layer, u32 idx) -> u32 { return s_tw[128u - (128u >> layer) + idx]; };
    const u32 col0 = bg.by * a.cols_per_block, col1 = min(a.ncols, col0 + a.cols_per_block);
    const u32 l4 = 4 * (t & (C / 4 - 1)), r4 = t >> (CL - 2);      // 16-byte phase: rows m = 8*r4 + q (q = 0..7), cells l4..l4+3
    for (u32 col = col0; col < col1; col++) {
        g_cu32p src = as_global(a.src[col]);
        g_u32p dst = as_global(a.dst[col]);
        __syncthreads();
        auto wide_stage = [&](u32 (&r)[8][4]) {   // layers 0..2 over q
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const int jl = INV ? s : 2 - s;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int q0 = ((b >> jl) << (jl + 1)) | (b & ((1 << jl) - 1));
                    u32 w = TW(jl, (8 * r4 + q0) >> (jl + 1));
#pragma unroll
                    for (int j = 0; j < 4; j++) bfly<INV>(r[q0][j], r[q0 | (1 << jl)][j], w);
                }
            }
        };
        auto narrow_task = [&](u32 id, bool from_global, bool to_global) {   // layers 3..6 over e for fixed (mlow3, l)
            u32 l = id & (C - 1), mlow = id >> CL;
            u32 v[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                u32 m = 8 * e + mlow;
                v[e] = from_global ? __builtin_nontemporal_load(src + ((base | (m << lo) | l) & a.src_mask)) : s_val[C * m + l];
            }
            radix16<INV>(v, 4, [&](int jl, int pe) -> u32 { return TW(3 + jl, pe); });
#pragma unroll
            for (int e = 0; e < 16; e++) {
                u32 m = 8 * e + mlow;
                if (to_global) __builtin_nontemporal_store((INV && a.scale != 1) ? m_mul_pow2(v[e], scale_shift(a.scale)) : v[e], dst + (base | (m << lo) | l));   // streamed: +2 %
                else s_val[C * m + l] = v[e];
            }
        };
        if (INV) {
            u32 r[8][4];
#pragma unroll
            for (int q = 0; q < 8; q++) { uint4 v = ld16_stream(src + ((base | ((8 * r4 + q) << lo) | l4) & a.src_mask)); r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w; }
            wide_stage(r);
#pragma unroll
            for (int q = 0; q < 8; q++) *reinterpret_cast<uint4*>(&s_val[C * (8 * r4 + q) + l4]) = make_uint4(r[q][0], r[q][1], r[q][2], r[q][3]);
            __syncthreads();
            narrow_task(t, false, true);
            narrow_task(t + NT, false, true);
        } else {
            narrow_task(t, true, false);
            // r04 ISA audit (tools/isa_mix.py): left to itself the scheduler hoists the 32 loads of both tasks to the top and the forward
            // kernel needs 168 VGPRs + 20 bytes of scratch inside the column loop; with the two tasks kept apart it takes 142 and no scratch
            // (128 x 2^24: 5.17-5.25 -> 5.10-5.14 ms per launch, profiles/r04_fft_isa_audit.txt). Two waves per SIMD instead: slower.
            __builtin_amdgcn_sched_barrier(0);
            narrow_task(t + NT, true, false);
            __syncthreads();
            u32 r[8][4];
#pragma unroll
            for (int q = 0; q < 8; q++) { uint4 v = *reinterpret_cast<uint4*>(&s_val[C * (8 * r4 + q) + l4]); r[q][0] = v.x; r[q][1] = v.y; r[q][2] = v.z; r[q][3] = v.w; }
            wide_stage(r);
#pragma unroll
            for (int q = 0; q < 8; q++) st16(dst + (base | ((8 * r4 + q) << lo) | l4), make_uint4(r[q][0], r[q][1], r[q][2], r[q][3]));
        }
    }
}

// Strided pass of K = 8, 9 or 10 layers (2^20 .. 2^22-cell transforms in TWO passes: 12 contiguous + K strided, instead of 6..8 + 7 + 7):
// 2^K rows of 2^CL cells (rows at stride 2^lo) staged in LDS — 32 KiB for K = 8, 64 KiB for K = 9 (128-byte rows) and K = 10 (64-byte rows) —
// read and written with 16-byte accesses, 32 cells per lane. The layers run as radix-16 rounds over 4 row bits each (16 values per lane
// in registers, lanes of a wave on consecutive cells: conflict-free LDS accesses); the last round covers the top 4 row bits and applies
// only the layers the earlier rounds have not. At 8-10 layers such a pass is as much butterfly work as HBM time (fft.hip header).
template <bool INV, class TW> __device__ __forceinline__ void radix16_range(u32 (&v)[16], int jl_lo, TW tw) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int jl = INV ? s : 3 - s;
        if (jl >= jl_lo) {
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int e0 = ((b >> jl) << (jl + 1)) | (b & ((1 << jl) - 1));
                bfly<INV>(v[e0], v[e0 | (1 << jl)], tw(jl, e0 >> (jl + 1)));
            }
        }
    }
}
template <bool INV, int K, int CL>
__global__ void __launch_bounds__((1 << (K + CL)) / 32) k_fft_stridedK(const PassArgs* __restrict__ groups, u32 ngroups) {
    constexpr u32 C = 1u << CL, ROWS = 1u << K, CELLS = ROWS * C, NT = CELLS / 32;
    constexpr int NR = (K + 3) / 4;                 // rounds
    const BlockOfGroup bg = find_group(groups, ngroups);
    const PassArgs a = groups[bg.g];
    __shared__ __attribute__((aligned(16))) u32 s_val[CELLS];
    __shared__ u32 s_tw[ROWS];
    const u32 t = threadIdx.x, lo = a.lo, tile = bg.tile;
    const u32 n_lhi_log = lo - CL;
    const u32 H = tile >> n_lhi_log, Lhi = tile & ((1u << n_lhi_log) - 1);
    const u32 base = (H << (lo + K)) | (Lhi << CL);
    // twiddles: local layer j (global lo + j), entry q < 2^(K-1-j) at ROWS - (ROWS >> j) + q; global h = (H << (K-1-j)) + q
    for (u32 e = t; e < ROWS - 1; e += NT) {
        const u32 j = __clz(~(e << (32 - K)));
        const u32 q = e - (ROWS - (ROWS >> j));
        s_tw[e] = 2u * layer_table(a, lo + j)[(H << (K - 1 - j)) + q];      // doubled: bfly
    }
    auto TW = [&](u32 layer, u32 idx) -> u32 { return s_tw[ROWS - (ROWS >> layer) + idx]; };
    const u32 col0 = bg.by * a.cols_per_block, col1 = min(a.ncols, col0 + a.cols_per_block);
    for (u32 col = col0; col < col1; col++) {
        g_cu32p src = as_global(a.src[col]);
        g_u32p dst = as_global(a.dst[col]);
        __syncthreads();      // twiddles staged / the previous column's stores have read s_val
#pragma unroll
        for (u32 i = 0; i < 8; i++) {
            const u32 idx = 4 * (t + NT * i), m = idx >> CL, l = idx & (C - 1);
            const uint4 v = ld16_stream(src + ((base | (m << lo) | l) & a.src_mask));
            *reinterpret_cast<uint4*>(&s_val[idx]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < NR; rr++) {
            const int r = INV ? rr : NR - 1 - rr;
            // window of 4 row bits [b, b + 4); the last round is shifted down to the top 4 bits and skips the layers already done
            const int b = (4 * r + 4 <= K) ? 4 * r : K - 4;
            const int jl_lo = 4 * r - b;
#pragma unroll
            for (u32 g2 = 0; g2 < 2; g2++) {
                const u32 id = t + NT * g2, l = id & (C - 1), rest = id >> CL;      // rest: the K - 4 row bits outside the window
                const u32 rest_lo = rest & ((1u << b) - 1), rest_hi = rest >> b;
                const u32 m0 = (rest_hi << (b + 4)) | rest_lo;
                u32 v[16];
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = s_val[((m0 | ((u32)e << b)) << CL) | l];
                radix16_range<INV>(v, jl_lo, [&](int jl, int pe) -> u32 { return TW(b + jl, (rest_hi << (3 - jl)) + pe); });
#pragma unroll
                for (int e = 0; e < 16; e++) s_val[((m0 | ((u32)e << b)) << CL) | l] = v[e];
            }
            __syncthreads();
        }
#pragma unroll
        for (u32 i = 0; i < 8; i++) {
            const u32 idx = 4 * (t + NT * i), m = idx >> CL, l = idx & (C - 1);
            uint4 v = *reinterpret_cast<uint4*>(&s_val[idx]);
            if (INV && a.scale != 1) { const u32 sk = scale_shift(a.scale); v.x = m_mul_pow2(v.x, sk); v.y = m_mul_pow2(v.y, sk); v.z = m_mul_pow2(v.z, sk); v.w = m_mul_pow2(v.w, sk); }
            st16(dst + (base | (m << lo) | l), v);
        }
    }
}

// Tiny transforms (log <= 5: the 16..32-cell columns of empty sub-component tables, end_of_execution, and their row-granular forms down to ONE cell): one LANE per
// cell, 64 >> log columns per wave, the butterflies as lane exchanges (__shfl_xor) — a launch of six such kernels per proof used to cost 13-15 us each (one thread
// per column looping over a 32-word array in scratch memory: a latency chain), r06: the whole transform is five dependent exchanges behind one round of loads.
template <bool INV>
__device__ __forceinline__ void fft_tiny_body(const PassArgs& a, const BlockOfGroup& bg) {
    const u32 n = 1u << a.log, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & (n - 1);
    const u32 col = (bg.tile * (blockDim.x >> 6) + wave) * (64u >> a.log) + (lane >> a.log);
    const bool act = col < a.ncols;
    u32 v = act ? a.src[col][i & a.src_mask] : 0u;
    // this cell's twiddle in every layer (pair block h = i >> (j + 1)): independent loads, one round trip
    u32 tw[5];
#pragma unroll
    for (u32 j = 0; j < 5; j++) {
        tw[j] = 1u;
        if (j < a.k) {
            const u32 h = i >> (j + 1);
            if (a.circle && j == 0) {
                const u32* l0 = layer_table(a, 1);
                const u32 x = l0[(h >> 2) * 2], y = l0[(h >> 2) * 2 + 1], sel = h & 3;
                tw[j] = sel == 0 ? y : sel == 1 ? m_neg(y) : sel == 2 ? m_neg(x) : x;
            } else tw[j] = layer_table(a, j)[h];
        }
    }
#pragma unroll
    for (u32 jj = 0; jj < 5; jj++) {
        if (jj >= a.k) break;                                         // uniform
        const u32 j = INV ? jj : a.k - 1 - jj;
        const u32 other = (u32)__shfl_xor((int)v, 1 << j);           // every lane of the wave takes part (inactive columns carry zeros)
        const bool hi = (i >> j) & 1u;
        const u32 x = hi ? other : v, y = hi ? v : other;            // (v0, v1) of this cell's butterfly; both lanes of a pair compute the product
        u32 t = 0;
#pragma unroll
        for (u32 q = 0; q < 5; q++) if (q == j) t = tw[q];
        if (INV) v = hi ? m_mul(m_sub(x, y), t) : m_add(x, y);
        else { const u32 w = m_mul(y, t); v = hi ? m_sub(x, w) : m_add(x, w); }
    }
    if (act) a.dst[col][i] = INV ? m_mul(v, a.scale) : v;
}
template <bool INV>
__global__ void __launch_bounds__(256) k_fft_tiny(const PassArgs* __restrict__ groups, u32 ngroups) {
    const BlockOfGroup bg = find_group(groups, ngroups);
    const PassArgs a = groups[bg.g];
    fft_tiny_body<INV>(a, bg);
}

// Host-side pass planner. Inverse: contiguous pass first then strided passes upward; forward: mirror image.
// fft_plan lays out the passes of every job (one job = the columns of one size and storage); the pass a job executes pi-th goes into
// the launch (pi, kernel kind), so a batch of jobs costs at most (passes of the largest job) x (kinds in use) launches.

void fft_plan(FftPlan& plan, bool inverse, const FftJob* jobs, size_t njobs, const u32* tw, const u32* itw, u32 tw_root_log) {
    plan.inverse = inverse; plan.groups.clear(); plan.launches.clear(); plan.d_groups = nullptr;
    struct Item { int pi, kind; PassArgs a; double bytes, alg; bool single = false; double bfly() const { return 0.5 * a.ncols * (double)(1u << a.log) * a.k; } };      // butterflies of the pass
    std::vector<Item> items;
    int max_np = 0;
    for (size_t ji = 0; ji < njobs; ji++) {
        const FftJob& job = jobs[ji];
        const u32 ncols = job.ncols, log = job.log, src_log = job.src_log;
        if (ncols == 0) continue;
        PassArgs a{};
        a.ncols = ncols; a.log = log; a.circle = job.circle ? 1 : 0; a.tw = inverse ? itw : tw; a.tw_total = 1u << tw_root_log; a.scale = 1;
        if (log > 5 && src_log < 2) throw std::runtime_error("fft: extending a polynomial with fewer than 4 coefficients to more than 32 cells is not supported");
        const u32 nl = inverse ? log : src_log;   // forward: the layers >= src_log only duplicate (zero extension) = wrap-around load
        if (log <= 5) {
            a.dst = job.d_dst; a.src = job.d_src; a.src_mask = (1u << src_log) - 1; a.lo = 0; a.k = nl;
            a.scale = inverse ? m_inv(1u << log) : 1;
            a.grid_x = (ncols + (256u >> log) - 1) / (256u >> log); a.cols_per_block = 1;      // k_fft_tiny: a lane per cell, 64 >> log columns per wave, 4 waves per workgroup
            items.push_back({0, K_TINY, a, 0.0, 0.0, true});
            max_np = std::max(max_np, 1);
            continue;
        }
        // ---- fast path: 2^12-cell contiguous pass with k0 in [6,12] layers + strided passes of exactly 7 layers ----------------
        if (log >= 12 && nl >= 6) {
            u32 ns = nl > 12 ? (nl - 12 + 6) / 7 : 0;
            u32 k0 = nl - 7 * ns;
            // 20..22 layers: 12 contiguous + ONE strided pass of 8..10 layers through LDS (two passes instead of three)
            static const bool two_pass = [] { const char* v = getenv("BFHIP_FFT_TWO_PASS"); return !v || v[0] != '0'; }();
            const u32 big_k = (two_pass && nl >= 20 && nl <= 22) ? nl - 12 : 0;
            if (big_k) { ns = 1; k0 = 12; }
            int np = 1 + (int)ns;
            max_np = std::max(max_np, np);
            // profiler accounting: `bytes` = what this pass moves (4 B in + 4 B out per cell), `alg` = this pass's share of the transform's
            // ALGORITHMIC bytes (iFFT 8N, LDE 12N per column = read the input once, write the output once)
            const double alg = 4.0 * ncols * ((double)(1u << src_log) + (double)(1u << log)) / np;
            for (int pi = 0; pi < np; pi++) {
                int p = inverse ? pi : np - 1 - pi;
                bool first = pi == 0, last = pi == np - 1;
                a.src = first ? job.d_src : (const u32* const*)job.d_dst;
                a.dst = job.d_dst;
                a.src_mask = first ? ((1u << src_log) - 1) : 0xffffffffu;
                a.scale = (inverse && last) ? m_inv(1u << log) : 1;
                u32 ntiles = 1u << (log - 12);
                // one workgroup walks as many columns as possible per tile (twiddles staged once), as long as >= 2048 workgroups remain
                u32 cpb = ncols;
                while (cpb > 1 && (u64)ntiles * ((ncols + cpb - 1) / cpb) < 2048) cpb = (cpb + 1) / 2;
                a.cols_per_block = cpb;
                const u32 gy = (ncols + cpb - 1) / cpb;
                const double bytes = 8.0 * ncols * (double)(1u << log);
                if (p == 0) {
                    a.lo = 0; a.k = k0; a.tile_log = 12; a.grid_x = ntiles;
                    PassArgs b = a; b.block0 = gy;      // block0 temporarily holds the group's grid_y
                    items.push_back({pi, K_TILE12, b, bytes, alg});
                } else if (big_k) {
                    a.lo = 12; a.k = big_k;
                    const u32 cl = big_k == 10 ? 4 : 5;                       // 64-byte rows at 10 layers (64 KiB of LDS), 128-byte rows otherwise
                    PassArgs b = a; b.grid_x = 1u << (log - big_k - cl); b.block0 = gy;
                    items.push_back({pi, big_k == 8 ? K_STRIDED_K8 : big_k == 9 ? K_STRIDED_K9 : K_STRIDED_K10, b, bytes, alg});
                } else {
                    a.lo = k0 + 7 * (p - 1); a.k = 7;
#ifndef BF_STRIDED_WIDE_MIN_LOG
#define BF_STRIDED_WIDE_MIN_LOG 20
#endif
                    // 256-byte rows for big transforms (enough tiles to fill the chip twice over), 128-byte rows otherwise
                    const bool wide = a.lo >= 6 && log >= BF_STRIDED_WIDE_MIN_LOG;
                    PassArgs b = a; b.grid_x = wide ? ntiles / 2 : ntiles; b.block0 = gy;
                    items.push_back({pi, wide ? K_STRIDED6 : K_STRIDED5, b, bytes, alg});
                }
            }
            continue;
        }
        // pass boundaries: [0, k0) contiguous, then strided passes [k0, k0 + k1), ...
        u32 bounds[8]; int np = 0;
        bounds[0] = 0;
        a.tile_log = log < (u32)TILE_LOG ? log : (u32)TILE_LOG;
        u32 k0 = nl < a.tile_log ? nl : a.tile_log;
        bounds[++np] = k0;
        while (bounds[np] < nl) {
            u32 rem = nl - bounds[np];
            u32 passes_left = (rem + STRIDED_K - 1) / STRIDED_K;
            u32 kk = (rem + passes_left - 1) / passes_left;  // balance the strided passes
            bounds[np + 1] = bounds[np] + kk; np++;
        }
        max_np = std::max(max_np, np);
        const double alg = 4.0 * ncols * ((double)(1u << src_log) + (double)(1u << log)) / np;
        for (int pi = 0; pi < np; pi++) {
            int p = inverse ? pi : np - 1 - pi;
            bool first = pi == 0, last = pi == np - 1;
            a.lo = bounds[p]; a.k = bounds[p + 1] - bounds[p];
            u32 tl = a.lo == 0 ? a.tile_log : a.k + CHUNK_LOG;
            u32 ntiles = 1u << (log - tl);
            a.src = first ? job.d_src : (const u32* const*)job.d_dst;
            a.dst = job.d_dst;
            a.src_mask = first ? ((1u << src_log) - 1) : 0xffffffffu;
            a.scale = (inverse && last) ? m_inv(1u << log) : 1;
            // columns per block: enough blocks to fill the chip, as few twiddle re-loads as possible
            u32 cpb = 1;
            while ((u64)ntiles * ((ncols + cpb - 1) / cpb) > 8192 && cpb < ncols) cpb *= 2;
            a.cols_per_block = cpb;
            PassArgs b = a; b.grid_x = ntiles; b.block0 = (ncols + cpb - 1) / cpb;
            items.push_back({pi, K_PASS, b, 8.0 * ncols * (double)(1u << log), alg, np == 1 && log < (u32)TILE_LOG});
        }
    }
    // The small fry rides along: single-pass transforms of < 2^12 cells and the tiny ones touch columns of their own, so they can join ANY launch of the plan — the
    // first contiguous-tile launch takes them as guest groups (PassArgs::kind). BFHIP_FFT_FUSE_SMALL=0: A/B knob (separate launches as before; same bytes).
    {
        static const bool fuse = [] { const char* v = getenv("BFHIP_FFT_FUSE_SMALL"); return !v || v[0] != '0'; }();
        int host_pi = -1;
        for (auto& it : items) if (it.kind == K_TILE12 && (host_pi < 0 || it.pi < host_pi)) host_pi = it.pi;
        if (fuse && host_pi >= 0)
            for (auto& it : items)
                if (it.single && (it.kind == K_PASS || it.kind == K_TINY)) { it.a.kind = (u32)it.kind; it.kind = K_TILE12; it.pi = host_pi; }
    }
    for (int pi = 0; pi < max_np; pi++)
        for (int kind = 0; kind < K_KINDS; kind++) {
            FftLaunch L{}; L.kind = kind; L.first_group = (u32)plan.groups.size();
            u32 blocks = 0;
            for (auto& it : items) {
                if (it.pi != pi || it.kind != kind) continue;
                PassArgs a = it.a;
                const u32 gy = (kind == K_TINY || a.kind == K_TINY) ? 1u : a.block0;
                a.block0 = blocks; blocks += a.grid_x * gy;
                plan.groups.push_back(a);
                L.bytes += it.bytes; L.alg += it.alg; L.bfly += it.bfly();
            }
            L.ngroups = (u32)plan.groups.size() - L.first_group; L.total_blocks = blocks;
            if (L.ngroups) plan.launches.push_back(L);
        }
}

// BFHIP_FFT_PROF_DETAIL=1 (tools/fft_inproof.py): the profiler's record of an FFT launch carries its shape — "name/b<workgroups>/g<size groups>" — so that
// the in-proof launches can be priced one by one (butterflies and bytes per launch against its time) instead of per kernel name.
static const char* fft_prof_name(const char* base, const FftLaunch& L) {
    static const bool detail = [] { const char* v = getenv("BFHIP_FFT_PROF_DETAIL"); return v && v[0] == '1'; }();
    if (!detail) return base;
    static std::mutex mu; static std::set<std::string> names;
    std::lock_guard<std::mutex> g(mu);
    return names.insert(std::string(base) + "/b" + std::to_string(L.total_blocks) + "/g" + std::to_string(L.ngroups)).first->c_str();
}

void fft_run(hipStream_t stream, const FftPlan& plan) {
    if (plan.launches.empty()) return;
    if (!plan.d_groups) throw std::runtime_error("fft_run: the plan's group table has not been staged");
    const bool inverse = plan.inverse;
    for (const FftLaunch& L : plan.launches) {
        const PassArgs* g = plan.d_groups + L.first_group;
        const dim3 grid(L.total_blocks);
        switch (L.kind) {
            case K_TILE12: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_tile12<true>" : "k_fft_tile12<false>", L), L.bytes, L.alg, false, L.bfly);
                static const u32 generic = [] { const char* v = getenv("BFHIP_FFT_TILE12_GENERIC"); return (v && v[0] == '1') ? 1u : 0u; }();
                if (inverse) hipLaunchKernelGGL(k_fft_tile12<true>, grid, dim3(256), 0, stream, g, L.ngroups, generic);
                else hipLaunchKernelGGL(k_fft_tile12<false>, grid, dim3(256), 0, stream, g, L.ngroups, generic);
                break; }
            case K_STRIDED5: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_strided7<true>" : "k_fft_strided7<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL((k_fft_strided7<true, 5>), grid, dim3(128), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL((k_fft_strided7<false, 5>), grid, dim3(128), 0, stream, g, L.ngroups);
                break; }
            case K_STRIDED6: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_strided7<true>" : "k_fft_strided7<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL((k_fft_strided7<true, 6>), grid, dim3(256), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL((k_fft_strided7<false, 6>), grid, dim3(256), 0, stream, g, L.ngroups);
                break; }
            case K_STRIDED_K8: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_stridedK<true>" : "k_fft_stridedK<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL((k_fft_stridedK<true, 8, 5>), grid, dim3(256), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL((k_fft_stridedK<false, 8, 5>), grid, dim3(256), 0, stream, g, L.ngroups);
                break; }
            case K_STRIDED_K9: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_stridedK<true>" : "k_fft_stridedK<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL((k_fft_stridedK<true, 9, 5>), grid, dim3(512), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL((k_fft_stridedK<false, 9, 5>), grid, dim3(512), 0, stream, g, L.ngroups);
                break; }
            case K_STRIDED_K10: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_stridedK<true>" : "k_fft_stridedK<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL((k_fft_stridedK<true, 10, 4>), grid, dim3(512), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL((k_fft_stridedK<false, 10, 4>), grid, dim3(512), 0, stream, g, L.ngroups);
                break; }
            case K_PASS: {
                ProfScope ps(stream, fft_prof_name(inverse ? "k_fft_pass<true>" : "k_fft_pass<false>", L), L.bytes, L.alg, false, L.bfly);
                if (inverse) hipLaunchKernelGGL(k_fft_pass<true>, grid, dim3(FFT_THREADS), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL(k_fft_pass<false>, grid, dim3(FFT_THREADS), 0, stream, g, L.ngroups);
                break; }
            default:
                if (inverse) hipLaunchKernelGGL(k_fft_tiny<true>, grid, dim3(256), 0, stream, g, L.ngroups);
                else hipLaunchKernelGGL(k_fft_tiny<false>, grid, dim3(256), 0, stream, g, L.ngroups);
        }
    }
}

void gen_twiddles(hipStream_t stream, u32* d_tw, u32* d_itw, u32 R, const uint2* d_tlo, const uint2* d_thi) {
    u32 total = 1u << R;
    hipLaunchKernelGGL(k_gen_twiddles, dim3((total + 255) / 256), dim3(256), 0, stream, d_tw, d_itw, R, d_tlo, d_thi);
}

}  // namespace bf
