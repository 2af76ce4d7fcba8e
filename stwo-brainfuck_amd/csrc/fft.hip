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

namespace bf {

static constexpr int TILE_LOG = 12;          // 4096 elements = 16 KiB of LDS
static constexpr int TILE = 1 << TILE_LOG;
static constexpr int CHUNK_LOG = 5;          // 32 x u32 = 128 B contiguous per strided row
static constexpr int STRIDED_K = TILE_LOG - CHUNK_LOG;  // 7 layers per strided pass
static constexpr int FFT_THREADS = 256;

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
// One pass = layers [lo, lo+k) of a size-2^log transform, for the columns cols[blockIdx.y * cpb .. +cpb).
// ---------------------------------------------------------------------------------------------------------------------
struct PassArgs {
    u32* const* dst;          // device array of column pointers (output / in-place)
    const u32* const* src;    // device array of column pointers (input of this pass)
    u32 ncols, cols_per_block;
    u32 log;                  // transform size
    u32 lo, k;                // layers [lo, lo + k)
    u32 tile_log;             // contiguous pass (lo == 0): tile = 2^tile_log >= 2^k cells; strided: tile = 2^(k + CHUNK_LOG)
    u32 src_mask;             // input index mask (2^src_log - 1): forward zero-extension = wrap-around load
    u32 circle;               // 1: layer 0 is the circle layer; 0: line mode
    u32 scale;                // inverse: multiply outputs by this (1 = none)
    const u32* tw;            // twiddle (forward) or inverse-twiddle (inverse) layered buffer
    u32 tw_total;             // 2^R
};

__device__ __forceinline__ const u32* layer_table(const PassArgs& a, u32 layer) {
    // line layer `layer` (>= 1 in circle mode, >= 0 in line mode): len = 2^(log-1-layer), table at tw_total - 2*len.
    // In line mode local layer j plays the role of circle-mode layer j+1 of a transform twice the size.
    u32 eff = a.circle ? layer : layer + 1;
    u32 lg = a.circle ? a.log : a.log + 1;
    u32 len = 1u << (lg - 1 - eff);
    return a.tw + (a.tw_total - 2 * len);
}

template <bool INV>
__global__ void __launch_bounds__(FFT_THREADS) k_fft_pass(PassArgs a) {
    __shared__ u32 s_val[TILE];
    __shared__ u32 s_tw[TILE];  // per-layer twiddle segments of this tile, packed
    const u32 t = threadIdx.x;
    const u32 lo = a.lo, k = a.k;
    const u32 c = lo == 0 ? 0 : CHUNK_LOG;
    const u32 tile_log = lo == 0 ? a.tile_log : k + c;
    const u32 tile_n = 1u << tile_log;
    const u32 kt = tile_log - c;   // twiddle span: local layer j needs 2^(kt-1-j) entries, staged at offset 2^kt - 2^(kt-j)
    const u32 tile = blockIdx.x;
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

    const u32 col0 = blockIdx.y * a.cols_per_block;
    const u32 col1 = min(a.ncols, col0 + a.cols_per_block);
    for (u32 col = col0; col < col1; col++) {
        const u32* src = a.src[col];
        u32* dst = a.dst[col];
        __syncthreads();  // s_tw ready / previous column's stores done reading s_val
        // ---- load tile (16 B per lane) -----------------------------------------------------------------------------
        for (u32 e4 = t * 4; e4 < tile_n; e4 += 4 * FFT_THREADS) {
            u32 gidx = lo ? (base | ((e4 >> c) << lo) | (e4 & ((1u << c) - 1))) : (base + e4);
            uint4 v = *reinterpret_cast<const uint4*>(src + (gidx & a.src_mask));
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
            if (INV && a.scale != 1) { v.x = m_mul(v.x, a.scale); v.y = m_mul(v.y, a.scale); v.z = m_mul(v.z, a.scale); v.w = m_mul(v.w, a.scale); }
            *reinterpret_cast<uint4*>(dst + gidx) = v;
        }
    }
}

// Tiny transforms (log <= 5): one thread per column, straight loops over registers/local memory. Only the handful of
// 16..32-cell columns of empty sub-component tables take this route.
template <bool INV>
__global__ void k_fft_tiny(PassArgs a) {
    u32 col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= a.ncols) return;
    u32 n = 1u << a.log;
    u32 v[32];
    for (u32 i = 0; i < n; i++) v[i] = a.src[col][i & a.src_mask];
    for (u32 jj = 0; jj < a.k; jj++) {
        u32 j = INV ? jj : a.k - 1 - jj;
        u32 dist = 1u << j;
        for (u32 b = 0; b < n / 2; b++) {
            u32 i0 = ((b >> j) << (j + 1)) | (b & (dist - 1));
            u32 h = b >> j, tw;
            if (a.circle && j == 0) {
                const u32* l0 = layer_table(a, 1);
                u32 x = l0[(h >> 2) * 2], y = l0[(h >> 2) * 2 + 1], sel = h & 3;
                tw = sel == 0 ? y : sel == 1 ? m_neg(y) : sel == 2 ? m_neg(x) : x;
            } else tw = layer_table(a, j)[h];
            if (INV) { u32 v0 = v[i0], v1 = v[i0 + dist]; v[i0] = m_add(v0, v1); v[i0 + dist] = m_mul(m_sub(v0, v1), tw); }
            else { u32 v0 = v[i0], v1 = m_mul(v[i0 + dist], tw); v[i0] = m_add(v0, v1); v[i0 + dist] = m_sub(v0, v1); }
        }
    }
    for (u32 i = 0; i < n; i++) a.dst[col][i] = INV ? m_mul(v[i], a.scale) : v[i];
}

// Host-side pass planner. Inverse: contiguous pass first then strided passes upward; forward: mirror image.
void fft_batch(hipStream_t stream, bool inverse, const u32* const* d_src, u32* const* d_dst, u32 ncols, u32 log, u32 src_log, bool circle,
               const u32* tw, const u32* itw, u32 tw_root_log) {
    if (ncols == 0) return;
    PassArgs a{};
    a.ncols = ncols; a.log = log; a.circle = circle ? 1 : 0; a.tw = inverse ? itw : tw; a.tw_total = 1u << tw_root_log; a.scale = 1;
    if (log > 5 && src_log < 2) { fprintf(stderr, "bfhip: fft_batch: src_log < 2 with log > 5 unsupported\n"); abort(); }
    const u32 nl = inverse ? log : src_log;   // forward: the layers >= src_log only duplicate (zero extension) = wrap-around load
    if (log <= 5) {
        a.dst = d_dst; a.src = d_src; a.src_mask = (1u << src_log) - 1; a.lo = 0; a.k = nl;
        a.scale = inverse ? m_inv(1u << log) : 1;
        if (inverse) hipLaunchKernelGGL(k_fft_tiny<true>, dim3((ncols + 63) / 64), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(k_fft_tiny<false>, dim3((ncols + 63) / 64), dim3(64), 0, stream, a);
        return;
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
    for (int pi = 0; pi < np; pi++) {
        int p = inverse ? pi : np - 1 - pi;
        bool first = pi == 0, last = pi == np - 1;
        a.lo = bounds[p]; a.k = bounds[p + 1] - bounds[p];
        u32 tl = a.lo == 0 ? a.tile_log : a.k + CHUNK_LOG;
        u32 ntiles = 1u << (log - tl);
        a.src = first ? d_src : (const u32* const*)d_dst;
        a.dst = d_dst;
        a.src_mask = first ? ((1u << src_log) - 1) : 0xffffffffu;
        a.scale = (inverse && last) ? m_inv(1u << log) : 1;
        // columns per block: enough blocks to fill the chip, as few twiddle re-loads as possible
        u32 cpb = 1;
        while ((u64)ntiles * ((ncols + cpb - 1) / cpb) > 8192 && cpb < ncols) cpb *= 2;
        a.cols_per_block = cpb;
        dim3 grid(ntiles, (ncols + cpb - 1) / cpb);
        ProfScope ps(stream, inverse ? "k_fft_pass<true>" : "k_fft_pass<false>", 8.0 * ncols * (double)(1u << log));
        if (inverse) hipLaunchKernelGGL(k_fft_pass<true>, grid, dim3(FFT_THREADS), 0, stream, a);
        else hipLaunchKernelGGL(k_fft_pass<false>, grid, dim3(FFT_THREADS), 0, stream, a);
    }
}

void gen_twiddles(hipStream_t stream, u32* d_tw, u32* d_itw, u32 R, const uint2* d_tlo, const uint2* d_thi) {
    u32 total = 1u << R;
    hipLaunchKernelGGL(k_gen_twiddles, dim3((total + 255) / 256), dim3(256), 0, stream, d_tw, d_itw, R, d_tlo, d_thi);
}

}  // namespace bf
