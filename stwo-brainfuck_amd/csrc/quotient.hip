// Out-of-domain sampling, DEEP/FRI quotients, FRI folding and query gathers for gfx950 — SURVEY.md §8 rows a8, a9, a10, a12.
// Replaces stwo `PolyOps::eval_at_point`, `QuotientOps::accumulate_quotients`, `FriOps::{fold_circle_into_line, fold_line}` and
// the column reads of the Merkle/FRI decommitment, all reached from prover::prove at
// crates/brainfuck_prover/src/brainfuck_air/mod.rs:732 (CommitmentSchemeProver::prove_values).
#include <algorithm>
#include "kernels.h"
#include <stdexcept>

namespace bf {

__device__ __forceinline__ Q31 ld_q(const uint4* p, size_t i) { uint4 v = p[i]; return q_make(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 pk_q(Q31 q) { return make_uint4(q.a.a, q.a.b, q.b.a, q.b.b); }

// ------------------------------------------------------------------------------------------------------------------------------
// eval_at_point (a8): f(P) = sum_j c_j * prod_{b : bit b of j set} F[b],  F = [P.y, P.x, 2x^2-1, ...]  (stwo `fold`).
// Stage 1: one workgroup folds a chunk of 4096 coefficients (16 per lane against a 16-entry weight table, then an LDS tree);
// Stage 2: one workgroup per job folds the chunk partials with the remaining factors. Fixed reduction tree => deterministic.
// Row-granular (replicated) coefficient columns hold only the coefficients of index 0 mod 16, so their factors start at F[4].
// ------------------------------------------------------------------------------------------------------------------------------
static constexpr u32 EAP_CHUNK_LOG = 12;

// Grid: one workgroup per (job, chunk) pair, flattened — workgroup b belongs to the job with partial_off <= b (binary search over the
// jobs' partial offsets, which are exactly the flattened chunk indices), so no workgroup is launched only to exit.
__global__ void __launch_bounds__(256) k_eval_at_point_stage1(const EvalJob* __restrict__ jobs, u32 n_jobs, const uint4* __restrict__ factors, uint4* __restrict__ partials) {
    u32 lo_j = 0, hi_j = n_jobs;
    while (hi_j - lo_j > 1) { u32 mid = (lo_j + hi_j) >> 1; if (jobs[mid].partial_off <= blockIdx.x) lo_j = mid; else hi_j = mid; }
    const EvalJob job = jobs[lo_j];
    const u32 n = 1u << job.log_n;
    const u32 chunk = blockIdx.x - job.partial_off;
    __shared__ uint4 s_w[16];
    __shared__ uint4 s_p[256];
    const uint4* F = factors + (size_t)job.point * 32 + job.factor_shift;   // F[b] multiplies bit b of the (row-granular) index
    const u32 t = threadIdx.x;
    if (t < 16) {
        Q31 w = q_one();
        for (u32 b = 0; b < 4; b++) if ((t >> b) & 1) w = q_mul(w, ld_q(F, b));
        s_w[t] = pk_q(w);
    }
    __syncthreads();
    const u32 base = (chunk << EAP_CHUNK_LOG) + t * 16;
    Q31 acc = q_zero();
    if (base < n) {
        if (n >= 16) {
            const uint4* src = reinterpret_cast<const uint4*>(job.coeffs + base);
            // 16-term dot product (QM31 weights x M31 coefficients) in 64-bit accumulators with lazy reduction (m31.h)
            u64 a64[4] = {0, 0, 0, 0};
#pragma unroll
            for (u32 v4 = 0; v4 < 4; v4++) {
                const uint4 c = src[v4];
                const u32 cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (u32 j = 0; j < 4; j++) {
                    const u32 idx = v4 * 4 + j;
                    // four products of canonical values on top of a folded accumulator (< 2^34) stay below 2^64: 4 (p - 1)^2 + 2^34 < 2^64
                    if (idx % 4 == 0 && idx) { a64[0] = m_fold(a64[0]); a64[1] = m_fold(a64[1]); a64[2] = m_fold(a64[2]); a64[3] = m_fold(a64[3]); }
                    const Q31 w = ld_q(s_w, idx);
                    a64[0] += (u64)w.a.a * cv[j]; a64[1] += (u64)w.a.b * cv[j]; a64[2] += (u64)w.b.a * cv[j]; a64[3] += (u64)w.b.b * cv[j];
                }
            }
            acc = q_make(m_red4(a64[0]), m_red4(a64[1]), m_red4(a64[2]), m_red4(a64[3]));
        } else {
            for (u32 k = 0; k < n; k++) acc = q_add(acc, q_mulm(ld_q(s_w, k), job.coeffs[k]));
        }
    }
    s_p[t] = pk_q(acc);
    __syncthreads();
    // tree over the 256 lane partials: level b uses factor F[4 + b] — a constant of the level, so the product is the 4 x 4 constant-matrix
    // form (m31.h: q_mul_const; the tree's nine wave-level products cost more than the 16-term dot products in front of it, r04)
    for (u32 b = 0; b < 8; b++) {
        u32 half = 128u >> b;
        Q31 r;
        bool act = t < half;
        if (act) {
            Q31 lo = ld_q(s_p, 2 * t), hi = ld_q(s_p, 2 * t + 1);
            // partial index bit b corresponds to coefficient-index bit 4 + b; beyond log_n the hi partial is zero anyway
            r = (4 + b < job.log_n) ? q_add(lo, q_mul_const(hi, q_const(ld_q(F, 4 + b)))) : lo;
        }
        __syncthreads();
        if (act) s_p[t] = pk_q(r);
        __syncthreads();
    }
    if (t == 0) partials[(size_t)job.partial_off + chunk] = s_p[0];
}

__global__ void __launch_bounds__(256) k_eval_at_point_stage2(const EvalJob* __restrict__ jobs, const uint4* __restrict__ factors, const uint4* __restrict__ partials, uint4* __restrict__ out) {
    const EvalJob job = jobs[blockIdx.x];
    const u32 nchunks = job.log_n > EAP_CHUNK_LOG ? 1u << (job.log_n - EAP_CHUNK_LOG) : 1u;
    const uint4* F = factors + (size_t)job.point * 32 + job.factor_shift;
    const uint4* p = partials + job.partial_off;
    __shared__ uint4 s[256];
    const u32 t = threadIdx.x;
    // each lane folds a contiguous run of `per` partials (per = nchunks / 256 when nchunks > 256), then an LDS tree
    u32 lanes = nchunks < 256 ? nchunks : 256;
    u32 per_log = 0; while ((lanes << per_log) < nchunks) per_log++;
    Q31 acc = q_zero();
    if (t < lanes) {
        // sequential fold of 2^per_log partials: weight of local index k = prod of F[12 + b] over set bits b of k
        u32 per = 1u << per_log;
        for (u32 k = 0; k < per; k++) {
            Q31 w = q_one();
            for (u32 b = 0; b < per_log; b++) if ((k >> b) & 1) w = q_mul(w, ld_q(F, EAP_CHUNK_LOG + b));
            acc = q_add(acc, q_mul(ld_q(p, ((size_t)t << per_log) + k), w));
        }
    }
    s[t] = pk_q(acc);
    __syncthreads();
    u32 lanes_log = 0; while ((1u << lanes_log) < lanes) lanes_log++;
    for (u32 b = 0; b < lanes_log; b++) {
        u32 half = lanes >> (b + 1);
        Q31 r; bool act = t < half;
        if (act) r = q_add(ld_q(s, 2 * t), q_mul_const(ld_q(s, 2 * t + 1), q_const(ld_q(F, EAP_CHUNK_LOG + per_log + b))));
        __syncthreads();
        if (act) s[t] = pk_q(r);
        __syncthreads();
    }
    if (t == 0) out[job.out_idx] = s[0];
}

// total_partials = sum over jobs of their chunk counts (= partial_off of the last job + its chunk count); jobs sorted by partial_off.
void eval_at_points(hipStream_t stream, const EvalJob* d_jobs, u32 n_jobs, u32 total_partials, const void* d_factors, void* d_partials, void* d_out) {
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_eval_at_point_stage1, dim3(total_partials), dim3(256), 0, stream, d_jobs, n_jobs, (const uint4*)d_factors, (uint4*)d_partials);
    hipLaunchKernelGGL(k_eval_at_point_stage2, dim3(n_jobs), dim3(256), 0, stream, d_jobs, (const uint4*)d_factors, (const uint4*)d_partials, (uint4*)d_out);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Domain points from the twiddle tree: for CanonicCoset(log).circle_domain() in bit-reversed order,
//   x(row) = +-T1[row >> 2] (negated when bit 1 of row is set), y(row) = +-circle_twiddle(row >> 1) (negated when bit 0 is set),
// where T1 is the first line-layer table of the domain (2^(log-2) entries at tw + 2^R - 2^(log-1)).
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void domain_point(const u32* __restrict__ tw, u32 tw_total, u32 log, u32 row, u32& x, u32& y) {
    const u32* t1 = tw + (tw_total - (1u << (log - 1)));
    u32 xv = t1[row >> 2];
    x = (row & 2) ? m_neg(xv) : xv;
    u32 h = row >> 1;
    u32 cx = t1[(h >> 2) * 2], cy = t1[(h >> 2) * 2 + 1];
    u32 sel = h & 3;
    u32 yv = sel == 0 ? cy : sel == 1 ? m_neg(cy) : sel == 2 ? m_neg(cx) : cx;
    y = (row & 1) ? m_neg(yv) : yv;
}

// ------------------------------------------------------------------------------------------------------------------------------
// accumulate_quotients (a9): row value = Horner over sample batches of  (sum_k c_k f_k(row) - (A y + B)) / den(batch, row)
// with A = sum_k a_k, B = sum_k b_k (line coefficients of complex_conjugate_line_coeffs, pre-summed on the host).
// ------------------------------------------------------------------------------------------------------------------------------
// Each lane owns 4 consecutive rows: full-size columns are read as one 16-byte access, replicated columns as one word, and the 4
// denominators of a batch share one M31 inversion (Montgomery trick on the CM31 norms) — values identical to 4 separate inverses.
// One launch covers every size group of a proof: groups[g] describes group g, its workgroups are [block0, next group's block0).
__global__ void __launch_bounds__(256) k_quotients(const QuotientArgs* __restrict__ groups, u32 n_groups) {
    u32 g = 0;
    while (g + 1 < n_groups && groups[g + 1].block0 <= blockIdx.x) g++;      // uniform: scalar loads
    const QuotientArgs& a = groups[g];
    u32 row0 = ((blockIdx.x - a.block0) * blockDim.x + threadIdx.x) * 4;
    if (a.n_rows) { if (row0 >= a.n_rows) return; row0 += a.row0; }
    else if (row0 >= (1u << a.log)) return;
    // The lane's 4 rows are the points (x, y), (x, -y), (-x, -y), (-x, y) of the bit-reversed domain (domain_point: rows 4j..4j+3 share
    // one twiddle-table entry up to signs), so every product with a row coordinate is formed once and only its sign varies per row.
    u32 xv, yv;
    domain_point(a.tw, a.tw_total, a.log, row0, xv, yv);
    Q31 acc[4] = {q_zero(), q_zero(), q_zero(), q_zero()};
    u32 e = 0;
    for (u32 b = 0; b < a.n_batches; b++) {
        const QuotientBatch qb = ld_constant(a.batches + b);
        // numerators sum_k c_k * f_k(row): 64-bit dot products with lazy reduction (m31.h: m_fold / m_canon). Three products of < 2^62 fit on
        // top of a folded accumulator (< 2^34), so the columns are taken three at a time: the loads of the NEXT three are issued before the
        // products of the current three (the loop is latency-bound otherwise: entry -> pointer -> cell is one dependent chain per column and
        // only the 5 resident waves overlap them). A batch lists its full-size columns first (n_full, one 16-byte access for the lane's 4
        // rows), then the columns stored once per 4+ rows, which go into one accumulator shared by the lane's 4 rows.
        u64 n64[4][4], s64[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { n64[r][0] = 0; n64[r][1] = 0; n64[r][2] = 0; n64[r][3] = 0; }
        s64[0] = 0; s64[1] = 0; s64[2] = 0; s64[3] = 0;
        const QuotientEntry* ent = a.entries + e;
        e += qb.n_cols;
        const u32 nf = qb.n_full, nr = qb.n_cols - qb.n_full;
        // Two buffers of three columns, used in turn: the loads into one are in flight during the products of the other. The loads are
        // unconditional (indices past the end re-read the last column and are not used): a load that is only issued on some paths makes the
        // wait before the products a wait for everything.
        if (nf) {
            uint4 A[3], B[3];
            const u32 last = nf - 1;
            auto load3 = [&](uint4 (&buf)[3], u32 k) {
#pragma unroll
                for (u32 j = 0; j < 3; j++) buf[j] = ld16(as_global(ld_constant(&ent[min(k + j, last)].ptr)) + row0);
            };
            auto mac3 = [&](const uint4 (&buf)[3], u32 k) {
#pragma unroll
                for (int r = 0; r < 4; r++) { n64[r][0] = m_fold(n64[r][0]); n64[r][1] = m_fold(n64[r][1]); n64[r][2] = m_fold(n64[r][2]); n64[r][3] = m_fold(n64[r][3]); }
#pragma unroll
                for (u32 j = 0; j < 3; j++) {
                    if (k + j < nf) {
                        const Q31 cc = ld_constant(&ent[k + j].c);
                        const u32 vr[4] = {buf[j].x, buf[j].y, buf[j].z, buf[j].w};
#pragma unroll
                        for (int r = 0; r < 4; r++) { n64[r][0] += (u64)cc.a.a * vr[r]; n64[r][1] += (u64)cc.a.b * vr[r]; n64[r][2] += (u64)cc.b.a * vr[r]; n64[r][3] += (u64)cc.b.b * vr[r]; }
                    }
                }
            };
            load3(A, 0);
            for (u32 k = 0; k < nf; k += 6) {
                load3(B, k + 3);
                mac3(A, k);
                load3(A, k + 6);
                mac3(B, k + 3);
            }
        }
        if (nr) {
            const QuotientEntry* er = ent + nf;
            u32 A[3], B[3];
            const u32 last = nr - 1;
            auto load3 = [&](u32 (&buf)[3], u32 k) {
#pragma unroll
                for (u32 j = 0; j < 3; j++) { const u32 i = min(k + j, last); buf[j] = as_global(ld_constant(&er[i].ptr))[row0 >> ld_constant(&er[i].shift)]; }
            };
            auto mac3 = [&](const u32 (&buf)[3], u32 k) {
                s64[0] = m_fold(s64[0]); s64[1] = m_fold(s64[1]); s64[2] = m_fold(s64[2]); s64[3] = m_fold(s64[3]);
#pragma unroll
                for (u32 j = 0; j < 3; j++) {
                    if (k + j < nr) {
                        const Q31 cc = ld_constant(&er[k + j].c);
                        s64[0] += (u64)cc.a.a * buf[j]; s64[1] += (u64)cc.a.b * buf[j]; s64[2] += (u64)cc.b.a * buf[j]; s64[3] += (u64)cc.b.b * buf[j];
                    }
                }
            };
            load3(A, 0);
            for (u32 k = 0; k < nr; k += 6) {
                load3(B, k + 3);
                mac3(A, k);
                load3(A, k + 6);
                mac3(B, k + 3);
            }
        }
        // line part A y + B for y = +yv (rows 0, 3) and y = -yv (rows 1, 2)
        const Q31 ay = q_mulm(qb.a_sum, yv);
        const Q31 line_p = q_add(qb.b_sum, ay), line_m = q_sub(qb.b_sum, ay);
        // den = (Pr.x - x) * Pi.y - (Pr.y - y) * Pi.x in CM31 = kden - x * Pi.y + y * Pi.x (x, y in M31): kden -+ P +- Q with P = xv * Pi.y, Q = yv * Pi.x
        const C31 P = {m_mul(xv, qb.piy.a), m_mul(xv, qb.piy.b)}, Q = {m_mul(yv, qb.pix.a), m_mul(yv, qb.pix.b)};
        const C31 km = c_sub(qb.kden, P), kp = c_add(qb.kden, P);
        C31 den[4] = {c_add(km, Q), c_sub(km, Q), c_sub(kp, Q), c_add(kp, Q)};
        Q31 num[4]; u32 nrm[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            num[r] = q_make(m_canon(m_fold(n64[r][0]) + m_fold(s64[0])), m_canon(m_fold(n64[r][1]) + m_fold(s64[1])),
                            m_canon(m_fold(n64[r][2]) + m_fold(s64[2])), m_canon(m_fold(n64[r][3]) + m_fold(s64[3])));
            num[r] = q_sub(num[r], (r == 0 || r == 3) ? line_p : line_m);
            nrm[r] = m_add(m_sqr(den[r].a), m_sqr(den[r].b));
        }
        u32 p01 = m_mul(nrm[0], nrm[1]), p012 = m_mul(p01, nrm[2]), inv_all = m_inv(m_mul(p012, nrm[3]));
        u32 i3 = m_mul(inv_all, p012), t3 = m_mul(inv_all, nrm[3]);
        u32 i2 = m_mul(t3, p01), t2 = m_mul(t3, nrm[2]);
        u32 i1 = m_mul(t2, nrm[0]), i0 = m_mul(t2, nrm[1]);
        u32 ninv[4] = {i0, i1, i2, i3};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            C31 dinv = {m_mul(den[r].a, ninv[r]), m_neg(m_mul(den[r].b, ninv[r]))};
            Q31 term = q_mulc(num[r], dinv);
            acc[r] = b ? q_add(acc[r], term) : term;     // the batches' Horner weights are folded into their constants (host/quotients.h)
        }
    }
    *reinterpret_cast<uint4*>(a.out[0] + row0) = make_uint4(acc[0].a.a, acc[1].a.a, acc[2].a.a, acc[3].a.a);
    *reinterpret_cast<uint4*>(a.out[1] + row0) = make_uint4(acc[0].a.b, acc[1].a.b, acc[2].a.b, acc[3].a.b);
    *reinterpret_cast<uint4*>(a.out[2] + row0) = make_uint4(acc[0].b.a, acc[1].b.a, acc[2].b.a, acc[3].b.a);
    *reinterpret_cast<uint4*>(a.out[3] + row0) = make_uint4(acc[0].b.b, acc[1].b.b, acc[2].b.b, acc[3].b.b);
}
void quotient_entries_finish(QuotientBatch* batches, size_t n_batches, QuotientEntry* entries, const ColDesc* cols) {
    for (size_t b = 0; b < n_batches; b++) {
        QuotientEntry* first = entries; QuotientEntry* last = entries + batches[b].n_cols;
        for (QuotientEntry* q = first; q != last; q++) { q->ptr = cols[q->col].ptr; q->shift = cols[q->col].shift; }
        batches[b].n_full = (u32)(std::stable_partition(first, last, [](const QuotientEntry& q) { return q.shift == 0; }) - first);
        entries = last;
    }
}
// h_groups: host copy of the table (block0 is filled in here BEFORE the caller stages it: call quotient_groups_layout first)
u32 quotient_groups_layout(QuotientArgs* h_groups, u32 n_groups) {
    u32 blocks = 0;
    for (u32 g = 0; g < n_groups; g++) {
        const u32 lanes = (h_groups[g].n_rows ? h_groups[g].n_rows : (1u << h_groups[g].log)) / 4;   // 4 rows per lane (all quotient domains have >= 2^5 rows)
        h_groups[g].block0 = blocks;
        blocks += (lanes + 255) / 256;
    }
    return blocks;
}
void accumulate_quotients(hipStream_t stream, const QuotientArgs* d_groups, u32 n_groups, u32 total_blocks) {
    if (!n_groups || !total_blocks) return;
    ProfScope ps(stream, "k_quotients", 0);
    hipLaunchKernelGGL(k_quotients, dim3(total_blocks), dim3(256), 0, stream, d_groups, n_groups);
}

// ------------------------------------------------------------------------------------------------------------------------------
// FRI folds (a10). Inverse twiddles come from the inverse twiddle tree:
//   circle -> line: pair i of a circle evaluation of size 2^log uses 1/y = inverse circle twiddle i of the domain;
//   line fold     : pair i of a line evaluation of size 2^log over Coset::half_odds(log) uses 1/x = first layer entry i of the
//                   layered buffer of that coset (tail of the tree: itw + 2^R - 2^log).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fold_circle_into_line(u32* const d0, u32* const d1, u32* const d2, u32* const d3,
                                                               const u32* __restrict__ s0, const u32* __restrict__ s1, const u32* __restrict__ s2, const u32* __restrict__ s3,
                                                               const u32* __restrict__ alpha8, const u32* __restrict__ itw, u32 tw_total, u32 log, u32 fresh, u32 first, u32 count) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    i += first;
    const QConst alpha = q_const(q_make(alpha8[0], alpha8[1], alpha8[2], alpha8[3])), alpha_sq = q_const(q_make(alpha8[4], alpha8[5], alpha8[6], alpha8[7]));
    const u32* t1 = itw + (tw_total - (1u << (log - 1)));
    u32 cx = t1[(i >> 2) * 2], cy = t1[(i >> 2) * 2 + 1], sel = i & 3;
    u32 yinv = sel == 0 ? cy : sel == 1 ? m_neg(cy) : sel == 2 ? m_neg(cx) : cx;
    uint2 a0 = reinterpret_cast<const uint2*>(s0)[i], a1 = reinterpret_cast<const uint2*>(s1)[i], a2 = reinterpret_cast<const uint2*>(s2)[i], a3 = reinterpret_cast<const uint2*>(s3)[i];
    Q31 fp = q_make(a0.x, a1.x, a2.x, a3.x), fn = q_make(a0.y, a1.y, a2.y, a3.y);
    Q31 f0 = q_add(fp, fn), f1 = q_mulm(q_sub(fp, fn), yinv);
    Q31 fprime = q_add(q_mul_const(f1, alpha), f0);
    Q31 r = fprime;                                   // fresh destination: 0 * alpha^2 + f'
    if (!fresh) { Q31 dst = q_make(d0[i], d1[i], d2[i], d3[i]); r = q_add(q_mul_const(dst, alpha_sq), fprime); }
    d0[i] = r.a.a; d1[i] = r.a.b; d2[i] = r.b.a; d3[i] = r.b.b;
}
__global__ void __launch_bounds__(256) k_fold_line(u32* __restrict__ d0, u32* __restrict__ d1, u32* __restrict__ d2, u32* __restrict__ d3,
                                                   const u32* __restrict__ s0, const u32* __restrict__ s1, const u32* __restrict__ s2, const u32* __restrict__ s3,
                                                   const u32* __restrict__ alpha8, const u32* __restrict__ itw, u32 tw_total, u32 log, u32 first, u32 count) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    i += first;
    const QConst alpha = q_const(q_make(alpha8[0], alpha8[1], alpha8[2], alpha8[3]));
    u32 xinv = itw[tw_total - (1u << log) + i];
    uint2 a0 = reinterpret_cast<const uint2*>(s0)[i], a1 = reinterpret_cast<const uint2*>(s1)[i], a2 = reinterpret_cast<const uint2*>(s2)[i], a3 = reinterpret_cast<const uint2*>(s3)[i];
    Q31 fx = q_make(a0.x, a1.x, a2.x, a3.x), fn = q_make(a0.y, a1.y, a2.y, a3.y);
    Q31 f0 = q_add(fx, fn), f1 = q_mulm(q_sub(fx, fn), xinv);
    Q31 r = q_add(f0, q_mul_const(f1, alpha));
    d0[i] = r.a.a; d1[i] = r.a.b; d2[i] = r.b.a; d3[i] = r.b.b;
}
// One FRI step below the first layer as ONE launch: next = fold_line(cur, alpha), then — when a quotient column of cur's size exists —
// next = next * alpha^2 + fold_circle(quotient, alpha) (FriProver::commit_inner_layers: fold_line of the previous layer, then the
// fold-in of the circle evaluation whose folded size equals the new line size; both use the alpha drawn after cur's commitment).
__global__ void __launch_bounds__(256) k_fold_line_circle(u32* __restrict__ d0, u32* __restrict__ d1, u32* __restrict__ d2, u32* __restrict__ d3,
                                                          const u32* __restrict__ s0, const u32* __restrict__ s1, const u32* __restrict__ s2, const u32* __restrict__ s3,
                                                          const u32* __restrict__ q0, const u32* __restrict__ q1, const u32* __restrict__ q2, const u32* __restrict__ q3,
                                                          const u32* __restrict__ alpha8, const u32* __restrict__ itw, u32 tw_total, u32 log, u32 first, u32 count) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    i += first;
    const QConst alpha = q_const(q_make(alpha8[0], alpha8[1], alpha8[2], alpha8[3]));
    Q31 r;
    {
        const u32 xinv = itw[tw_total - (1u << log) + i];
        const uint2 a0 = reinterpret_cast<const uint2*>(s0)[i], a1 = reinterpret_cast<const uint2*>(s1)[i], a2 = reinterpret_cast<const uint2*>(s2)[i], a3 = reinterpret_cast<const uint2*>(s3)[i];
        const Q31 fx = q_make(a0.x, a1.x, a2.x, a3.x), fn = q_make(a0.y, a1.y, a2.y, a3.y);
        r = q_add(q_add(fx, fn), q_mul_const(q_mulm(q_sub(fx, fn), xinv), alpha));
    }
    if (q0) {
        const QConst alpha_sq = q_const(q_make(alpha8[4], alpha8[5], alpha8[6], alpha8[7]));
        const u32* t1 = itw + (tw_total - (1u << (log - 1)));
        const u32 cx = t1[(i >> 2) * 2], cy = t1[(i >> 2) * 2 + 1], sel = i & 3;
        const u32 yinv = sel == 0 ? cy : sel == 1 ? m_neg(cy) : sel == 2 ? m_neg(cx) : cx;
        const uint2 a0 = reinterpret_cast<const uint2*>(q0)[i], a1 = reinterpret_cast<const uint2*>(q1)[i], a2 = reinterpret_cast<const uint2*>(q2)[i], a3 = reinterpret_cast<const uint2*>(q3)[i];
        const Q31 fp = q_make(a0.x, a1.x, a2.x, a3.x), fn = q_make(a0.y, a1.y, a2.y, a3.y);
        const Q31 fprime = q_add(q_mul_const(q_mulm(q_sub(fp, fn), yinv), alpha), q_add(fp, fn));
        r = q_add(q_mul_const(r, alpha_sq), fprime);
    }
    d0[i] = r.a.a; d1[i] = r.a.b; d2[i] = r.b.a; d3[i] = r.b.b;
}
void fold_line_circle(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* const quot[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log,
                      u32 first, u32 count) {
    if (!count) { first = 0; count = 1u << (log - 1); }
    hipLaunchKernelGGL(k_fold_line_circle, dim3((count + 255) / 256), dim3(256), 0, stream, dst[0], dst[1], dst[2], dst[3], src[0], src[1], src[2], src[3],
                       quot ? quot[0] : nullptr, quot ? quot[1] : nullptr, quot ? quot[2] : nullptr, quot ? quot[3] : nullptr, d_alpha8, itw, 1u << tw_root_log, log, first, count);
}

// d_alpha8 = device pointer to alpha[4] || alpha^2[4] (written by k_channel_mix_root_draw or staged from the host)
void fold_circle_into_line(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log, bool fresh,
                           u32 first, u32 count) {
    if (!count) { first = 0; count = 1u << (log - 1); }
    hipLaunchKernelGGL(k_fold_circle_into_line, dim3((count + 255) / 256), dim3(256), 0, stream, dst[0], dst[1], dst[2], dst[3], src[0], src[1], src[2], src[3],
                       d_alpha8, itw, 1u << tw_root_log, log, fresh ? 1u : 0u, first, count);
}
void fold_line(hipStream_t stream, u32* const dst[4], const u32* const src[4], const u32* d_alpha8, const u32* itw, u32 tw_root_log, u32 log, u32 first, u32 count) {
    if (!count) { first = 0; count = 1u << (log - 1); }
    hipLaunchKernelGGL(k_fold_line, dim3((count + 255) / 256), dim3(256), 0, stream, dst[0], dst[1], dst[2], dst[3], src[0], src[1], src[2], src[3], d_alpha8, itw, 1u << tw_root_log, log,
                       first, count);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Decommitment gathers (a12): out[j] = word `word` of element `index` of a u32 array (column cell or hash word).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void k_gather(const GatherReq* __restrict__ req, u32 n, u32* __restrict__ out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    GatherReq r = req[j];
    // null base: words another rank of the shard group owns (filled in by the max-reduce)
    for (u32 w = 0; w < r.n_words; w++) out[r.out_off + w] = r.base ? r.base[r.index + w] : 0u;
}
void gather_u32(hipStream_t stream, const GatherReq* d_req, u32 n, u32* d_out) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, stream, d_req, n, d_out);
}

// AccumulationOps::accumulate: dst += src (M31, elementwise) — used when merging the per-size composition accumulators.
__global__ void k_accumulate(u32* __restrict__ dst, const u32* __restrict__ src, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = m_add(dst[i], src[i]);
}
void accumulate(hipStream_t stream, u32* dst, const u32* src, u32 n) {
    hipLaunchKernelGGL(k_accumulate, dim3((n + 255) / 256), dim3(256), 0, stream, dst, src, n);
}
// dst[w][i] += sum over the sources s with i < 2^log[s] of src[s][w][i], for the 4 coordinate columns w: the composition polynomial's
// coefficients = the largest size's plus the zero-extended coefficients of every smaller size (DomainEvaluationAccumulator::finalize by
// linearity), one launch instead of 4 per size.
__global__ void __launch_bounds__(256) k_accumulate_sizes(AccumulateSizes a) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (i >= (1u << a.log[0])) return;      // sources are sorted by descending size
    u32 v = a.dst[w][i];
    for (u32 s = 0; s < a.n; s++) { if (i >= (1u << a.log[s])) break; v = m_add(v, a.src[s][w][i]); }
    a.dst[w][i] = v;
}
void accumulate_sizes(hipStream_t stream, const AccumulateSizes& a) {
    if (!a.n) return;
    for (u32 s = 1; s < a.n; s++) if (a.log[s] > a.log[s - 1]) throw std::runtime_error("accumulate_sizes: sources must be sorted by descending size");
    hipLaunchKernelGGL(k_accumulate_sizes, dim3(((1u << a.log[0]) + 255) / 256, 4), dim3(256), 0, stream, a);
}

// FieldOps::batch_inverse over M31: 8 elements per lane share one inversion (Montgomery trick) — values equal elementwise inverses.
__global__ void __launch_bounds__(256) k_batch_inverse_m31(const u32* __restrict__ src, u32* __restrict__ dst, u32 n) {
    u32 base = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (base >= n) return;
    u32 v[8], pre[8];
    u32 cnt = min(8u, n - base);
    u32 acc = 1;
#pragma unroll
    for (u32 k = 0; k < 8; k++) { v[k] = k < cnt ? src[base + k] : 1u; pre[k] = acc; acc = m_mul(acc, v[k]); }
    u32 inv = m_inv(acc);
#pragma unroll
    for (int k = 7; k >= 0; k--) { u32 r = m_mul(inv, pre[k]); inv = m_mul(inv, v[k]); if ((u32)k < cnt) dst[base + k] = r; }
}
void batch_inverse_m31(hipStream_t stream, const u32* src, u32* dst, u32 n) {
    if (!n) return;
    u32 lanes = (n + 7) / 8;
    hipLaunchKernelGGL(k_batch_inverse_m31, dim3((lanes + 255) / 256), dim3(256), 0, stream, src, dst, n);
}

// FieldOps::batch_inverse over QM31 (SecureColumnByCoords, 4 coordinate columns): one norm inversion per element; the M31 inversion
// of 4 consecutive elements is shared (Montgomery trick) — values equal elementwise inverses.
__global__ void __launch_bounds__(256) k_batch_inverse_qm31(const u32* s0, const u32* s1, const u32* s2, const u32* s3, u32* d0, u32* d1, u32* d2, u32* d3, u32 n) {   // dst may alias src
    u32 base = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (base >= n) return;
    u32 cnt = min(4u, n - base);
    Q31 x[4]; C31 den[4]; u32 nrm[4], pre[4];
    u32 acc = 1;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
        u32 i = base + (k < cnt ? k : 0);
        x[k] = q_make(s0[i], s1[i], s2[i], s3[i]);
        den[k] = c_sub(c_mul(x[k].a, x[k].a), c_mulR(c_mul(x[k].b, x[k].b)));      // a^2 - (2 + i) b^2 in CM31
        nrm[k] = m_add(m_sqr(den[k].a), m_sqr(den[k].b));
        pre[k] = acc; acc = m_mul(acc, nrm[k]);
    }
    u32 inv = m_inv(acc);
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        u32 ninv = m_mul(inv, pre[k]); inv = m_mul(inv, nrm[k]);
        C31 dinv = {m_mul(den[k].a, ninv), m_neg(m_mul(den[k].b, ninv))};
        Q31 r = {c_mul(x[k].a, dinv), c_neg(c_mul(x[k].b, dinv))};
        if ((u32)k < cnt) { u32 i = base + k; d0[i] = r.a.a; d1[i] = r.a.b; d2[i] = r.b.a; d3[i] = r.b.b; }
    }
}
void batch_inverse_qm31(hipStream_t stream, const u32* const src[4], u32* const dst[4], u32 n) {
    if (!n) return;
    u32 lanes = (n + 3) / 4;
    hipLaunchKernelGGL(k_batch_inverse_qm31, dim3((lanes + 255) / 256), dim3(256), 0, stream, src[0], src[1], src[2], src[3], dst[0], dst[1], dst[2], dst[3], n);
}

// ColumnOps::bit_reverse_column (not on the prove path — the reference stores traces already bit-reversed — but part of the
// backend surface): out-of-place permutation.
__global__ void k_bit_reverse(const u32* __restrict__ src, u32* __restrict__ dst, u32 log) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (1u << log)) dst[bit_rev(i, log)] = src[i];
}
void bit_reverse(hipStream_t stream, const u32* src, u32* dst, u32 log) {
    u32 n = 1u << log;
    hipLaunchKernelGGL(k_bit_reverse, dim3((n + 255) / 256), dim3(256), 0, stream, src, dst, log);
}

// gen_is_first (preprocessed columns, mod.rs:497): only its interpolation matters; the coefficients of the indicator of cell 0
// are written directly by the FFT of a one-hot column, so this just builds the one-hot column.
__global__ void k_one_hot(u32* __restrict__ dst, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i == 0 ? 1u : 0u;
}
void one_hot(hipStream_t stream, u32* dst, u32 n) { hipLaunchKernelGGL(k_one_hot, dim3((n + 255) / 256), dim3(256), 0, stream, dst, n); }

}  // namespace bf
