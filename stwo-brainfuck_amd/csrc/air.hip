// Constraint-quotient evaluation on the LDE domain and logUp interaction-trace generation (row a5).
//
// Constraint kernels replace stwo's `FrameworkComponent::evaluate_constraint_quotients_on_domain` (SimdBackend only upstream)
// driven from prover::prove, crates/brainfuck_prover/src/brainfuck_air/mod.rs:732, over the 13 evals of
// crates/brainfuck_prover/src/components/**/component.rs. One fused kernel per AIR: a lane owns one LDE row, reads every column of
// the component exactly once (coalesced; 16x-replicated main columns are read row-granular), evaluates all constraints in M31,
// the logUp constraints in QM31, combines them with the random-coefficient powers, multiplies by 1/vanishing and accumulates
// into the per-size QM31 accumulator (4 x u32 SoA).
//
// logUp kernels replace `LogupTraceGenerator::{new_col, write_frac, finalize_col, finalize_last}` as used by the
// interaction_trace_evaluation functions (memory/table.rs:485-518, instruction/table.rs:456, program/table.rs:233,
// processor/table.rs:456-529, instructions/table.rs:466, jump/table.rs:436, end_of_execution/table.rs:220).
#include "kernels.h"
#include "air.h"
#include <cstdlib>
#include <stdexcept>

namespace bf {

// ------------------------------------------------------------------------------------------------------------------------------
// Constraint evaluation
// ------------------------------------------------------------------------------------------------------------------------------
typedef ConstraintLaunch ConstraintArgs;   // lives in HBM (staged by the host); read through scalar loads


// LDE row (bit-reversed storage, blowup 2) of the point at trace-coset offset -1 from LDE row `row` of a component of 2^log_size rows
__device__ __forceinline__ u32 prev_lde_row(u32 row, u32 log_size) {
    u32 el = log_size + 1, half = 1u << log_size;
    u32 d = bit_rev(row, el);
    u32 pd = d < half ? ((d + half - 1) & (half - 1)) : (((d - half + 1) & (half - 1)) + half);
    return bit_rev(pd, el);
}
__global__ void __launch_bounds__(256) k_prev_row_copy(u32* __restrict__ dst, const u32* __restrict__ src, u32 log_size) {
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < (2u << log_size)) dst[row] = src[prev_lde_row(row, log_size)];
}
void prev_row_copy(hipStream_t stream, u32* dst, const u32* src, u32 log_size) {
    u32 n = 2u << log_size;
    hipLaunchKernelGGL(k_prev_row_copy, dim3((n + 255) / 256), dim3(256), 0, stream, dst, src, log_size);
}

struct DomainEval : LogupState<DomainEval, Fm> {
    typedef Fm F;
    const ConstraintArgs& a; u32 row; int ti = 0, ii = 0, ci = 0;
    // sum_j coeff_j * c_j. Base-field constraints (all but the logUp ones) are accumulated as four 64-bit dot products with lazy
    // reduction: a product of canonical values is < 2^62, so three of them fit on top of a folded accumulator (< 2^34) before the
    // next fold x -> (x & P) + (x >> 31). One multiply-add per coordinate instead of a full modular multiply and add; the canonical
    // result is the same.
    u64 acc[4] = {0, 0, 0, 0}; int pending = 0; Q31 res_ext;
    __device__ DomainEval(const ConstraintArgs& a_, u32 row_) : a(a_), row(row_) { res_ext = q_zero(); total_sum = a_.total_sum; }
    __device__ __forceinline__ void fold() {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = m_fold(acc[k]);
        pending = 0;
    }
    __device__ __forceinline__ Q31 result() { return q_add(q_make(m_canon(acc[0]), m_canon(acc[1]), m_canon(acc[2]), m_canon(acc[3])), res_ext); }
    __device__ __forceinline__ Fm is_first() { return {as_global(a.is_first)[row]}; }
    __device__ __forceinline__ Fm trace() { return {ld_col(a.trace[ti++], row)}; }
    __device__ __forceinline__ Fm cst(u32 k) { return {k}; }
    __device__ __forceinline__ Q31 rd(int i0, u32 r) {
        return q_make(ld_col(a.inter[i0], r), ld_col(a.inter[i0 + 1], r), ld_col(a.inter[i0 + 2], r), ld_col(a.inter[i0 + 3], r));
    }
    __device__ __forceinline__ Fq inter_cur() { Fq v{rd(ii, row)}; ii += 4; return v; }
    __device__ __forceinline__ void inter_cur_prev(Fq& cur, Fq& prev) {
        // previous trace row = point - trace_step: on the 2x LDE domain (bit-reversed storage) this is d-1 cyclic in the first
        // half-coset and d+1 cyclic in the conjugate half (stwo offset_bit_reversed_circle_domain_index with offset -1).
        cur.v = rd(ii, row);
        if (a.inter_prev[0]) {   // materialised previous-row copy (row-sharded columns)
            prev.v = q_make(as_global(a.inter_prev[0])[row], as_global(a.inter_prev[1])[row], as_global(a.inter_prev[2])[row], as_global(a.inter_prev[3])[row]);
        } else prev.v = rd(ii, prev_lde_row(row, a.log_size));
        ii += 4;
    }
    __device__ __forceinline__ void constraint(Fm c) {
        const Q31 k = a.coeff[ci++];
        if (pending == 4) fold();      // four products of canonical values on top of a folded accumulator stay below 2^64: 4 (p - 1)^2 + 2^34 < 2^64
        acc[0] += (u64)k.a.a * c.v; acc[1] += (u64)k.a.b * c.v; acc[2] += (u64)k.b.a * c.v; acc[3] += (u64)k.b.b * c.v;
        pending++;
    }
    __device__ __forceinline__ void constraint(Fq c) { res_ext = q_add(res_ext, q_mul(a.coeff[ci++], c.v)); }
};

template <int COMP>
__global__ void __launch_bounds__(256) k_constraints(const ConstraintArgs* __restrict__ ap) {
    const ConstraintArgs& a = *ap;
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.n_rows) { if (row >= a.n_rows) return; row += a.row0; }
    else if (row >= (2u << a.log_size)) return;
    DomainEval e(a, row);
    air_eval<COMP>(e, a.el);
    Q31 r = q_mulm(e.result(), a.denom_inv[row >> a.log_size]);
    g_u32p acc0 = as_global(a.acc[0]), acc1 = as_global(a.acc[1]), acc2 = as_global(a.acc[2]), acc3 = as_global(a.acc[3]);
    if (a.overwrite) { acc0[row] = r.a.a; acc1[row] = r.a.b; acc2[row] = r.b.a; acc3[row] = r.b.b; return; }
    acc0[row] = m_add(acc0[row], r.a.a);
    acc1[row] = m_add(acc1[row], r.a.b);
    acc2[row] = m_add(acc2[row], r.b.a);
    acc3[row] = m_add(acc3[row], r.b.b);
}

// ---- row-group variant --------------------------------------------------------------------------------------------------------
// Every main-trace column and every logUp column but the last is 16x replicated, so within 16 consecutive LDE rows
// only IsFirst (t), the last logUp column (cur) and its previous-row value (prev) change, and the combined constraint value is affine in
// them:   sum_j coeff_j c_j  =  A + t * B + K * (cur - prev)      (A, B, K in QM31, functions of the replicated cells only)
//   base constraints  t * g_j   -> B += coeff_j g_j ; the others -> A += coeff_j c_j
//   non-last logUp    (cur_k - prev_col) d - n  is replicated    -> A
//   last logUp        (cur - (prev - total t) - prev_col) d - n  -> K = coeff d, B += K total, A -= K prev_col + coeff n
// The AIR is evaluated once per group of 16 rows, then 24 products per row instead of ~100. Exact field arithmetic: the same canonical
// values as the per-row kernel (which stays for columns that are not stored replicated).
struct IsFirstTag {};
struct FmT { u32 v; };   // t * v
__device__ __forceinline__ FmT operator*(IsFirstTag, Fm x) { return {x.v}; }

struct GroupEval {
    typedef Fm F;
    const ConstraintArgs& a; u32 row; int ti = 0, ii = 0, ci = 0;
    u64 accA[4] = {0, 0, 0, 0}, accB[4] = {0, 0, 0, 0}; int pendA = 0, pendB = 0;
    Q31 extA, extB, K, prev_col;
    __device__ GroupEval(const ConstraintArgs& a_, u32 row_) : a(a_), row(row_) { extA = q_zero(); extB = q_zero(); K = q_zero(); prev_col = q_zero(); }
    __device__ __forceinline__ IsFirstTag is_first() { return {}; }
    __device__ __forceinline__ Fm trace() { return {ld_col(a.trace[ti++], row)}; }
    __device__ __forceinline__ Fm cst(u32 k) { return {k}; }
    __device__ __forceinline__ void dot(u64 (&acc)[4], int& pending, const Q31& k, u32 v) {
        if (pending == 4) { for (int w = 0; w < 4; w++) acc[w] = m_fold(acc[w]); pending = 0; }      // 4 (p - 1)^2 + 2^34 < 2^64
        acc[0] += (u64)k.a.a * v; acc[1] += (u64)k.a.b * v; acc[2] += (u64)k.b.a * v; acc[3] += (u64)k.b.b * v;
        pending++;
    }
    __device__ __forceinline__ void constraint(Fm c) { dot(accA, pendA, a.coeff[ci++], c.v); }
    __device__ __forceinline__ void constraint(FmT c) { dot(accB, pendB, a.coeff[ci++], c.v); }
    __device__ __forceinline__ void logup_mid(Fq n, Fq d) {
        const Q31 cur = q_make(ld_col(a.inter[ii], row), ld_col(a.inter[ii + 1], row), ld_col(a.inter[ii + 2], row), ld_col(a.inter[ii + 3], row));
        ii += 4;
        const Q31 diff = q_sub(cur, prev_col);
        prev_col = cur;
        extA = q_add(extA, q_mul(a.coeff[ci++], q_sub(q_mul(diff, d.v), n.v)));
    }
    __device__ __forceinline__ void logup_last(Fq n, Fq d) {
        const Q31 c = a.coeff[ci++];
        K = q_mul(c, d.v);
        extB = q_add(extB, q_mul(K, a.total_sum));
        extA = q_sub(extA, q_add(q_mul(K, prev_col), q_mul(c, n.v)));
    }
    __device__ __forceinline__ Q31 A() { return q_add(q_make(m_canon(accA[0]), m_canon(accA[1]), m_canon(accA[2]), m_canon(accA[3])), extA); }
    __device__ __forceinline__ Q31 B() { return q_add(q_make(m_canon(accB[0]), m_canon(accB[1]), m_canon(accB[2]), m_canon(accB[3])), extB); }
};

// One workgroup covers 4096 rows = 256 groups of 16: every lane first evaluates the AIR for one group (replicated cells: consecutive lanes
// read consecutive stored cells) and leaves A, B, K (times 1/vanishing) in LDS; then every lane walks 4 quads of rows, 256 quads apart, so
// that the full-size columns and the accumulator are read and written as coalesced 16-byte accesses.
// Previous row of the last logUp column: in bit-reversed storage prev(row) = row ^ M, M covering only high bits (the leading zeros / ones of
// row >> 1 and the bit below them; even rows step back in the first half-coset, odd rows forward in the conjugate one), so the four rows of a
// quad read elements 0, 2 of the quad at prev(r0) and elements 1, 3 of the quad at prev(r0 + 1) - 1 — except next to the wrap-around.
template <int COMP>
__device__ __forceinline__ void constraints_block_body(const ConstraintArgs& a, uint4* __restrict__ s_abk, u32 block) {
    const u32 n = a.n_rows ? a.n_rows : 2u << a.log_size, row_first = a.n_rows ? a.row0 : 0u;
    const u32 base = block * 4096u;
    {
        const u32 rel = base + threadIdx.x * 16u;
        if (rel < n) {
            const u32 row = row_first + rel;
            GroupEval e(a, row);
            air_eval<COMP>(e, a.el);
            const u32 dinv = a.denom_inv[row >> a.log_size];      // a group lies in one half of the domain
            const Q31 A = q_mulm(e.A(), dinv), B = q_mulm(e.B(), dinv), K = q_mulm(e.K, dinv);
            s_abk[threadIdx.x * 3] = make_uint4(A.a.a, A.a.b, A.b.a, A.b.b);
            s_abk[threadIdx.x * 3 + 1] = make_uint4(B.a.a, B.a.b, B.b.a, B.b.b);
            s_abk[threadIdx.x * 3 + 2] = make_uint4(K.a.a, K.a.b, K.b.a, K.b.b);
        }
    }
    __syncthreads();
    constexpr int last = 4 * ((COMP == C_PROCESSOR ? 3 : 1) - 1);   // first coordinate of the last logUp column (full size)
    g_cu32p first = as_global(a.is_first);
    g_cu32p cur_p[4] = {as_global(a.inter[last].ptr), as_global(a.inter[last + 1].ptr), as_global(a.inter[last + 2].ptr), as_global(a.inter[last + 3].ptr)};
    g_cu32p acc_p[4] = {as_global(a.acc[0]), as_global(a.acc[1]), as_global(a.acc[2]), as_global(a.acc[3])};
#pragma unroll 1
    for (u32 i = 0; i < 4; i++) {
        const u32 q = i * 256u + threadIdx.x, rel = base + 4u * q;
        if (rel >= n) break;
        const u32 r0 = row_first + rel;
        const uint4 a4 = s_abk[(q >> 2) * 3], b4 = s_abk[(q >> 2) * 3 + 1], k4 = s_abk[(q >> 2) * 3 + 2];
        const Q31 A = q_make(a4.x, a4.y, a4.z, a4.w), B = q_make(b4.x, b4.y, b4.z, b4.w), K = q_make(k4.x, k4.y, k4.z, k4.w);
        const uint4 t4 = ld16(first + r0);
        uint4 c4[4];
        u32 pv[4][4];                                             // [coordinate][row of the quad]
#pragma unroll
        for (int w = 0; w < 4; w++) c4[w] = ld16(cur_p[w] + r0);
        if (a.inter_prev[0]) {                                    // materialised previous-row copy (row-sharded columns)
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint4 p = ld16(as_global(a.inter_prev[w]) + r0); pv[w][0] = p.x; pv[w][1] = p.y; pv[w][2] = p.z; pv[w][3] = p.w; }
        } else {
            const u32 pr[4] = {prev_lde_row(r0, a.log_size), prev_lde_row(r0 + 1, a.log_size), prev_lde_row(r0 + 2, a.log_size), prev_lde_row(r0 + 3, a.log_size)};
            if ((pr[0] & 3u) == 0 && pr[2] == pr[0] + 2 && (pr[1] & 3u) == 1 && pr[3] == pr[1] + 2) {
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint4 pe = ld16(cur_p[w] + pr[0]), po = ld16(cur_p[w] + (pr[1] - 1));
                    pv[w][0] = pe.x; pv[w][2] = pe.z; pv[w][1] = po.y; pv[w][3] = po.w;
                }
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++) { pv[w][0] = cur_p[w][pr[0]]; pv[w][1] = cur_p[w][pr[1]]; pv[w][2] = cur_p[w][pr[2]]; pv[w][3] = cur_p[w][pr[3]]; }
            }
        }
        const u32 t[4] = {t4.x, t4.y, t4.z, t4.w};
        const u32 cv[4][4] = {{c4[0].x, c4[0].y, c4[0].z, c4[0].w}, {c4[1].x, c4[1].y, c4[1].z, c4[1].w}, {c4[2].x, c4[2].y, c4[2].z, c4[2].w}, {c4[3].x, c4[3].y, c4[3].z, c4[3].w}};
        u32 out[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const Q31 X = q_make(m_sub(cv[0][r], pv[0][r]), m_sub(cv[1][r], pv[1][r]), m_sub(cv[2][r], pv[2][r]), m_sub(cv[3][r], pv[3][r]));
            const Q31 v = q_add(q_add(A, q_mulm(B, t[r])), q_mul(K, X));
            out[0][r] = v.a.a; out[1][r] = v.a.b; out[2][r] = v.b.a; out[3][r] = v.b.b;
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint4 o = make_uint4(out[w][0], out[w][1], out[w][2], out[w][3]);
            if (!a.overwrite) { const uint4 old = ld16(acc_p[w] + r0); o = make_uint4(m_add(old.x, o.x), m_add(old.y, o.y), m_add(old.z, o.z), m_add(old.w, o.w)); }
            *reinterpret_cast<uint4*>(a.acc[w] + r0) = o;
        }
    }
}
template <int COMP>
__global__ void __launch_bounds__(256) k_constraints_block(const ConstraintArgs* __restrict__ ap) {
    __shared__ uint4 s_abk[256 * 3];
    constraints_block_body<COMP>(*ap, s_abk, blockIdx.x);
}
// Paired blocks (r06; full-domain launches only). The previous row of the last logUp column is row ^ M with M a run of HIGH bits: a quad's
// even elements find theirs in the even elements of ANOTHER quad, its odd elements in the odd elements of a third, and a 16-byte read of either
// uses half of what it fetches — the column was read three times per launch (cur + 2 x half-used: the 1.28x of profiles/r05_pmc_traffic.json).
// In half-index terms (e = row >> 1, natural index = bit_rev(e)) even rows step back by one and odd rows forward by one, and a step that only
// toggles the top bit of e stays at the same offset of the OTHER half of the column: the even rows of the upper half (top bit set) step back onto
// the even rows of the lower half at the same offset, the odd rows of the lower half step forward onto the odd rows of the upper half. So one
// workgroup takes the 4096-row block L of the lower half together with the block U = L + n/2: each lane loads a quad of L and the quad of U at the
// same offset and has, in registers, the previous rows of U's even and L's odd elements for free; only L's even and U's odd elements still fetch a
// (half-used) quad elsewhere. 2 reads of the column per launch instead of 3, bytes and values unchanged.
template <int COMP>
__device__ __forceinline__ void constraints_pair_body(const ConstraintArgs& a, uint4* __restrict__ s_abk, u32 block) {
    const u32 n = 2u << a.log_size, half = n >> 1;
    const u32 baseL = block * 4096u;
#pragma unroll 1
    for (u32 h = 0; h < 2; h++) {
        const u32 row = baseL + h * half + threadIdx.x * 16u;
        GroupEval e(a, row);
        air_eval<COMP>(e, a.el);
        const u32 dinv = a.denom_inv[row >> a.log_size];
        const Q31 A = q_mulm(e.A(), dinv), B = q_mulm(e.B(), dinv), K = q_mulm(e.K, dinv);
        uint4* o = s_abk + 768u * h + threadIdx.x * 3;
        o[0] = make_uint4(A.a.a, A.a.b, A.b.a, A.b.b); o[1] = make_uint4(B.a.a, B.a.b, B.b.a, B.b.b); o[2] = make_uint4(K.a.a, K.a.b, K.b.a, K.b.b);
    }
    __syncthreads();
    constexpr int last = 4 * ((COMP == C_PROCESSOR ? 3 : 1) - 1);
    g_cu32p first = as_global(a.is_first);
    g_cu32p cur_p[4] = {as_global(a.inter[last].ptr), as_global(a.inter[last + 1].ptr), as_global(a.inter[last + 2].ptr), as_global(a.inter[last + 3].ptr)};
    g_cu32p acc_p[4] = {as_global(a.acc[0]), as_global(a.acc[1]), as_global(a.acc[2]), as_global(a.acc[3])};
#pragma unroll 1
    for (u32 i = 0; i < 4; i++) {
        const u32 q = i * 256u + threadIdx.x;
        const u32 rL = baseL + 4u * q, rU = rL + half;
        uint4 cL[4], cU[4];
#pragma unroll
        for (int w = 0; w < 4; w++) { cL[w] = ld16(cur_p[w] + rL); cU[w] = ld16(cur_p[w] + rU); }
        const uint4 tL = ld16(first + rL), tU = ld16(first + rU);
        // the two steps that leave the pair: L's even elements (back, borrow past the top bit) and U's odd elements (forward, carry past it)
        const u32 peL = prev_lde_row(rL, a.log_size), peL2 = prev_lde_row(rL + 2, a.log_size);
        const u32 poU = prev_lde_row(rU + 1, a.log_size), poU3 = prev_lde_row(rU + 3, a.log_size);
        u32 pvL[4][4], pvU[4][4];                                  // [coordinate][row of the quad]
        const bool quad_e = (peL & 3u) == 0 && peL2 == peL + 2, quad_o = (poU & 3u) == 1 && poU3 == poU + 2;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (quad_e) { const uint4 pe = ld16(cur_p[w] + peL); pvL[w][0] = pe.x; pvL[w][2] = pe.z; }
            else { pvL[w][0] = cur_p[w][peL]; pvL[w][2] = cur_p[w][peL2]; }
            if (quad_o) { const uint4 po = ld16(cur_p[w] + (poU - 1)); pvU[w][1] = po.y; pvU[w][3] = po.w; }
            else { pvU[w][1] = cur_p[w][poU]; pvU[w][3] = cur_p[w][poU3]; }
            pvL[w][1] = cU[w].y; pvL[w][3] = cU[w].w;             // odd rows of L step forward onto the odd rows of U, same offset
            pvU[w][0] = cL[w].x; pvU[w][2] = cL[w].z;             // even rows of U step back onto the even rows of L, same offset
        }
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            const u32 r0 = h ? rU : rL;
            const uint4* sa = s_abk + 768u * h + (q >> 2) * 3;
            const uint4 a4 = sa[0], b4 = sa[1], k4 = sa[2];
            const Q31 A = q_make(a4.x, a4.y, a4.z, a4.w), B = q_make(b4.x, b4.y, b4.z, b4.w), K = q_make(k4.x, k4.y, k4.z, k4.w);
            const uint4 t4 = h ? tU : tL;
            const u32 t[4] = {t4.x, t4.y, t4.z, t4.w};
            u32 out[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                u32 x[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint4 c = h ? cU[w] : cL[w];
                    const u32 cv = r == 0 ? c.x : r == 1 ? c.y : r == 2 ? c.z : c.w;
                    x[w] = m_sub(cv, h ? pvU[w][r] : pvL[w][r]);
                }
                const Q31 v = q_add(q_add(A, q_mulm(B, t[r])), q_mul(K, q_make(x[0], x[1], x[2], x[3])));
                out[0][r] = v.a.a; out[1][r] = v.a.b; out[2][r] = v.b.a; out[3][r] = v.b.b;
            }
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint4 o = make_uint4(out[w][0], out[w][1], out[w][2], out[w][3]);
                if (!a.overwrite) { const uint4 old = ld16(acc_p[w] + r0); o = make_uint4(m_add(old.x, o.x), m_add(old.y, o.y), m_add(old.z, o.z), m_add(old.w, o.w)); }
                *reinterpret_cast<uint4*>(a.acc[w] + r0) = o;
            }
        }
    }
}
template <int COMP>
__global__ void __launch_bounds__(256) k_constraints_pair(const ConstraintArgs* __restrict__ ap) {
    __shared__ uint4 s_abk[2 * 256 * 3];
    constraints_pair_body<COMP>(*ap, s_abk, blockIdx.x);
}

// ---- all components of a proof in ONE launch ----------------------------------------------------------------------------------------
// The components are grouped into CLASSES by evaluation-domain size (= one accumulator each). A workgroup belongs to one class and
// evaluates every component of it for its rows: per-row mode sums them in registers and writes the accumulator once (no read-modify-write,
// no ordering between components needed); row-group mode runs the components one after the other on the workgroup's 4096 rows (the same
// lane owns the same rows each time, so its accumulator updates are ordered by program order).
template <int COMP>
__device__ __forceinline__ Q31 constraints_row_value(const ConstraintArgs& a, u32 row) {
    DomainEval e(a, row);
    air_eval<COMP>(e, a.el);
    return q_mulm(e.result(), a.denom_inv[row >> a.log_size]);
}
#define BF_FOR_EACH_COMPONENT(X) X(C_MEMORY) X(C_INSTRUCTION) X(C_PROGRAM) X(C_PROCESSOR) X(C_JNZ) X(C_JZ) X(C_INPUT) X(C_LEFT) X(C_MINUS) X(C_OUTPUT) X(C_PLUS) X(C_RIGHT) X(C_EOE)
__global__ void __launch_bounds__(256) k_constraints_batch(const ConstraintBatch* __restrict__ bp, const ConstraintArgs* __restrict__ args) {
    __shared__ uint4 s_abk[2 * 256 * 3];
    const ConstraintBatch& b = *bp;
    u32 ci = 0;
    while (ci + 1 < b.n_classes && b.cls[ci + 1].block0 <= blockIdx.x) ci++;      // uniform: scalar loads
    const ConstraintClass& cl = b.cls[ci];
    const u32 block = blockIdx.x - cl.block0;
    if (cl.group_rows) {
        for (u32 q = 0; q < cl.n_comps; q++) {
            const ConstraintArgs& a = args[cl.comp[q]];
            if (cl.group_rows == 32) {                     // paired blocks: 2 x 4096 rows per workgroup (full-domain launches)
                switch (cl.comp[q]) {
#define X(C) case C: constraints_pair_body<C>(a, s_abk, block); break;
                    BF_FOR_EACH_COMPONENT(X)
#undef X
                }
            } else {
                switch (cl.comp[q]) {
#define X(C) case C: constraints_block_body<C>(a, s_abk, block); break;
                    BF_FOR_EACH_COMPONENT(X)
#undef X
                }
            }
            __syncthreads();      // s_abk is reused by the next component
        }
        return;
    }
    const ConstraintArgs& a0 = args[cl.comp[0]];
    u32 row = block * blockDim.x + threadIdx.x;
    if (a0.n_rows) { if (row >= a0.n_rows) return; row += a0.row0; }
    else if (row >= (2u << a0.log_size)) return;
    Q31 sum = q_zero();
    for (u32 q = 0; q < cl.n_comps; q++) {
        const ConstraintArgs& a = args[cl.comp[q]];
        Q31 v = q_zero();
        switch (cl.comp[q]) {
#define X(C) case C: v = constraints_row_value<C>(a, row); break;
            BF_FOR_EACH_COMPONENT(X)
#undef X
        }
        sum = q_add(sum, v);
    }
    as_global(a0.acc[0])[row] = sum.a.a; as_global(a0.acc[1])[row] = sum.a.b; as_global(a0.acc[2])[row] = sum.b.a; as_global(a0.acc[3])[row] = sum.b.b;
}
// launches[k] = the parameter block of component k (all 13 present, staged at d_args); components with equal log_size form a class.
void constraint_batch_init(ConstraintBatch& b, const ConstraintLaunch* launches, u32 n) {
    b = ConstraintBatch{};
    u32 blocks = 0;
    for (u32 k = 0; k < n; k++) {
        u32 ci = 0;
        while (ci < b.n_classes && launches[b.cls[ci].comp[0]].log_size != launches[k].log_size) ci++;
        if (ci == b.n_classes) { b.n_classes++; b.cls[ci].group_rows = constraint_group_rows(launches[k], (int)k); b.cls[ci].comp[0] = k; }
        ConstraintClass& cl = b.cls[ci];
        if (constraint_group_rows(launches[k], (int)k) != cl.group_rows || launches[k].n_rows != launches[cl.comp[0]].n_rows || launches[k].row0 != launches[cl.comp[0]].row0 ||
            launches[k].acc[0] != launches[cl.comp[0]].acc[0])
            throw std::runtime_error("constraint batch: components of one size disagree on mode, row range or accumulator");
        cl.comp[cl.n_comps++] = k;
    }
    for (u32 ci = 0; ci < b.n_classes; ci++) {
        ConstraintClass& cl = b.cls[ci];
        const ConstraintLaunch& L = launches[cl.comp[0]];
        const u32 rows = L.n_rows ? L.n_rows : 2u << L.log_size;
        cl.block0 = blocks;
        blocks += cl.group_rows == 32 ? rows / 8192 : cl.group_rows ? (rows + 4095) / 4096 : (rows + 255) / 256;
    }
    b.total_blocks = blocks;
}
void eval_constraints_batch(hipStream_t stream, const ConstraintBatch* d_batch, const ConstraintBatch& h_batch, const ConstraintLaunch* d_args) {
    if (!h_batch.total_blocks) return;
    ProfScope ps(stream, "k_constraints", 0);
    hipLaunchKernelGGL(k_constraints_batch, dim3(h_batch.total_blocks), dim3(256), 0, stream, d_batch, d_args);
}

template <int COMP>
static void launch_c(hipStream_t s, const ConstraintArgs* a, u32 log_size, u32 n_rows, u32 group_rows) {
    u32 n = n_rows ? n_rows : 2u << log_size;
    ProfScope ps(s, "k_constraints", 0);
    if (group_rows == 32) hipLaunchKernelGGL(k_constraints_pair<COMP>, dim3(n / 8192), dim3(256), 0, s, a);
    else if (group_rows) hipLaunchKernelGGL(k_constraints_block<COMP>, dim3((n + 4095) / 4096), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_constraints<COMP>, dim3((n + 255) / 256), dim3(256), 0, s, a);
}

// 16 = the row-group kernel applies (every main column and every logUp column but the last stored replicated, shift >= 4, the last one
// full size), 32 = the same with two blocks per workgroup (constraints_pair_body: the launch covers the whole domain and the previous rows are
// read from the column itself), 0 = per-row kernel. BFHIP_CONSTRAINT_PAIRS=0: A/B knob (16 instead of 32; same bytes).
u32 constraint_group_rows(const ConstraintLaunch& L, int comp) {
    const u32 last = 4 * (n_logup_cols(comp) - 1);
    for (u32 j = 0; j < n_main_cols(comp); j++) if (L.trace[j].shift < LOG_N_LANES) return 0;
    for (u32 j = 0; j < last; j++) if (L.inter[j].shift < LOG_N_LANES) return 0;
    for (u32 j = last; j < last + 4; j++) if (L.inter[j].shift != 0) return 0;
    const u32 rows = L.n_rows ? L.n_rows : 2u << L.log_size;
    if (rows % 16 != 0 || L.row0 % 16 != 0) return 0;
    // below 512 workgroups the per-row kernel's wider launch wins (measured: equal at 2^20 rows). BFHIP_CONSTRAINT_GROUP_MIN_LOG: test knob
    // (the suite runs the row-group kernel on small domains too)
    u32 min_log = 21;
    if (const char* v = getenv("BFHIP_CONSTRAINT_GROUP_MIN_LOG")) min_log = (u32)atoi(v);
    if (!(min_log < 32 && rows >= (1u << min_log))) return 0;
    static const bool pairs = [] { const char* v = getenv("BFHIP_CONSTRAINT_PAIRS"); return !v || v[0] != '0'; }();
    // a pair = block L of the lower half + block U = L + rows / 2. From 2^24 rows up (8 x the row-group threshold): measured on fib19 (2^25 rows) the
    // launch drops 799 -> 729 us (profiles/r06_constraint_pairs_ab.txt); at 2^23 rows the halved number of workgroups costs what the bytes save
    return (pairs && L.n_rows == 0 && !L.inter_prev[0] && rows >= 8192 && min_log < 28 && rows >= (8u << min_log)) ? 32 : 16;
}

void eval_constraints(hipStream_t stream, int comp, const ConstraintLaunch* a, u32 log_size, u32 n_rows, u32 group_rows) {
    switch (comp) {
        case C_MEMORY: launch_c<C_MEMORY>(stream, a, log_size, n_rows, group_rows); break;
        case C_INSTRUCTION: launch_c<C_INSTRUCTION>(stream, a, log_size, n_rows, group_rows); break;
        case C_PROGRAM: launch_c<C_PROGRAM>(stream, a, log_size, n_rows, group_rows); break;
        case C_PROCESSOR: launch_c<C_PROCESSOR>(stream, a, log_size, n_rows, group_rows); break;
        case C_JNZ: launch_c<C_JNZ>(stream, a, log_size, n_rows, group_rows); break;
        case C_JZ: launch_c<C_JZ>(stream, a, log_size, n_rows, group_rows); break;
        case C_INPUT: launch_c<C_INPUT>(stream, a, log_size, n_rows, group_rows); break;
        case C_LEFT: launch_c<C_LEFT>(stream, a, log_size, n_rows, group_rows); break;
        case C_MINUS: launch_c<C_MINUS>(stream, a, log_size, n_rows, group_rows); break;
        case C_OUTPUT: launch_c<C_OUTPUT>(stream, a, log_size, n_rows, group_rows); break;
        case C_PLUS: launch_c<C_PLUS>(stream, a, log_size, n_rows, group_rows); break;
        case C_RIGHT: launch_c<C_RIGHT>(stream, a, log_size, n_rows, group_rows); break;
        default: launch_c<C_EOE>(stream, a, log_size, n_rows, group_rows); break;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// logUp interaction trace
// ------------------------------------------------------------------------------------------------------------------------------
// All components of a proof go through the SAME four launches (LogupBatch in HBM: per-component pointers and, per stage, the first
// workgroup of each component): the reference's 13 interaction_trace_evaluation calls (mod.rs:596-687) cost 4 launches, not 52.
// Stage 1 — one lane per table row: fractions num/denom of every logUp column, running sum over columns.
// Columns before the last are written row-granular (they are 16x-replicated like the main trace); the last column's
// per-row value goes to `vrow` for the coset-order prefix sum.
__device__ __forceinline__ u32 logup_item_of_block(const u32* __restrict__ blk0, u32 n) {
    u32 k = 0;
    while (k + 1 < n && blk0[k + 1] <= blockIdx.x) k++;      // uniform: scalar loads
    return k;
}

__device__ __forceinline__ Q31 logup_denominator(const Lookups& el, int rel, const u32* v) {
    const Lookup& l = rel == 0 ? el.memory : rel == 1 ? el.instruction : el.processor;
    int n = rel == 2 ? 7 : 3;
    Q31 acc = q_zero();
    for (int i = 0; i < n; i++) acc = q_add(acc, q_mulm(l.alpha_pow[i], v[i]));
    return q_sub(acc, l.z);
}

__global__ void __launch_bounds__(256) k_logup_rows(const LogupBatch* __restrict__ bp) {
    const LogupBatch& b = *bp;
    const u32 k = logup_item_of_block(b.rows_blk0, b.n);
    const LogupItem& a = b.item[k];
    u32 r = (blockIdx.x - b.rows_blk0[k]) * blockDim.x + threadIdx.x;
    if (r >= (1u << a.log_rows)) return;
    const int comp = a.comp;
    u32 v[7];
    Q31 cur = q_zero();
    if (comp == C_PROCESSOR) {
        // processor/table.rs:479-529: columns Processor(7), Instruction(3), Memory(3); numerator 1 - d
        u32 reg[8];
        for (int i = 0; i < 8; i++) reg[i] = a.cols[i][r];
        Q31 num = q_subm(q_one(), reg[7]);
        Q31 d0 = logup_denominator(b.el, 2, reg);
        v[0] = reg[1]; v[1] = reg[2]; v[2] = reg[3];
        Q31 d1 = logup_denominator(b.el, 1, v);
        v[0] = reg[0]; v[1] = reg[4]; v[2] = reg[5];
        Q31 d2 = logup_denominator(b.el, 0, v);
        // one shared inversion (Montgomery trick) — results equal the three separate inverses
        Q31 d01 = q_mul(d0, d1), inv_all = q_inv(q_mul(d01, d2));
        Q31 i2 = q_mul(inv_all, d01), i01 = q_mul(inv_all, d2);
        Q31 i0 = q_mul(i01, d1), i1 = q_mul(i01, d0);
        // (a null coordinate column is one this rank of a shard group does not keep: its owner writes it)
        cur = q_mul(num, i0);
        if (a.out_rep[0]) a.out_rep[0][r] = cur.a.a;
        if (a.out_rep[1]) a.out_rep[1][r] = cur.a.b;
        if (a.out_rep[2]) a.out_rep[2][r] = cur.b.a;
        if (a.out_rep[3]) a.out_rep[3][r] = cur.b.b;
        cur = q_add(cur, q_mul(num, i1));
        if (a.out_rep[4]) a.out_rep[4][r] = cur.a.a;
        if (a.out_rep[5]) a.out_rep[5][r] = cur.a.b;
        if (a.out_rep[6]) a.out_rep[6][r] = cur.b.a;
        if (a.out_rep[7]) a.out_rep[7][r] = cur.b.b;
        cur = q_add(cur, q_mul(num, i2));
    } else {
        int rel, dcol, mode;   // mode 0: d - 1, 1: 1 - d, 2: -1
        if (comp == C_MEMORY) { rel = 0; dcol = 3; mode = 0; }
        else if (comp == C_INSTRUCTION) { rel = 1; dcol = 3; mode = 0; }
        else if (comp == C_PROGRAM) { rel = 1; dcol = 3; mode = 1; }
        else if (comp == C_JNZ || comp == C_JZ) { rel = 2; dcol = 11; mode = 0; }
        else if (comp == C_EOE) { rel = 2; dcol = -1; mode = 2; }
        else { rel = 2; dcol = 7; mode = 0; }
        int n = rel == 2 ? 7 : 3;
        for (int i = 0; i < n; i++) v[i] = a.cols[i][r];
        Q31 den = logup_denominator(b.el, rel, v);
        Q31 num;
        if (mode == 2) num = q_neg(q_one());
        else { u32 d = a.cols[dcol][r]; num = mode == 0 ? q_subm(q_from_m(d), 1) : q_subm(q_one(), d); }
        cur = q_mul(num, q_inv(den));
    }
    a.vrow[r] = make_uint4(cur.a.a, cur.a.b, cur.b.a, cur.b.b);
}

__device__ __forceinline__ Q31 q_ld(const uint4* p, u32 i) { uint4 v = p[i]; return q_make(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void q_st(uint4* p, u32 i, Q31 q) { p[i] = make_uint4(q.a.a, q.a.b, q.b.a, q.b.b); }

// Stage 2 — inclusive scan over R = bit_reverse(row) of w[R] = v[R] + v[M-1-R]  (M rows; see DESIGN.md "coset-order prefix sum").
// Block-local scan of SCAN_TILE entries; block totals are scanned by k_scan_totals.
static constexpr u32 SCAN_TILE = 1024;
__global__ void __launch_bounds__(256) k_logup_scan_local(const LogupBatch* __restrict__ bp) {
    __shared__ uint4 s[SCAN_TILE];
    const LogupBatch& b = *bp;
    const u32 k = logup_item_of_block(b.scan_blk0, b.n);
    const LogupItem& a = b.item[k];
    const uint4* __restrict__ vrow = a.vrow; uint4* __restrict__ wloc = a.wloc; uint4* __restrict__ totals = a.totals;
    const u32 log_rows = a.log_rows, blk = blockIdx.x - b.scan_blk0[k];
    u32 M = 1u << log_rows;
    u32 base = blk * SCAN_TILE;
    for (u32 i = threadIdx.x; i < SCAN_TILE; i += blockDim.x) {
        u32 R = base + i;
        Q31 w = q_zero();
        if (R < M) w = q_add(q_ld(vrow, bit_rev(R, log_rows)), q_ld(vrow, bit_rev(M - 1 - R, log_rows)));
        q_st(s, i, w);
    }
    __syncthreads();
    // Hillis-Steele over 1024 entries, 4 per thread
    for (u32 off = 1; off < SCAN_TILE; off <<= 1) {
        Q31 t[4];
        for (u32 q = 0; q < 4; q++) { u32 i = threadIdx.x + q * 256; t[q] = i >= off ? q_add(q_ld(s, i), q_ld(s, i - off)) : q_ld(s, i); }
        __syncthreads();
        for (u32 q = 0; q < 4; q++) q_st(s, threadIdx.x + q * 256, t[q]);
        __syncthreads();
    }
    for (u32 i = threadIdx.x; i < SCAN_TILE; i += blockDim.x) if (base + i < M) wloc[base + i] = s[i];
    if (threadIdx.x == 0) totals[blk] = s[SCAN_TILE - 1];
}
// Exclusive scan of the block totals (one workgroup per component, serial over chunks of 256); totals[nb] receives the grand total.
__global__ void __launch_bounds__(256) k_scan_totals(const LogupBatch* __restrict__ bp) {
    __shared__ uint4 s[256];
    const LogupItem& a = bp->item[blockIdx.x];
    uint4* __restrict__ totals = a.totals;
    const u32 nb = a.nb;
    Q31 carry = q_zero();
    for (u32 base = 0; base < nb; base += 256) {
        u32 i = base + threadIdx.x;
        Q31 v = i < nb ? q_ld(totals, i) : q_zero();
        q_st(s, threadIdx.x, v);
        __syncthreads();
        for (u32 off = 1; off < 256; off <<= 1) {
            Q31 t = threadIdx.x >= off ? q_add(q_ld(s, threadIdx.x), q_ld(s, threadIdx.x - off)) : q_ld(s, threadIdx.x);
            __syncthreads();
            q_st(s, threadIdx.x, t);
            __syncthreads();
        }
        Q31 incl = q_ld(s, threadIdx.x);
        Q31 excl = q_add(carry, q_sub(incl, v));
        Q31 chunk_total = q_ld(s, 255);
        __syncthreads();
        if (i < nb) q_st(totals, i, excl);
        carry = q_add(carry, chunk_total);
    }
    // the grand total W and the component's claimed sum (LogupTraceGenerator::finalize_last returns the last cell of the coset-order prefix sum = 8 W)
    if (threadIdx.x == 0) { q_st(totals, nb, carry); q_st(a.claimed, 0, q_mulm(carry, 8)); }
}
// Stage 3 — write the last logUp column (4 coordinate columns of N = 16 M cells) and the claimed sum.
// cell s = 16 r + l, R = bit_reverse(r), L = bit_reverse4(l):
//   L <  8: S = L*Wtot + W[R] - v[M-1-R]     (even coset position 2q, q = L*M + R)
//   L >= 8: S = (15-L)*Wtot + W[M-1-R]       (odd coset position)
__global__ void __launch_bounds__(256) k_logup_last(const LogupBatch* __restrict__ bp) {
    const LogupBatch& b = *bp;
    const u32 k = logup_item_of_block(b.last_blk0, b.n);
    const LogupItem& a = b.item[k];
    const uint4* __restrict__ vrow = a.vrow; const uint4* __restrict__ wloc = a.wloc; const uint4* __restrict__ totals = a.totals;
    const u32 log_rows = a.log_rows, nb = a.nb;
    u32 s = (blockIdx.x - b.last_blk0[k]) * blockDim.x + threadIdx.x;
    u32 M = 1u << log_rows;
    if (s >= 16 * M) return;
    u32 r = s >> 4, l = s & 15;
    u32 R = bit_rev(r, log_rows), L = bit_rev(l, 4);
    Q31 wtot = q_ld(totals, nb);
    Q31 res;
    if (L < 8) {
        Q31 W = q_add(q_ld(wloc, R), q_ld(totals, R / SCAN_TILE));
        res = q_sub(q_add(q_mulm(wtot, L), W), q_ld(vrow, bit_rev(M - 1 - R, log_rows)));
    } else {
        u32 X = M - 1 - R;
        Q31 W = q_add(q_ld(wloc, X), q_ld(totals, X / SCAN_TILE));
        res = q_add(q_mulm(wtot, 15 - L), W);
    }
    // a null coordinate column is not written: in a shard group only the rank that transforms a coordinate column (its owner) needs it
    if (a.out_last[0]) a.out_last[0][s] = res.a.a;
    if (a.out_last[1]) a.out_last[1][s] = res.a.b;
    if (a.out_last[2]) a.out_last[2][s] = res.b.a;
    if (a.out_last[3]) a.out_last[3][s] = res.b.b;
}

void logup_batch_init(LogupBatch& b, const Lookups& el, const LogupLaunch* L, u32 n) {
    if (n > N_COMPONENTS) throw std::runtime_error("logup batch: too many components");
    b = LogupBatch{};
    b.el = el; b.n = n;
    u32 rows = 0, scan = 0, last = 0;
    for (u32 k = 0; k < n; k++) {
        LogupItem& it = b.item[k];
        for (int i = 0; i < 13; i++) it.cols[i] = L[k].cols[i];
        for (int i = 0; i < 8; i++) it.out_rep[i] = L[k].out_rep[i];
        for (int i = 0; i < 4; i++) it.out_last[i] = L[k].out_last[i];
        it.vrow = (uint4*)L[k].vrow; it.wloc = (uint4*)L[k].wloc; it.totals = (uint4*)L[k].totals; it.claimed = (uint4*)L[k].claimed;
        it.log_rows = L[k].log_rows; it.comp = L[k].comp;
        const u32 M = 1u << it.log_rows;
        it.nb = (M + SCAN_TILE - 1) / SCAN_TILE;
        b.rows_blk0[k] = rows; rows += (M + 255) / 256;
        b.scan_blk0[k] = scan; scan += it.nb;
        // no workgroups for a component none of whose four coordinate columns this rank keeps
        const bool any_last = it.out_last[0] || it.out_last[1] || it.out_last[2] || it.out_last[3];
        b.last_blk0[k] = last; last += any_last ? (16 * M + 255) / 256 : 0u;
    }
    b.rows_blk0[n] = rows; b.scan_blk0[n] = scan; b.last_blk0[n] = last;
}
void logup_batch_run(hipStream_t stream, const LogupBatch* d_b, const LogupBatch& h) {
    if (!h.n) return;
    hipLaunchKernelGGL(k_logup_rows, dim3(h.rows_blk0[h.n]), dim3(256), 0, stream, d_b);
    hipLaunchKernelGGL(k_logup_scan_local, dim3(h.scan_blk0[h.n]), dim3(256), 0, stream, d_b);
    hipLaunchKernelGGL(k_scan_totals, dim3(h.n), dim3(256), 0, stream, d_b);
    if (h.last_blk0[h.n]) hipLaunchKernelGGL(k_logup_last, dim3(h.last_blk0[h.n]), dim3(256), 0, stream, d_b);
}

// Broadcast upload helper (a14): rows -> 16 consecutive cells. Only used by the C-ABI when a caller wants the full-size column.
__global__ void k_broadcast16(const u32* __restrict__ rows, u32* __restrict__ out, u32 n_cells) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) out[i] = rows[i >> 4];
}
void broadcast16(hipStream_t stream, const u32* d_rows, u32* d_out, u32 n_cells) {
    hipLaunchKernelGGL(k_broadcast16, dim3((n_cells + 255) / 256), dim3(256), 0, stream, d_rows, d_out, n_cells);
}

}  // namespace bf
