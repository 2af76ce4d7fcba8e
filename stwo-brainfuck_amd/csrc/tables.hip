// Witness/trace generation on the GPU — SURVEY.md §8(f)1, the first "next" row after the hot path: the 13 `XTable::from(&vm_trace)`
// builders of crates/brainfuck_prover/src/brainfuck_air/mod.rs:511-547 (table.rs of every component), producing the row-granular
// columns the prover consumes. The VM itself stays on the host (sequential interpreter); its register trace is uploaded once (7 x u32
// per step, SoA) and everything downstream — the (mp, clk) and (ip, clk) sorts, clk-gap dummy fill, padding to a power of two,
// current/next pairing, per-opcode selection — runs here. Bit-exact against the host builders (tests/test_gpu_tables.py).
//
// Sorting uses rocPRIM's stable device radix sort on 64-bit keys; everything else is hand-written (scans, gap-fill by binary search
// from the output side so that one long clk gap does not serialise on one lane).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "kernels.h"
#include "ctx.h"

namespace bf {

struct TraceSoA { const u32 *clk, *ip, *ci, *ni, *mp, *mv, *mvi; u32 n; };

// ---- generic u32 exclusive scan (block-local + totals) -----------------------------------------------------------------------------
static constexpr u32 SC_TILE = 2048;   // 256 lanes x 8
__global__ void __launch_bounds__(256) k_scan_u32_local(const u32* __restrict__ in, u32* __restrict__ out, u32* __restrict__ totals, u32 n) {
    __shared__ u32 s[256];
    u32 base = blockIdx.x * SC_TILE + threadIdx.x * 8;
    u32 v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : 0; sum += v[k]; }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    u32 excl = s[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < 8; k++) { if (base + k < n) out[base + k] = excl; excl += v[k]; }
    if (threadIdx.x == 255) totals[blockIdx.x] = s[255];
}
__global__ void __launch_bounds__(256) k_scan_u32_totals(u32* __restrict__ totals, u32 nb) {   // exclusive, in place; totals[nb] = grand total
    __shared__ u32 s[256];
    u32 carry = 0;
    for (u32 b0 = 0; b0 < nb; b0 += 256) {
        u32 i = b0 + threadIdx.x;
        u32 v = i < nb ? totals[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (u32 off = 1; off < 256; off <<= 1) {
            u32 t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        u32 incl = s[threadIdx.x], tot = s[255];
        __syncthreads();
        if (i < nb) totals[i] = carry + incl - v;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[nb] = carry;
}
__global__ void __launch_bounds__(256) k_scan_u32_add(u32* __restrict__ out, const u32* __restrict__ totals, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += totals[i / SC_TILE];
}
// out[i] = sum_{j<i} in[j]; returns a device pointer to the grand total (inside `totals`, which needs n/2048 + 2 words)
static const u32* exclusive_scan_u32(hipStream_t s, const u32* in, u32* out, u32* totals, u32 n) {
    u32 nb = (n + SC_TILE - 1) / SC_TILE;
    hipLaunchKernelGGL(k_scan_u32_local, dim3(nb), dim3(256), 0, s, in, out, totals, n);
    hipLaunchKernelGGL(k_scan_u32_totals, dim3(1), dim3(256), 0, s, totals, nb);
    hipLaunchKernelGGL(k_scan_u32_add, dim3((n + 255) / 256), dim3(256), 0, s, out, totals, n);
    return totals + nb;
}

// ---- processor table: processor/table.rs:255-265 (entries), :241-253 (pad), :117-145 (pairing) ------------------------------------------
__global__ void __launch_bounds__(256) k_processor_table(TraceSoA t, u32 rows, u32* c0, u32* c1, u32* c2, u32* c3, u32* c4, u32* c5, u32* c6, u32* c7, u32* c8) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    u32 last_clk = t.clk[t.n - 1], last_ip = t.ip[t.n - 1];
    auto clk_at = [&](u32 i) { return i < t.n ? t.clk[i] : m_add(last_clk, (i - t.n + 1) % P31); };
    if (r < t.n) { c0[r] = t.clk[r]; c1[r] = t.ip[r]; c2[r] = t.ci[r]; c3[r] = t.ni[r]; c4[r] = t.mp[r]; c5[r] = t.mv[r]; c6[r] = t.mvi[r]; c7[r] = 0; }
    else { c0[r] = clk_at(r); c1[r] = last_ip; c2[r] = 0; c3[r] = 0; c4[r] = 0; c5[r] = 0; c6[r] = 0; c7[r] = 1; }
    c8[r] = r + 1 < rows ? clk_at(r + 1) : m_add(clk_at(rows - 1), 1);
}

// ---- per-opcode sub tables: instructions/table.rs:310-328, :293-307, :134-161; jump/table.rs:280-297, :264-277, :122-146, :191-208 ----
__global__ void __launch_bounds__(256) k_opcode_flags(TraceSoA t, u32 opcode, u32* __restrict__ flags) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < t.n) flags[k] = (k + 1 < t.n && t.ci[k] == opcode) ? 1u : 0u;
}
struct SubOut { u32* c[13]; };
// count = number of selected steps (device pointer); rows = padded row count (host knew count already)
__global__ void __launch_bounds__(256) k_sub_table(TraceSoA t, const u32* __restrict__ flags, const u32* __restrict__ pos, u32 count, u32 rows, int is_jump, SubOut o,
                                                   const u32* __restrict__ last_sel) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    // real rows: scatter from the selected steps
    if (k < t.n && flags[k]) {
        u32 j = pos[k];
        u32 a = k, b = k + 1;
        if (!is_jump) {
            o.c[0][j] = t.clk[a]; o.c[1][j] = t.ip[a]; o.c[2][j] = t.ci[a]; o.c[3][j] = t.ni[a]; o.c[4][j] = t.mp[a]; o.c[5][j] = t.mv[a]; o.c[6][j] = t.mvi[a];
            o.c[7][j] = 0; o.c[8][j] = t.ip[b]; o.c[9][j] = t.mp[b]; o.c[10][j] = t.mv[b];
        } else {
            o.c[0][j] = t.clk[a]; o.c[1][j] = t.ip[a]; o.c[2][j] = t.ci[a]; o.c[3][j] = t.ni[a]; o.c[4][j] = t.mp[a]; o.c[5][j] = t.mv[a]; o.c[6][j] = t.mvi[a];
            o.c[7][j] = t.clk[b]; o.c[8][j] = t.ip[b]; o.c[9][j] = t.mp[b]; o.c[10][j] = t.mv[b]; o.c[11][j] = 0;
            o.c[12][j] = m_sub(1, m_mul(t.mv[a], t.mvi[a]));
        }
    }
    // dummy rows j in [count, rows): entries 2j, 2j+1 are dummies (last_clk + (i - 2 count), last_ip); the lone-entry case (count == 0,
    // one padded entry) pairs with dummy(clk + 1, ip)
    if (k < rows && k >= count) {
        u32 j = k;
        u32 last_clk = 0, last_ip = 0;
        if (count) { u32 s = *last_sel + 1; last_clk = t.clk[s]; last_ip = t.ip[s]; }   // last real entry = register after the last selected step
        u32 e0 = 2 * j - 2 * count;
        u32 clk0 = m_add(last_clk, e0 % P31), clk1 = m_add(clk0, 1);
        int ncol = is_jump ? 13 : 11;
        for (int c = 0; c < ncol; c++) o.c[c][j] = 0;
        o.c[0][j] = clk0; o.c[1][j] = last_ip;
        if (!is_jump) { o.c[7][j] = 1; o.c[8][j] = last_ip; }
        else { o.c[7][j] = clk1; o.c[8][j] = last_ip; o.c[11][j] = 1; o.c[12][j] = 1; }
    }
}
// index of the last selected step (for the padding rule): max over flagged k — computed with one atomicMax
__global__ void __launch_bounds__(256) k_last_selected(const u32* __restrict__ flags, u32 n, u32* __restrict__ out) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && flags[k]) atomicMax(out, k);
}

// ---- instruction table: instruction/table.rs:250-284 (program ∪ trace, stable sort by (ip, clk)), :239-248 (pad), :116-145 (pairing) -------
__global__ void __launch_bounds__(256) k_make_keys(const u32* __restrict__ hi, const u32* __restrict__ lo, u64* __restrict__ keys, u32* __restrict__ vals, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ((u64)hi[i] << 32) | lo[i]; vals[i] = i; }
}
// entries[(ip, ci, ni, d)] of size rows + 1 (the extra one is the pairing dummy)
__global__ void __launch_bounds__(256) k_instruction_entries(TraceSoA t, const u32* __restrict__ order, const u64* __restrict__ sorted_keys, const u32* __restrict__ code, u32 L,
                                                             u32 rows, u32* e_ip, u32* e_ci, u32* e_ni, u32* e_d) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 E = t.n + L;
    if (i < t.n) {                       // sorted trace row i: preceded by the program rows with ip' <= ip
        u32 k = order[i];
        u32 ip = t.ip[k];
        u32 p = i + (ip + 1 < L ? ip + 1 : L);
        e_ip[p] = ip; e_ci[p] = t.ci[k]; e_ni[p] = t.ni[k]; e_d[p] = 0;
    } else if (i < E) {                  // program row a: preceded by the trace rows with ip < a (lower bound in the sorted keys)
        u32 a = i - t.n;
        u64 key = (u64)a << 32;
        u32 lo = 0, hi = t.n;
        while (lo < hi) { u32 mid = (lo + hi) >> 1; if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid; }
        u32 p = a + lo;
        e_ip[p] = a; e_ci[p] = code[a]; e_ni[p] = a + 1 < L ? code[a + 1] : 0; e_d[p] = 0;
    }
}
// last entry's ip: the larger of (last sorted trace row, program row L-1) in the merged order = entry E-1
__global__ void k_instruction_pad(TraceSoA t, const u32* __restrict__ order, u32 L, u32 rows, u32* e_ip, u32* e_ci, u32* e_ni, u32* e_d) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 E = t.n + L;
    if (i + E > rows) return;           // pad positions E .. rows (inclusive: rows is the pairing dummy)
    u32 last_trace_ip = t.ip[order[t.n - 1]];
    // entry E-1 is the last sorted trace row when its ip >= L-1 (program row L-1 precedes trace rows of equal ip), else program row L-1
    u32 last_ip = (L == 0 || last_trace_ip >= L - 1) ? last_trace_ip : L - 1;
    u32 p = E + i;
    e_ip[p] = last_ip; e_ci[p] = 0; e_ni[p] = 0; e_d[p] = 1;
}
__global__ void __launch_bounds__(256) k_pair4(const u32* __restrict__ a0, const u32* __restrict__ a1, const u32* __restrict__ a2, const u32* __restrict__ a3, u32 rows,
                                               u32* o0, u32* o1, u32* o2, u32* o3, u32* o4, u32* o5, u32* o6, u32* o7) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    o0[r] = a0[r]; o1[r] = a1[r]; o2[r] = a2[r]; o3[r] = a3[r];
    o4[r] = a0[r + 1]; o5[r] = a1[r + 1]; o6[r] = a2[r + 1]; o7[r] = a3[r + 1];
}

// ---- memory table: memory/table.rs:249-251 (sort), :259-283 (clk-gap fill), :291-303 (pad), :121-151 (pairing) --------------------------------
// count[i] = 1 + number of dummies inserted before sorted entry i
__global__ void __launch_bounds__(256) k_memory_counts(TraceSoA t, const u32* __restrict__ order, u32* __restrict__ counts, unsigned long long* __restrict__ total64) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (i < t.n) {
        c = 1;
        if (i > 0) {
            u32 k = order[i], kp = order[i - 1];
            u32 next_clk = m_add(t.clk[kp], 1);
            if (t.mp[k] == t.mp[kp] && t.clk[k] > next_clk) c += t.clk[k] - next_clk;
        }
        counts[i] = c;
    }
    // exact row total in 64 bits (the u32 prefix sums wrap on register rows that are not a VM trace: bfhip_trace_create_from_registers
    // takes caller data) — one atomic per wave
    unsigned long long s = c;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total64, s);
}
// entries (clk, mp, mv, d) of size rows + 1, produced from the output side: entry j belongs to the sorted trace row i with pos[i] <= j < pos[i] + count[i]
__global__ void __launch_bounds__(256) k_memory_entries(TraceSoA t, const u32* __restrict__ order, const u32* __restrict__ pos, const u32* __restrict__ total_ptr, u32 rows,
                                                        u32* e_clk, u32* e_mp, u32* e_mv, u32* e_d) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > rows) return;
    u32 M = *total_ptr;
    if (j >= M) {                      // padding (and the pairing dummy at j == rows): last.clk + (j - M + 1), last.mp, last.mv
        u32 kl = order[t.n - 1];
        e_clk[j] = m_add(t.clk[kl], (j - M + 1) % P31); e_mp[j] = t.mp[kl]; e_mv[j] = t.mv[kl]; e_d[j] = 1;
        return;
    }
    u32 lo = 0, hi = t.n;             // largest i with pos[i] <= j
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (pos[mid] <= j) lo = mid; else hi = mid; }
    u32 i = lo, k = order[i];
    u32 nxt = i + 1 < t.n ? pos[i + 1] : M;
    if (j == nxt - 1) { e_clk[j] = t.clk[k]; e_mp[j] = t.mp[k]; e_mv[j] = t.mv[k]; e_d[j] = 0; }
    else { u32 kp = order[i - 1]; e_clk[j] = m_add(t.clk[kp], (1 + (j - pos[i])) % P31); e_mp[j] = t.mp[kp]; e_mv[j] = t.mv[kp]; e_d[j] = 1; }
}

static u32 next_pow2_u32(u32 x) { u32 p = 1; while (p < x) p <<= 1; return p; }
static u32 log2_u32(u32 x) { u32 l = 0; while ((1u << l) < x) l++; return l; }

// Builds all 13 row-granular tables on the device. `alloc(words)` provides output storage (stable device memory).
// cols_out[k][j] = device pointer of column j of component k; log_sizes_out[k] = log2(rows) + 4.
void build_tables_device(Ctx& c, const std::vector<u32> trace7_soa[7], u32 n, const std::vector<u32>& code, const std::function<u32*(size_t)>& alloc,
                         std::vector<std::vector<u32*>>& cols_out, u32 log_sizes_out[13]) {
    hipStream_t s = c.stream;
    if (n == 0) throw HipError("EmptyTrace");
    c.stage_checkpoint();
    // scratch from the arena (only the outputs need to outlive the proof when the caller says so)
    auto tmp_u32 = [&](size_t words) { return c.alloc_u32(words); };
    u32* d_tr[7];
    for (int k = 0; k < 7; k++) { d_tr[k] = tmp_u32(n); BF_HIP(hipMemcpyAsync(d_tr[k], trace7_soa[k].data(), n * sizeof(u32), hipMemcpyHostToDevice, s)); }
    TraceSoA t{d_tr[0], d_tr[1], d_tr[2], d_tr[3], d_tr[4], d_tr[5], d_tr[6], n};
    u32 L = (u32)code.size();
    u32* d_code = tmp_u32(L + 1);
    BF_HIP(hipMemcpyAsync(d_code, code.data(), L * sizeof(u32), hipMemcpyHostToDevice, s));
    cols_out.assign(13, {});

    // ---- counts needed on the host first (they fix every table's size): 8 opcode counts + memory row count ---------------------
    static const u32 OPS[8] = {OP_JNZ, OP_JZ, OP_READCHAR, OP_LEFT, OP_MINUS, OP_PUTCHAR, OP_PLUS, OP_RIGHT};   // components 4..11
    u32 nb = (n + SC_TILE - 1) / SC_TILE;
    u32* flags[8]; u32* pos[8]; u32* tot[8]; u32* last_sel = tmp_u32(8);
    BF_HIP(hipMemsetAsync(last_sel, 0, 8 * sizeof(u32), s));
    const u32* d_count[8];
    for (int q = 0; q < 8; q++) {
        flags[q] = tmp_u32(n); pos[q] = tmp_u32(n); tot[q] = tmp_u32(nb + 2);
        hipLaunchKernelGGL(k_opcode_flags, dim3((n + 255) / 256), dim3(256), 0, s, t, OPS[q], flags[q]);
        d_count[q] = exclusive_scan_u32(s, flags[q], pos[q], tot[q], n);
        hipLaunchKernelGGL(k_last_selected, dim3((n + 255) / 256), dim3(256), 0, s, flags[q], n, last_sel + q);
    }
    // memory: sort by (mp, clk)
    u64* keys = (u64*)c.arena.alloc(sizeof(u64) * n); u64* keys_sorted = (u64*)c.arena.alloc(sizeof(u64) * n);
    u32* vals = tmp_u32(n); u32* mem_order = tmp_u32(n); u32* ins_order = tmp_u32(n);
    u64* ins_keys_sorted = (u64*)c.arena.alloc(sizeof(u64) * n);
    size_t sort_tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_tmp_bytes, keys, keys_sorted, vals, mem_order, n, 0, 64, s);
    void* sort_tmp = c.arena.alloc(sort_tmp_bytes + 256);
    hipLaunchKernelGGL(k_make_keys, dim3((n + 255) / 256), dim3(256), 0, s, t.mp, t.clk, keys, vals, n);
    BF_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys, keys_sorted, vals, mem_order, n, 0, 64, s));
    u32* mem_counts = tmp_u32(n); u32* mem_pos = tmp_u32(n); u32* mem_tot = tmp_u32(nb + 2);
    unsigned long long* d_mem_total64 = (unsigned long long*)c.arena.alloc(256);
    BF_HIP(hipMemsetAsync(d_mem_total64, 0, 8, s));
    hipLaunchKernelGGL(k_memory_counts, dim3((n + 255) / 256), dim3(256), 0, s, t, mem_order, mem_counts, d_mem_total64);
    const u32* d_mem_total = exclusive_scan_u32(s, mem_counts, mem_pos, mem_tot, n);
    // instruction: sort by (ip, clk)
    hipLaunchKernelGGL(k_make_keys, dim3((n + 255) / 256), dim3(256), 0, s, t.ip, t.clk, keys, vals, n);
    BF_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys, ins_keys_sorted, vals, ins_order, n, 0, 64, s));
    // bring the 9 counts to the host
    u32 h_counts[9];
    for (int q = 0; q < 8; q++) BF_HIP(hipMemcpyAsync(&h_counts[q], d_count[q], 4, hipMemcpyDeviceToHost, s));
    BF_HIP(hipMemcpyAsync(&h_counts[8], d_mem_total, 4, hipMemcpyDeviceToHost, s));
    unsigned long long h_mem_total64 = 0;
    BF_HIP(hipMemcpyAsync(&h_mem_total64, d_mem_total64, 8, hipMemcpyDeviceToHost, s));
    c.sync();
    if (h_mem_total64 > (1ull << 28)) throw HipError("the Memory table would have more than 2^28 rows (2^32 domain rows): not a provable trace");

    auto make_cols = [&](int comp, u32 ncols, u32 rows) {
        cols_out[comp].resize(ncols);
        for (u32 j = 0; j < ncols; j++) cols_out[comp][j] = alloc(rows);
        log_sizes_out[comp] = log2_u32(rows) + 4;
    };
    // ---- memory (component 0) ----------------------------------------------------------------------------------------------------
    {
        u32 M = h_counts[8], rows = next_pow2_u32(M);
        u32* e[4]; for (auto& p : e) p = tmp_u32(rows + 1);
        hipLaunchKernelGGL(k_memory_entries, dim3((rows + 1 + 255) / 256), dim3(256), 0, s, t, mem_order, mem_pos, d_mem_total, rows, e[0], e[1], e[2], e[3]);
        make_cols(C_MEMORY, 8, rows);
        auto& o = cols_out[C_MEMORY];
        hipLaunchKernelGGL(k_pair4, dim3((rows + 255) / 256), dim3(256), 0, s, e[0], e[1], e[2], e[3], rows, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    }
    // ---- instruction (component 1) ---------------------------------------------------------------------------------------------------
    {
        u32 E = n + L, rows = next_pow2_u32(E);
        u32* e[4]; for (auto& p : e) p = tmp_u32(rows + 1);
        hipLaunchKernelGGL(k_instruction_entries, dim3((E + 255) / 256), dim3(256), 0, s, t, ins_order, ins_keys_sorted, d_code, L, rows, e[0], e[1], e[2], e[3]);
        u32 npad = rows + 1 - E;
        hipLaunchKernelGGL(k_instruction_pad, dim3((npad + 255) / 256), dim3(256), 0, s, t, ins_order, L, rows, e[0], e[1], e[2], e[3]);
        make_cols(C_INSTRUCTION, 8, rows);
        auto& o = cols_out[C_INSTRUCTION];
        hipLaunchKernelGGL(k_pair4, dim3((rows + 255) / 256), dim3(256), 0, s, e[0], e[1], e[2], e[3], rows, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
    }
    // ---- program (component 2): a few hundred rows, built on the host (program/table.rs:111-141, pad :62-71) -------------------------------
    {
        if (L == 0) throw HipError("EmptyTrace");
        u32 rows = next_pow2_u32(L);
        std::vector<u32> h[4];
        for (auto& v : h) v.assign(rows, 0);
        for (u32 r = 0; r < rows; r++) {
            if (r < L) { h[0][r] = r; h[1][r] = code[r]; h[2][r] = r + 1 < L ? code[r + 1] : 0; h[3][r] = 0; }
            else { h[0][r] = L - 1; h[3][r] = 1; }
        }
        make_cols(C_PROGRAM, 4, rows);
        // small programs go through the pinned staging ring (no pageable copy on the timeline); a program too large for it is copied directly
        const bool direct = size_t(rows) * sizeof(u32) * 4 > c.stage_bytes / 8;
        for (int j = 0; j < 4; j++) {
            if (direct) { BF_HIP(hipMemcpyAsync(cols_out[C_PROGRAM][j], h[j].data(), rows * sizeof(u32), hipMemcpyHostToDevice, s)); BF_HIP(hipStreamSynchronize(s)); }
            else { const u32* st = c.stage(h[j].data(), rows); BF_HIP(hipMemcpyAsync(cols_out[C_PROGRAM][j], st, rows * sizeof(u32), hipMemcpyDeviceToDevice, s)); }
        }
    }
    // ---- processor (component 3) --------------------------------------------------------------------------------------------------------
    {
        u32 rows = next_pow2_u32(n);
        make_cols(C_PROCESSOR, 9, rows);
        auto& o = cols_out[C_PROCESSOR];
        hipLaunchKernelGGL(k_processor_table, dim3((rows + 255) / 256), dim3(256), 0, s, t, rows, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
    }
    // ---- the 8 per-opcode tables (components 4..11) ----------------------------------------------------------------------------------------
    for (int q = 0; q < 8; q++) {
        int comp = C_JNZ + q;
        int is_jump = q < 2;
        u32 count = h_counts[q];
        u32 entries = next_pow2_u32(2 * count);          // usize::next_power_of_two(0) == 1
        u32 rows = (entries + 1) / 2;
        make_cols(comp, is_jump ? 13 : 11, rows);
        SubOut so{};
        for (size_t j = 0; j < cols_out[comp].size(); j++) so.c[j] = cols_out[comp][j];
        u32 span = n > rows ? n : rows;
        hipLaunchKernelGGL(k_sub_table, dim3((span + 255) / 256), dim3(256), 0, s, t, flags[q], pos[q], count, rows, is_jump, so, last_sel + q);
    }
    // ---- end of execution (component 12): exactly one row with ci == 0 (end_of_execution/table.rs:71-111) ------------------------------------
    {
        u32 hits = 0, at = 0;
        for (u32 i = 0; i < n; i++) if (trace7_soa[2][i] == 0) { hits++; at = i; }
        if (hits != 1) throw HipError("InvalidEndOfExecution");
        make_cols(C_EOE, 7, 1);
        u32 v[7];
        for (int j = 0; j < 7; j++) v[j] = trace7_soa[j][at];
        const u32* st = c.stage(v, 7);
        for (int j = 0; j < 7; j++) BF_HIP(hipMemcpyAsync(cols_out[C_EOE][j], st + j, sizeof(u32), hipMemcpyDeviceToDevice, s));
    }
    BF_HIP(hipGetLastError());
}

}  // namespace bf
