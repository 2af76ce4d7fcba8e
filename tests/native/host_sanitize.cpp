// CPU sanitizer harness for the product's host-side code (VM, table builders, JSON reader, verifier) — the GPU half cannot run under
// AddressSanitizer on this pool, the host half can. Built by tests/test_host_sanitize.py with g++ -fsanitize=address,undefined.
//   host_sanitize run    <code-file> <input-file>          : compile + execute + build the 13 tables, print step count
//   host_sanitize verify <proof.json> <log_max_rows>       : parse and verify, print "ok" or the rejection reason
//   host_sanitize field  <count> <seed>                    : M31 products (plain and by a doubled constant), add, sub against wide arithmetic
#include "../../stwo-brainfuck_amd/csrc/host/verifier.h"
#include <cstdio>
#include <fstream>
#include <sstream>

using namespace bf;

static std::string slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::string mode = argv[1];
    try {
        if (mode == "run") {
            std::string code = slurp(argv[2]), in = slurp(argv[3]);
            std::vector<u32> ins = compile(code.c_str());
            Machine m(ins, std::vector<u8>(in.begin(), in.end()));
            m.execute();
            std::vector<Table> tables = build_tables(m.trace, ins);
            size_t cells = 0;
            for (auto& t : tables) cells += t.cols.size() * t.n_rows;
            printf("steps %zu tables %zu cells %zu\n", m.trace.size(), tables.size(), cells);
            return 0;
        }
        if (mode == "field") {
            // m31.h: the product by a doubled constant (the FFT butterflies' twiddles, r04) against the plain product and against 128-bit
            // arithmetic, on the edge values and on <count> pseudo-random pairs; the modular add/sub against their definitions
            const u64 count = strtoull(argv[2], nullptr, 10);
            const u32 edges[] = {0u, 1u, 2u, 3u, 0xFFFFu, 0x10000u, 0x3FFFFFFFu, 0x40000000u, 0x40000001u, P31 - 2, P31 - 1};
            u64 checked = 0, x = strtoull(argv[3], nullptr, 10) | 1;
            auto check = [&](u32 a, u32 w) {
                const u32 want = (u32)(((unsigned __int128)a * w) % P31);
                if (m_mul(a, w) != want || m_mul_pre2(a, 2u * w) != want) { printf("mismatch a=%u w=%u\n", a, w); exit(1); }
                // the product by a power of two as a 31-bit rotation (the inverse transforms' scaling by 2^-n, r05) against 128-bit arithmetic
                { const u32 k = w % 31u; if (m_mul_pow2(a, k) != (u32)((((unsigned __int128)a) << k) % P31)) { printf("m_mul_pow2 mismatch a=%u k=%u\n", a, k); exit(1); } }
                if (m_add(a, w) != (u32)(((u64)a + w) % P31) || m_sub(a, w) != (u32)(((u64)a + P31 - w) % P31)) { printf("add/sub mismatch a=%u w=%u\n", a, w); exit(1); }
                checked++;
            };
            for (u32 a : edges) for (u32 w : edges) check(a, w);
            for (u64 i = 0; i < count; i++) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                check((u32)(x % P31), (u32)((x >> 32) % P31));
            }
            // QM31 product by a constant of the kernel (the FRI folds, r04): the 4 x 4 matrix form with one u64 accumulator per component
            // against q_mul, on edge values in every position and on count / 8 pseudo-random pairs
            auto qcheck = [&](Q31 a, Q31 y) {
                const Q31 want = q_mul(a, y), got = q_mul_const(a, q_const(y));
                if (!q_eq(want, got)) { printf("q_mul_const mismatch\n"); exit(1); }
            };
            const u32 qe[] = {0u, 1u, 2u, P31 - 2, P31 - 1, 0x40000000u};
            for (u32 e0 : qe) for (u32 e1 : qe) for (u32 e2 : qe) for (u32 e3 : qe) { qcheck(q_make(e0, e1, e2, e3), q_make(e3, e2, e1, e0)); qcheck(q_make(P31 - 1, P31 - 1, P31 - 1, P31 - 1), q_make(e0, e1, e2, e3)); }
            for (u64 i = 0; i < count / 8; i++) {
                u32 w[8];
                for (int j = 0; j < 8; j++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w[j] = (u32)(x % P31); }
                qcheck(q_make(w[0], w[1], w[2], w[3]), q_make(w[4], w[5], w[6], w[7]));
            }
            if (m_red4(~0ull) != (u32)(((unsigned __int128)~0ull) % P31) || m_red4(0) != 0 || m_red4(P31) != 0) { printf("m_red4 mismatch\n"); exit(1); }
            printf("ok %llu\n", (unsigned long long)checked);
            return 0;
        }
        if (mode == "verify") {
            std::string js = slurp(argv[2]);
            std::string reason;
            try { BrainfuckProof bp = proof_from_json(js.data(), js.size()); reason = verify_brainfuck(bp, (u32)atoi(argv[3])); }
            catch (const std::exception& e) { reason = std::string("malformed: ") + e.what(); }
            printf("%s\n", reason.empty() ? "ok" : reason.c_str());
            return 0;
        }
    } catch (const std::exception& e) { printf("error: %s\n", e.what()); return 1; }
    return 2;
}
