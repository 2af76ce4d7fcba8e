// CPU sanitizer harness for the product's host-side code (VM, table builders, JSON reader, verifier) — the GPU half cannot run under
// AddressSanitizer on this pool, the host half can. Built by tests/test_host_sanitize.py with g++ -fsanitize=address,undefined.
//   host_sanitize run    <code-file> <input-file>          : compile + execute + build the 13 tables, print step count
//   host_sanitize verify <proof.json> <log_max_rows>       : parse and verify, print "ok" or the rejection reason
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
