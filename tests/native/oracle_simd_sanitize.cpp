// Test infrastructure: the CPU oracle's scalar and SIMD (oracle/simd_port.cpp, AVX-512) provers on one program, built with -fsanitize=address,undefined by
// tests/test_oracle_simd.py — the vector loads / gathers / scatters and the parallel chunked loops of the SIMD mode must stay inside their buffers, and the two
// proofs must be the same bytes. Usage: oracle_simd_sanitize <program.bf> <log_max_rows> [input file]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <fstream>
#include <sstream>
extern "C" {
int orc_set_simd(int); int orc_set_threads(int);
int orc_prove(const char* code, const unsigned char* input, size_t n_in, unsigned log_max_rows, char** json_out, size_t* json_len, char** transcript_out, double* seconds);
const char* orc_last_error(); void orc_free(void*);
}
int main(int argc, char** argv) {
    std::ifstream f(argv[1]); std::stringstream ss; ss << f.rdbuf(); std::string code = ss.str();
    unsigned lmr = atoi(argv[2]); std::string in; if (argc > 3) { std::ifstream fi(argv[3], std::ios::binary); std::stringstream si; si << fi.rdbuf(); in = si.str(); }
    std::string out[2];
    for (int simd = 0; simd < 2; simd++) {
        orc_set_threads(4); orc_set_simd(simd);
        char* js = nullptr; size_t n = 0; char* tr = nullptr; double sec = 0;
        int rc = orc_prove(code.c_str(), (const unsigned char*)in.data(), in.size(), lmr, &js, &n, &tr, &sec);
        if (rc < 0) { printf("error %s\n", orc_last_error()); return 1; }
        out[simd].assign(js, n); orc_free(js); orc_free(tr);
        printf("simd %d: %zu bytes, %.1f s\n", simd, n, sec);
    }
    printf("identical: %d\n", out[0] == out[1]);
    return out[0] == out[1] ? 0 : 2;
}
