// ORACLE — TEST INFRASTRUCTURE ONLY (see field.h header). Interface of the SIMD mode of the CPU port (oracle/simd_port.cpp): the AVX-512
// versions of MerkleProver::commit's layer loop and of CpuBackend::evaluate / ::interpolate, switched on with orc_set_simd(1) for bench.py's
// `cpu_baseline` (stand-in for the reference's SimdBackend + rayon path). Off by default: the scalar code is the checker.
#pragma once
#include <cstddef>
#include <cstdint>

namespace orc { namespace simd {
bool available();               // the host has AVX-512 F/BW/VL/DQ
void set_enabled(bool on);      // stays off when !available()
bool enabled();
// Blake2s Merkle layer in the stwo-compress convention: nodes [0, n_nodes), n_nodes % 16 == 0; prev = the 2 * n_nodes hashes below or null
void merkle_layer_blake2s(const uint8_t* prev, const uint32_t* const* cols, size_t n_cols, size_t n_nodes, uint8_t* out);
// circle FFT / iFFT of 2^log_size values (log_size >= 5) in place; twiddles = the twiddle tree's buffer (TwiddleTree::twiddles / ::itwiddles)
void circle_fft(uint32_t* values, unsigned log_size, const uint32_t* twiddles, size_t tw_len);
void circle_ifft(uint32_t* values, unsigned log_size, const uint32_t* itwiddles, size_t tw_len, uint32_t n_inverse);
// FRI quotients of rows [r0, r1) (multiples of 16) of one size group: per sample batch the columns (index into cols), their three QM31 line
// coefficients (12 words per column: alpha a, alpha b, alpha c), the batch's random-coefficient power and the CM31 denominator inverses of the
// rows (r - r0); ys = the domain points' y coordinates of the rows. Same values as accumulate_row_quotients (prover.h), 16 rows per instruction.
struct QuotientBatch { const uint32_t* col_index; const uint32_t* line_coeffs; size_t n_cols; uint32_t batch_coeff[4]; const uint32_t* deninv_a; const uint32_t* deninv_b; };
void quotient_rows(const uint32_t* const* cols, const uint32_t* ys, const QuotientBatch* batches, size_t n_batches, size_t r0, size_t r1, uint32_t* const out[4]);
}}
