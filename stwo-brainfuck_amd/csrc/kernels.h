// Internal launcher declarations shared by the .hip translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include "m31.h"

namespace bf {

// fft.hip
void gen_twiddles(hipStream_t stream, u32* d_tw, u32* d_itw, u32 R, const uint2* d_tlo, const uint2* d_thi);
// Batched transform of `ncols` columns of 2^log cells. inverse: evaluations (bit-reversed) -> coefficients, scaled by 2^-log.
// forward: coefficients of 2^src_log (zero-extended to 2^log) -> evaluations on the canonic domain of size 2^log.
// circle = false selects "line mode" (no circle layer) used for 16x-replicated columns stored row-granular.
void fft_batch(hipStream_t stream, bool inverse, const u32* const* d_src, u32* const* d_dst, u32 ncols, u32 log, u32 src_log, bool circle,
               const u32* tw, const u32* itw, u32 tw_root_log);

}  // namespace bf
