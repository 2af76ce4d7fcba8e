// C ABI (include/bfhip.h) over the gfx950 kernels. No torch types, plain pointers and sizes.
#include "../../include/bfhip.h"
#include "ctx.h"
#include <cstdio>

using namespace bf;

static thread_local std::string g_err;
void bfhip_set_error(const std::string& s) { g_err = s; }

#define API_TRY try {
#define API_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -1; }

namespace bf {

void Ctx::init(int dev, u32 max_log_domain) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw HipError("no HIP device: the bfhip backend has no CPU fallback");
    if (dev < 0 || dev >= n) throw HipError("bad device id");
    if (max_log_domain < 6 || max_log_domain > 30) throw HipError("max_log_domain out of range [6, 30]");
    device = dev;
    BF_HIP(hipSetDevice(dev));
    BF_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    BF_HIP(hipHostMalloc((void**)&h_stage, stage_bytes));
    BF_HIP(hipMalloc((void**)&d_stage, stage_bytes));
    // point tables: G^a and G^(b << 16) for the M31 circle generator G = (2, 1268011823)
    std::vector<uint2> tlo(1 << 16), thi(1 << 15);
    auto mulp = [](uint2 p, uint2 q) { return uint2{m_sub(m_mul(p.x, q.x), m_mul(p.y, q.y)), m_add(m_mul(p.x, q.y), m_mul(p.y, q.x))}; };
    uint2 g{2u, 1268011823u}, cur{1u, 0u};
    for (u32 i = 0; i < (1u << 16); i++) { tlo[i] = cur; cur = mulp(cur, g); }
    uint2 g16 = cur;  // G^(2^16)
    cur = uint2{1u, 0u};
    for (u32 i = 0; i < (1u << 15); i++) { thi[i] = cur; cur = mulp(cur, g16); }
    BF_HIP(hipMalloc((void**)&d_tlo, tlo.size() * sizeof(uint2)));
    BF_HIP(hipMalloc((void**)&d_thi, thi.size() * sizeof(uint2)));
    BF_HIP(hipMemcpy(d_tlo, tlo.data(), tlo.size() * sizeof(uint2), hipMemcpyHostToDevice));
    BF_HIP(hipMemcpy(d_thi, thi.data(), thi.size() * sizeof(uint2), hipMemcpyHostToDevice));
    tw_root_log = max_log_domain - 1;
    BF_HIP(hipMalloc((void**)&d_tw, sizeof(u32) << tw_root_log));
    BF_HIP(hipMalloc((void**)&d_itw, sizeof(u32) << tw_root_log));
    gen_twiddles(stream, d_tw, d_itw, tw_root_log, d_tlo, d_thi);
    BF_HIP(hipGetLastError());
    sync();
}

void Ctx::destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    arena.release();
    (void)hipFree(d_tw); (void)hipFree(d_itw); (void)hipFree(d_tlo); (void)hipFree(d_thi); (void)hipFree(d_stage);
    if (h_stage) (void)hipHostFree(h_stage);
    if (stream) (void)hipStreamDestroy(stream);
}

}  // namespace bf

extern "C" {

const char* bfhip_last_error(void) { return g_err.c_str(); }

int32_t bfhip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int32_t bfhip_ctx_create(int32_t device_id, uint32_t max_log_domain, bfhip_ctx** out) {
    API_TRY
    auto* c = new bfhip_ctx();
    try { c->c.init(device_id, max_log_domain); } catch (...) { delete c; throw; }
    *out = c;
    return 0;
    API_CATCH
}
int32_t bfhip_ctx_destroy(bfhip_ctx* ctx) { API_TRY if (ctx) { ctx->c.destroy(); delete ctx; } return 0; API_CATCH }
int32_t bfhip_ctx_sync(bfhip_ctx* ctx) { API_TRY ctx->c.sync(); return 0; API_CATCH }

int32_t bfhip_malloc(bfhip_ctx* ctx, size_t bytes, void** out_d) { API_TRY BF_HIP(hipSetDevice(ctx->c.device)); BF_HIP(hipMalloc(out_d, bytes ? bytes : 4)); return 0; API_CATCH }
int32_t bfhip_free(bfhip_ctx* ctx, void* p) { API_TRY (void)ctx; BF_HIP(hipFree(p)); return 0; API_CATCH }
int32_t bfhip_upload(bfhip_ctx* ctx, void* dst_d, const void* src_h, size_t bytes) {
    API_TRY BF_HIP(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, ctx->c.stream)); ctx->c.sync(); return 0; API_CATCH
}
int32_t bfhip_download(bfhip_ctx* ctx, void* dst_h, const void* src_d, size_t bytes) {
    API_TRY BF_HIP(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, ctx->c.stream)); ctx->c.sync(); return 0; API_CATCH
}
int32_t bfhip_memset_zero(bfhip_ctx* ctx, void* dst_d, size_t bytes) { API_TRY BF_HIP(hipMemsetAsync(dst_d, 0, bytes, ctx->c.stream)); return 0; API_CATCH }

int32_t bfhip_twiddles(bfhip_ctx* ctx, const uint32_t** tw_d, const uint32_t** itw_d, uint32_t* root_log) {
    API_TRY *tw_d = ctx->c.d_tw; *itw_d = ctx->c.d_itw; *root_log = ctx->c.tw_root_log; return 0; API_CATCH
}

int32_t bfhip_interpolate(bfhip_ctx* ctx, uint32_t* const* src_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, int32_t replicated) {
    API_TRY
    Ctx& c = ctx->c;
    if (replicated && log_size < 4) throw HipError("replicated columns need log_size >= 4");
    if (!replicated && log_size < 3) throw HipError("circle transforms need log_size >= 3");
    if (log_size > c.tw_root_log + 1) throw HipError("log_size exceeds the context's twiddle tree");
    u32 log = replicated ? log_size - 4 : log_size;
    c.stage_checkpoint();
    auto* s = c.stage(src_cols_h, n_cols);
    auto* d = c.stage(dst_cols_h, n_cols);
    fft_batch(c.stream, true, (const u32* const*)s, (u32* const*)d, n_cols, log, log, !replicated, c.d_tw, c.d_itw, c.tw_root_log);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}

int32_t bfhip_evaluate(bfhip_ctx* ctx, uint32_t* const* coeff_cols_h, uint32_t* const* dst_cols_h, uint32_t n_cols, uint32_t log_size, uint32_t log_eval, int32_t replicated) {
    API_TRY
    Ctx& c = ctx->c;
    if (log_eval < log_size) throw HipError("log_eval < log_size");
    if (replicated && log_size < 4) throw HipError("replicated columns need log_size >= 4");
    if (!replicated && log_eval < 3) throw HipError("circle transforms need log_eval >= 3");
    if (log_eval > c.tw_root_log + 1) throw HipError("log_eval exceeds the context's twiddle tree");
    u32 sh = replicated ? 4 : 0;
    c.stage_checkpoint();
    auto* s = c.stage(coeff_cols_h, n_cols);
    auto* d = c.stage(dst_cols_h, n_cols);
    fft_batch(c.stream, false, (const u32* const*)s, (u32* const*)d, n_cols, log_eval - sh, log_size - sh, !replicated, c.d_tw, c.d_itw, c.tw_root_log);
    BF_HIP(hipGetLastError());
    return 0;
    API_CATCH
}

}  // extern "C"
