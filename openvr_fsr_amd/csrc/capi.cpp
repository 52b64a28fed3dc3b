// capi.cpp -- the C ABI declared in include/openvr_fsr_amd.h over ovrfsr::PostProcessor.
// No exception crosses this boundary (the reference swallows failures the same way,
// PostProcessor.cpp:145-152).
#include <hip/hip_runtime.h>
#include <cstring>
#include <new>
#include "postprocessor.hpp"
#include "nis_tables.h"

struct ovrfsr_ctx {
    ovrfsr::PostProcessor *pp;
};

namespace {
bool config_ok(const ovrfsr_config *cfg) { return cfg && cfg->struct_size == sizeof(ovrfsr_config); }
} // namespace

extern "C" {

OVRFSR_API uint32_t ovrfsr_abi_version(void) { return OVRFSR_ABI_VERSION; }

OVRFSR_API void ovrfsr_config_default(ovrfsr_config *cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(*cfg);
    cfg->fsr_enabled = 0;      // Config.h:11
    cfg->use_nis = 0;          // :17
    cfg->debug_mode = 0;       // :16
    cfg->render_scale = 1.f;   // :13
    cfg->sharpness = 0.75f;    // :14
    cfg->radius = 0.5f;        // :15
    cfg->proj_centre[0] = cfg->proj_centre[1] = cfg->proj_centre[2] = cfg->proj_centre[3] = 0.5f;
    cfg->precision = OVRFSR_PRECISION_FP32;
    cfg->quantize_intermediate = 1;
    cfg->fused = -1;
}

OVRFSR_API int ovrfsr_output_size(const ovrfsr_config *cfg, uint32_t in_w, uint32_t in_h, uint32_t *out_w, uint32_t *out_h)
{
    if (!config_ok(cfg) || !out_w || !out_h) return OVRFSR_ERR_INVALID_ARGUMENT;
    if (cfg->out_width != 0 && cfg->out_height != 0) {
        *out_w = cfg->out_width;
        *out_h = cfg->out_height;
        return OVRFSR_OK;
    }
    // uint32 <- float truncation, PostProcessor.cpp:512-518
    if (cfg->render_scale < 1.f) {
        *out_w = (uint32_t)(in_w / cfg->render_scale);
        *out_h = (uint32_t)(in_h / cfg->render_scale);
    } else {
        *out_w = (uint32_t)(in_w * cfg->render_scale);
        *out_h = (uint32_t)(in_h * cfg->render_scale);
    }
    return OVRFSR_OK;
}

OVRFSR_API int ovrfsr_create(int device, const ovrfsr_config *cfg, ovrfsr_ctx **out_ctx)
{
    if (!out_ctx || !config_ok(cfg)) return OVRFSR_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return OVRFSR_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OVRFSR_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OVRFSR_ERR_NO_DEVICE; // kernels are gfx950-only
    ovrfsr_ctx *c = new (std::nothrow) ovrfsr_ctx;
    if (!c) return OVRFSR_ERR_OUT_OF_MEMORY;
    c->pp = new (std::nothrow) ovrfsr::PostProcessor(device, *cfg);
    if (!c->pp) { delete c; return OVRFSR_ERR_OUT_OF_MEMORY; }
    *out_ctx = c;
    return OVRFSR_OK;
}

OVRFSR_API void ovrfsr_destroy(ovrfsr_ctx *ctx)
{
    if (!ctx) return;
    delete ctx->pp;
    delete ctx;
}

OVRFSR_API int ovrfsr_set_config(ovrfsr_ctx *ctx, const ovrfsr_config *cfg)
{
    if (!ctx || !config_ok(cfg)) return OVRFSR_ERR_INVALID_ARGUMENT;
    return ctx->pp->SetConfig(*cfg);
}

OVRFSR_API int ovrfsr_get_config(const ovrfsr_ctx *ctx, ovrfsr_config *cfg)
{
    if (!ctx || !cfg) return OVRFSR_ERR_INVALID_ARGUMENT;
    *cfg = ctx->pp->GetConfig();
    return OVRFSR_OK;
}

OVRFSR_API int ovrfsr_reset(ovrfsr_ctx *ctx)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    ctx->pp->Reset();
    return OVRFSR_OK;
}

OVRFSR_API int ovrfsr_apply(ovrfsr_ctx *ctx, int eye, const ovrfsr_image *in, const ovrfsr_bounds *bounds,
                            ovrfsr_image *out, void *stream)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return ctx->pp->Apply(eye, in, bounds, out, static_cast<hipStream_t>(stream));
}

OVRFSR_API int ovrfsr_apply_batch(ovrfsr_ctx *ctx, uint32_t n, int first_eye, int alternate_eyes, const ovrfsr_image *in0,
                                  size_t in_stride_bytes, const ovrfsr_image *out0, size_t out_stride_bytes, void *stream)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return ctx->pp->ApplyBatch(n, first_eye, alternate_eyes, in0, in_stride_bytes, out0, out_stride_bytes,
                               static_cast<hipStream_t>(stream));
}

OVRFSR_API const char *ovrfsr_last_error(const ovrfsr_ctx *ctx) { return ctx ? ctx->pp->LastError() : "null ctx"; }

OVRFSR_API int ovrfsr_last_gpu_time_ms(ovrfsr_ctx *ctx, float *ms)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return ctx->pp->LastGpuTimeMs(ms);
}

OVRFSR_API void ovrfsr_easu_con(uint32_t con[16], float vw, float vh, float iw, float ih, float ow, float oh)
{
    ovrfsr::easu_con(con, vw, vh, iw, ih, ow, oh);
}

OVRFSR_API void ovrfsr_rcas_con(uint32_t con[4], float stops) { ovrfsr::rcas_con(con, stops); }

OVRFSR_API void ovrfsr_mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t out_w, uint32_t out_h,
                                      const float proj_centre[4], float cfg_radius, int only_one_eye, int eye)
{
    ovrfsr::mask_constants(centre, radius, out_w, out_h, proj_centre, cfg_radius, only_one_eye, eye);
}

OVRFSR_API int ovrfsr_nis_scaler_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h)
{
    return ovrfsr::nis_scaler_config(cfg256, sharpness, in_w, in_h, out_w, out_h);
}

OVRFSR_API int ovrfsr_nis_sharpen_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h)
{
    return ovrfsr::nis_scaler_config(cfg256, sharpness, in_w, in_h, in_w, in_h);
}

OVRFSR_API const float *ovrfsr_nis_coef_scale(void) { return ovrfsr::nis_coef_scale(); }
OVRFSR_API const float *ovrfsr_nis_coef_usm(void) { return ovrfsr::nis_coef_usm(); }

} // extern "C"
