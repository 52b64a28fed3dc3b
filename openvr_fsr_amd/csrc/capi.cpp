// capi.cpp -- the C ABI declared in include/openvr_fsr_amd.h over ovrfsr::PostProcessor.
// No exception crosses this boundary (the reference swallows failures the same way,
// PostProcessor.cpp:145-152).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include "postprocessor.hpp"
#include "fsr_launch.h"
#include "fsr_bounds.h"
#include "nis_tables.h"

struct ovrfsr_ctx {
    ovrfsr::PostProcessor *pp;
};

namespace {
bool config_ok(const ovrfsr_config *cfg) { return cfg && cfg->struct_size == sizeof(ovrfsr_config); }
// what create / set_config accept: a ctx is never built around a configuration no kernel exists for (and so never
// disables itself over one at the first apply)
// The float fields feed float -> uint32 conversions (mask centre and radius, PostProcessor.cpp:298-305) and clamps: a ctx is only ever built
// around values for which every one of them is defined (header, "Configuration values").  The reference validates none of this -- its only
// rule is `if (sharpness < 0) sharpness = 0` (Config.h:40) -- and is undefined for the values refused here.
bool floats_valid(const ovrfsr_config *cfg)
{
    if (!std::isfinite(cfg->radius) || cfg->radius < 0.0f) return false;
    if (!std::isfinite(cfg->sharpness) || !std::isfinite(cfg->render_scale)) return false;
    for (int i = 0; i < 4; ++i)
        if (!(cfg->proj_centre[i] >= -1.0f && cfg->proj_centre[i] <= 2.0f)) return false; // NaN fails the comparison
    return true;
}
bool config_valid(const ovrfsr_config *cfg)
{
    return config_ok(cfg) && (cfg->precision == OVRFSR_PRECISION_FP32 || cfg->precision == OVRFSR_PRECISION_FP32_STRICT) &&
           cfg->stage_mask >= 0 && cfg->stage_mask <= 2 && cfg->fused >= -1 && cfg->fused <= 1 && (cfg->pair_submit == 0 || cfg->pair_submit == 1) &&
           floats_valid(cfg);
}

// Nothing may unwind through the extern "C" boundary (header: "nothing here throws"): host-side containers of the launch
// manager can throw std::bad_alloc, which becomes a status like every other failure.
template <typename F>
int guarded(F &&f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return OVRFSR_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return OVRFSR_ERR_INVALID_ARGUMENT;
    }
}
constexpr uint32_t kMaxExtent = 16384; // same limit CheckImage puts on caller images
} // namespace

extern "C" {

OVRFSR_API uint32_t ovrfsr_abi_version(void) { return OVRFSR_ABI_VERSION; }

#ifdef OVRFSR_TIE_AUDIT
// AUDIT BUILDS ONLY (-DOVRFSR_TIE_AUDIT; not part of the ABI, not declared in include/openvr_fsr_amd.h, absent from the shipped library):
// the current device's near-tie audit counters {audited, listed, flips, small-channel half differences, max distance bytes (fp32 bits),
// max distance half spacings (fp32 bits)} -- see g_ovrfsr_tie_audit in fsr_kernels.hip.  tools/debug/tie_audit.py drives it.
OVRFSR_API int ovrfsr_debug_tie_audit(unsigned long long counts[6], int reset)
{
    return ovrfsr::tie_audit_read(counts, reset != 0) == hipSuccess ? OVRFSR_OK : OVRFSR_ERR_HIP;
}
#endif

#ifdef OVRFSR_BOUNDS
// CHECKED BUILDS ONLY (-DOVRFSR_BOUNDS, fsr_bounds.h; not part of the ABI, not declared in include/openvr_fsr_amd.h, absent from the shipped
// library): the current device's checked-accessor counters, summed over the two kernel translation units, n = ovrfsr_chk::kSlots values
// (layout: fsr_bounds.h); and the self-test launch.  tests/test_gpu_bounds.py drives them.
OVRFSR_API int ovrfsr_debug_bounds_slots(void) { return ovrfsr_chk::kSlots; }
OVRFSR_API int ovrfsr_debug_bounds(unsigned long long *counts, int n, int reset)
{
    if (!counts || n != ovrfsr_chk::kSlots) return OVRFSR_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) counts[i] = 0;
    if (ovrfsr::bounds_read_fsr(counts, reset != 0) != hipSuccess) return OVRFSR_ERR_HIP;
    return ovrfsr::bounds_read_nis(counts, reset != 0) == hipSuccess ? OVRFSR_OK : OVRFSR_ERR_HIP;
}
OVRFSR_API int ovrfsr_debug_bounds_selftest(void) { return ovrfsr::bounds_selftest() == hipSuccess ? OVRFSR_OK : OVRFSR_ERR_HIP; }
// fault injection (checked builds only): the nth device allocation / stream / event creation of the launch manager from now on fails, once; 0 disarms
OVRFSR_API void ovrfsr_debug_fail_resource(int nth) { ovrfsr::debug_fail_resource(nth); }
#endif

OVRFSR_API int ovrfsr_pair_pending(const ovrfsr_ctx *ctx) { return ctx && ctx->pp && ctx->pp->PairPending() ? 1 : 0; }

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
        if (cfg->out_width > kMaxExtent || cfg->out_height > kMaxExtent) return OVRFSR_ERR_INVALID_ARGUMENT;
        *out_w = cfg->out_width;
        *out_h = cfg->out_height;
        return OVRFSR_OK;
    }
    // uint32 <- float truncation, PostProcessor.cpp:512-518.  A zero, negative or non-finite scale (a typo in
    // openvr_mod.cfg) has no defined uint conversion, and a tiny one asks for an unbounded image: both are rejected,
    // like any size beyond the 16384 texels an image may have here.
    const float s = cfg->render_scale;
    if (!std::isfinite(s) || !(s > 0.f)) return OVRFSR_ERR_INVALID_ARGUMENT;
    const float fw = s < 1.f ? in_w / s : in_w * s, fh = s < 1.f ? in_h / s : in_h * s;
    if (!(fw < (float)(kMaxExtent + 1)) || !(fh < (float)(kMaxExtent + 1))) return OVRFSR_ERR_INVALID_ARGUMENT;
    *out_w = (uint32_t)fw;
    *out_h = (uint32_t)fh;
    return OVRFSR_OK;
}

OVRFSR_API int ovrfsr_create(int device, const ovrfsr_config *cfg, ovrfsr_ctx **out_ctx)
{
    if (!out_ctx || !config_valid(cfg)) return OVRFSR_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return OVRFSR_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OVRFSR_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OVRFSR_ERR_NO_DEVICE; // kernels are gfx950-only
    return guarded([&] {
        // owned until success: a throwing PostProcessor constructor (its std::string / std::vector members) must not leak the ctx
        std::unique_ptr<ovrfsr_ctx> c(new (std::nothrow) ovrfsr_ctx);
        if (!c) return (int)OVRFSR_ERR_OUT_OF_MEMORY;
        c->pp = nullptr;
        std::unique_ptr<ovrfsr::PostProcessor> pp(new (std::nothrow) ovrfsr::PostProcessor(device, *cfg));
        if (!pp) return (int)OVRFSR_ERR_OUT_OF_MEMORY;
        c->pp = pp.release();
        *out_ctx = c.release();
        return (int)OVRFSR_OK;
    });
}

OVRFSR_API void ovrfsr_destroy(ovrfsr_ctx *ctx)
{
    if (!ctx) return;
    delete ctx->pp;
    delete ctx;
}

OVRFSR_API int ovrfsr_set_config(ovrfsr_ctx *ctx, const ovrfsr_config *cfg)
{
    if (!ctx || !config_valid(cfg)) return OVRFSR_ERR_INVALID_ARGUMENT; // the ctx keeps its previous configuration
    return guarded([&] { return ctx->pp->SetConfig(*cfg); });
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
    return guarded([&] { ctx->pp->Reset(); return (int)OVRFSR_OK; });
}

OVRFSR_API int ovrfsr_apply(ovrfsr_ctx *ctx, int eye, const ovrfsr_image *in, const ovrfsr_bounds *bounds,
                            ovrfsr_image *out, void *stream)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return guarded([&] { return ctx->pp->Apply(eye, in, bounds, out, static_cast<hipStream_t>(stream)); });
}

OVRFSR_API int ovrfsr_apply_batch(ovrfsr_ctx *ctx, uint32_t n, int first_eye, int alternate_eyes, const ovrfsr_image *in0,
                                  size_t in_stride_bytes, const ovrfsr_image *out0, size_t out_stride_bytes, void *stream)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return guarded([&] {
        return ctx->pp->ApplyBatch(n, first_eye, alternate_eyes, in0, in_stride_bytes, out0, out_stride_bytes, static_cast<hipStream_t>(stream));
    });
}

OVRFSR_API int ovrfsr_apply_batch_shared(ovrfsr_ctx *ctx, uint32_t n, const ovrfsr_image *in0, size_t in_stride_bytes,
                                         const ovrfsr_image *out0, size_t out_stride_bytes, void *stream)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return guarded([&] {
        return ctx->pp->ApplyBatch(n, OVRFSR_EYE_LEFT, 0, in0, in_stride_bytes, out0, out_stride_bytes, static_cast<hipStream_t>(stream), true);
    });
}

OVRFSR_API const char *ovrfsr_last_error(const ovrfsr_ctx *ctx) { return ctx ? ctx->pp->LastError() : "null ctx"; }

OVRFSR_API int ovrfsr_last_gpu_time_ms(ovrfsr_ctx *ctx, float *ms)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return guarded([&] { return ctx->pp->LastGpuTimeMs(ms); });
}

OVRFSR_API int ovrfsr_average_gpu_time_ms(ovrfsr_ctx *ctx, float *ms, uint32_t *reports)
{
    if (!ctx) return OVRFSR_ERR_INVALID_ARGUMENT;
    return guarded([&] { return ctx->pp->AverageGpuTimeMs(ms, reports); });
}

OVRFSR_API void ovrfsr_easu_con(uint32_t con[16], float vw, float vh, float iw, float ih, float ow, float oh)
{
    if (con) ovrfsr::easu_con(con, vw, vh, iw, ih, ow, oh); // (null outputs: nothing to do -- no entry point dereferences a null it was handed)
}

OVRFSR_API void ovrfsr_rcas_con(uint32_t con[4], float stops) { if (con) ovrfsr::rcas_con(con, stops); }

OVRFSR_API void ovrfsr_mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t out_w, uint32_t out_h,
                                      const float proj_centre[4], float cfg_radius, int only_one_eye, int eye)
{
    if (centre && radius && proj_centre) ovrfsr::mask_constants(centre, radius, out_w, out_h, proj_centre, cfg_radius, only_one_eye, eye);
}

OVRFSR_API int ovrfsr_nis_scaler_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h)
{
    return cfg256 ? ovrfsr::nis_scaler_config(cfg256, sharpness, in_w, in_h, out_w, out_h) : 0;
}

OVRFSR_API int ovrfsr_nis_sharpen_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h)
{
    return cfg256 ? ovrfsr::nis_scaler_config(cfg256, sharpness, in_w, in_h, in_w, in_h) : 0;
}

OVRFSR_API const float *ovrfsr_nis_coef_scale(void) { return ovrfsr::nis_coef_scale(); }
OVRFSR_API const float *ovrfsr_nis_coef_usm(void) { return ovrfsr::nis_coef_usm(); }

} // extern "C"
