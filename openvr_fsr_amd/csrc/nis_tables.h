// nis_tables.h -- NIS host constants: the 256-byte constant block and the coefficient banks.
// Reference: src/nis/NIS_Config.h:37-77 (NISConfig layout), :144-255 (UpdateConfig), :261-393 (banks).
#pragma once
#include <stdint.h>

namespace ovrfsr {

// Same field order and size as the reference's NISConfig / the HLSL cbuffer (NIS_Upscale.hlsl:28-68):
// the kernels read it as uploaded, and the known-answer tests compare all 256 bytes.
struct alignas(256) NisConstants {
    float kDetectRatio, kDetectThres, kMinContrastRatio, kRatioNorm;
    float kContrastBoost, kEps, kSharpStartY, kSharpScaleY;
    float kSharpStrengthMin, kSharpStrengthScale, kSharpLimitMin, kSharpLimitScale;
    float kScaleX, kScaleY, kDstNormX, kDstNormY;
    float kSrcNormX, kSrcNormY;
    uint32_t kInputViewportOriginX, kInputViewportOriginY, kInputViewportWidth, kInputViewportHeight;
    uint32_t kOutputViewportOriginX, kOutputViewportOriginY, kOutputViewportWidth, kOutputViewportHeight;
    float reserved0, reserved1;
    uint32_t imageCentre[4];
    uint32_t radius[4];
};
static_assert(sizeof(NisConstants) == 256, "NISConfig is a 256-byte cbuffer");

// NVScalerUpdateConfig as PostProcessor.cpp:308 calls it (viewport == texture, origins 0).
// Returns 1/0 like the reference's bool; on 0 the block is only partly filled, exactly as there.
int nis_scaler_config(void *cfg256, float sharpness, uint32_t inW, uint32_t inH, uint32_t outW, uint32_t outH);
const float *nis_coef_scale();
const float *nis_coef_usm();

} // namespace ovrfsr
