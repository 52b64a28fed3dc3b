// nis_config.cpp -- NIS sharpness-slider -> tuning constants (SDR mode), restated from
// src/nis/NIS_Config.h:144-241 for the call shape of src/postprocess/PostProcessor.cpp:308,:433.
// fp32 throughout, one rounding per operator, so the 256-byte block is bit-identical to the
// reference's (checked by tests/test_constants.py against tests/golden/ref_consts.json).
#include "nis_tables.h"
#include <cstring>
#include "nis_coef_tables.inc"

namespace ovrfsr {

int nis_scaler_config(void *cfg256, float sharpness, uint32_t inW, uint32_t inH, uint32_t outW, uint32_t outH)
{
    NisConstants c;
    std::memcpy(&c, cfg256, sizeof(c)); // keep whatever the caller had: a failed update leaves fields untouched

    sharpness = sharpness < 1.f ? sharpness : 1.f;
    sharpness = sharpness > 0.f ? sharpness : 0.f;
    const float slider = sharpness - 0.5f; // [0,1] -> [-0.5,+0.5]
    const bool upper = slider >= 0.0f;     // separate gains above and below 50 %
    const float minScale = upper ? 1.25f : 1.0f;
    const float limitScale = upper ? 1.25f : 1.0f;

    const float detectRatio = 1127.f / 1024.f;
    const float detectThres = 64.0f / 1024.0f;
    const float minContrast = 2.0f, maxContrast = 10.0f;
    const float startY = 0.45f, endY = 0.9f;

    float strengthMin = 0.4f + slider * minScale * 1.2f;
    strengthMin = strengthMin > 0.0f ? strengthMin : 0.0f;
    const float strengthMax = 1.6f + slider * 1.8f;
    float limitMin = 0.14f + slider * limitScale * 0.32f;
    limitMin = limitMin > 0.1f ? limitMin : 0.1f;
    const float limitMax = 0.5f + slider * limitScale * 0.6f;

    const float ratioNorm = 1.0f / (maxContrast - minContrast);
    const float scaleY = 1.0f / (endY - startY);
    const float strengthScale = strengthMax - strengthMin;
    const float limitRange = limitMax - limitMin;

    // viewport width/height of 0 mean "whole texture"; PostProcessor passes the texture size for both
    c.kInputViewportWidth = inW;
    c.kInputViewportHeight = inH;
    c.kOutputViewportWidth = outW;
    c.kOutputViewportHeight = outH;
    if (inW == 0 || inH == 0 || outW == 0 || outH == 0) { std::memcpy(cfg256, &c, sizeof(c)); return 0; }

    c.kInputViewportOriginX = c.kInputViewportOriginY = 0;
    c.kOutputViewportOriginX = c.kOutputViewportOriginY = 0;
    c.kSrcNormX = 1.f / inW;
    c.kSrcNormY = 1.f / inH;
    c.kDstNormX = 1.f / outW;
    c.kDstNormY = 1.f / outH;
    c.kScaleX = inW / float(outW);
    c.kScaleY = inH / float(outH);
    if (c.kScaleX < 0.5f || c.kScaleX > 1.f || c.kScaleY < 0.5f || c.kScaleY > 1.f) {
        std::memcpy(cfg256, &c, sizeof(c)); // NIS only scales 1x..2x; the reference ignores this result
        return 0;
    }
    c.kDetectRatio = detectRatio;
    c.kDetectThres = detectThres;
    c.kMinContrastRatio = minContrast;
    c.kRatioNorm = ratioNorm;
    c.kContrastBoost = 1.0f;
    c.kEps = 1.0f;
    c.kSharpStartY = startY;
    c.kSharpScaleY = scaleY;
    c.kSharpStrengthMin = strengthMin;
    c.kSharpStrengthScale = strengthScale;
    c.kSharpLimitMin = limitMin;
    c.kSharpLimitScale = limitRange;
    std::memcpy(cfg256, &c, sizeof(c));
    return 1;
}

const float *nis_coef_scale() { return reinterpret_cast<const float *>(kNisCoefScaleBits); }
const float *nis_coef_usm() { return reinterpret_cast<const float *>(kNisCoefUsmBits); }

} // namespace ovrfsr
