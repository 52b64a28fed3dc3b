// oracle/ref_consts.cpp -- the reference's own CPU-side constant setup, compiled untouched.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/build_ref.py with -I/root/reference/src into
// oracle/_ref/libovrfsr_ref.so; never part of the product library.
//
// Under A_CPU the FSR headers expose only FsrEasuCon / FsrRcasCon (the filter bodies are gated on
// A_GPU, fsr/ffx_fsr1.h:232,:679); nis/NIS_Config.h is plain C++.  This is exactly how
// postprocess/PostProcessor.cpp:7-11 includes them.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define A_CPU
#include "fsr/ffx_a.h"
#include "fsr/ffx_fsr1.h"
#include "nis/NIS_Config.h"

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API void ref_easu_con(uint32_t *con, float inVpW, float inVpH, float inW, float inH, float outW, float outH)
{
    FsrEasuCon(con, con + 4, con + 8, con + 12, inVpW, inVpH, inW, inH, outW, outH);
}

REF_API void ref_rcas_con(uint32_t *con, float stops) { FsrRcasCon(con, stops); }

REF_API uint32_t ref_f32_to_f16(float f) { return AU1_AH1_AF1(f); }

REF_API float ref_clamp_f1(float x, float n, float m) { return AClampF1(x, n, m); }

REF_API int ref_nis_config_size(void) { return (int)sizeof(NISConfig); }

// PostProcessor.cpp:308 call shape: viewport = texture, origins 0.
REF_API int ref_nis_scaler_config(void *cfg256, float sharpness, uint32_t inW, uint32_t inH, uint32_t outW, uint32_t outH)
{
    NISConfig c;
    std::memset(&c, 0, sizeof(c));
    bool ok = NVScalerUpdateConfig(c, sharpness, 0, 0, inW, inH, inW, inH, 0, 0, outW, outH, outW, outH);
    std::memcpy(cfg256, &c, sizeof(c));
    return ok ? 1 : 0;
}

// PostProcessor.cpp:433 call shape.
REF_API int ref_nis_sharpen_config(void *cfg256, float sharpness, uint32_t inW, uint32_t inH)
{
    NISConfig c;
    std::memset(&c, 0, sizeof(c));
    bool ok = NVSharpenUpdateConfig(c, sharpness, 0, 0, inW, inH, inW, inH, 0, 0);
    std::memcpy(cfg256, &c, sizeof(c));
    return ok ? 1 : 0;
}

// 64 x 8 floats each (NIS_Config.h:261-393)
REF_API void ref_nis_coefs(float *scale, float *usm)
{
    std::memcpy(scale, coef_scale, sizeof(float) * kPhaseCount * kFilterSize);
    std::memcpy(usm, coef_usm, sizeof(float) * kPhaseCount * kFilterSize);
}
