// constants.cpp -- host-side constant setup of the upscale path (product code; no oracle involved).
// Restates, for a headless HIP host:
//   FsrEasuCon                         src/fsr/ffx_fsr1.h:156-202
//   FsrRcasCon + truncating f32->f16   src/fsr/ffx_fsr1.h:662-672, src/fsr/ffx_a.h:482-552
//   imageCentre / radius               src/postprocess/PostProcessor.cpp:298-305, 331-335
// All arithmetic is fp32 with one rounding per operator (separate statements: the host compiler may
// not fuse across them), so the bit patterns equal what the reference's A_CPU build produces.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "postprocessor.hpp"

namespace ovrfsr {

static inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

void easu_con(uint32_t con[16], float inVpW, float inVpH, float inW, float inH, float outW, float outH)
{
    const float rOutW = 1.0f / outW, rOutH = 1.0f / outH, rInW = 1.0f / inW, rInH = 1.0f / inH;
    // con0: output pixel -> input pixel position.  scale = viewport/out ; bias = 0.5*scale - 0.5
    float sx = inVpW * rOutW;
    float sy = inVpH * rOutH;
    float hx = 0.5f * inVpW; hx = hx * rOutW; hx = hx - 0.5f;
    float hy = 0.5f * inVpH; hy = hy * rOutH; hy = hy - 0.5f;
    con[0] = bits(sx); con[1] = bits(sy); con[2] = bits(hx); con[3] = bits(hy);
    // con1..con3: normalised gather offsets.  Integer addressing never reads them, but they are part
    // of the cbuffer the reference uploads, so the known-answer tests cover them too.
    const float gx[4] = {1.0f, -1.0f, 1.0f, 0.0f};  // con1.z, con2.x, con2.z, con3.x  (x rInW)
    const float gy[4] = {-1.0f, 2.0f, 2.0f, 4.0f};  // con1.w, con2.y, con2.w, con3.y  (x rInH)
    con[4] = bits(rInW); con[5] = bits(rInH);
    con[6] = bits(gx[0] * rInW); con[7] = bits(gy[0] * rInH);
    con[8] = bits(gx[1] * rInW); con[9] = bits(gy[1] * rInH);
    con[10] = bits(gx[2] * rInW); con[11] = bits(gy[2] * rInH);
    con[12] = bits(gx[3] * rInW); con[13] = bits(gy[3] * rInH);
    con[14] = 0; con[15] = 0;
}

// Truncating (round-toward-zero) f32 -> f16 with saturation of inf/nan to +-65504 and denormal
// support: the arithmetic form of the base[]/shift[] tables at ffx_a.h:482-549.
static uint32_t f32_to_f16_trunc(float f)
{
    const uint32_t u = bits(f);
    const uint32_t sign = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (e < 103u) return sign;                                           // underflow to +-0
    if (e < 113u) return (sign | (1u << (e - 103u))) + (m >> (126u - e)); // half denormal
    if (e < 143u) return (sign | ((e - 112u) << 10)) + (m >> 13);        // normal
    return sign | 0x7bffu;                                               // >= 65536, inf, nan
}

void rcas_con(uint32_t con[4], float stops)
{
    const float lin = exp2f(-stops); // stops -> linear
    const uint32_t h = f32_to_f16_trunc(lin);
    con[0] = bits(lin);
    con[1] = h + (h << 16);
    con[2] = 0;
    con[3] = 0;
}

// float -> uint32 as the reference's `(uint32_t)x` gives it wherever that is defined (0 <= x < 2^32: truncation), and TOTAL elsewhere:
// NaN and negative values give 0, values >= 2^32 give 0xffffffff -- the saturating ftou of the shader model the cbuffer is read by.
// The C++ cast itself is undefined behaviour outside the range (PostProcessor.cpp:298-305 has the same hazard; x86 happens to
// produce 0x80000000-flavoured garbage): a C ABI that promises "nothing here throws" does not inherit it.  oracle/fsr_oracle.c
// (ovo_mask_constants) converts the same way.
static inline uint32_t f2u_sat(float x)
{
    if (!(x > 0.0f)) return 0u;                 // NaN, -0, negatives
    if (x >= 4294967296.0f) return 0xffffffffu; // +inf included
    return (uint32_t)x;
}

void mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t outW, uint32_t outH, const float proj[4],
                    float cfgRadius, int onlyOneEye, int eye)
{
    // every value is a float expression truncated into a uint32 cbuffer slot
    if (onlyOneEye) {
        const float px = eye ? proj[2] : proj[0], py = eye ? proj[3] : proj[1];
        centre[0] = centre[2] = f2u_sat(outW * px);
        centre[1] = centre[3] = f2u_sat(outH * py);
    } else {
        const uint32_t half = outW / 2; // integer halving first (PostProcessor.cpp:298,300)
        centre[0] = f2u_sat(half * proj[0]);
        centre[1] = f2u_sat(outH * proj[1]);
        centre[2] = f2u_sat(half * (1 + proj[2]));
        centre[3] = f2u_sat(outH * proj[3]);
    }
    radius[0] = f2u_sat(0.5f * cfgRadius * outH);
    radius[1] = radius[0] * radius[0];
    radius[2] = outW;
    radius[3] = outH;
}

// Which branch do the gw x gh groups of an outW x outH dispatch take?  FSR: 16x16 groups, centre +8
// (fsr_easu.hlsl:41-45); NVScaler 32x24, centre +16,+12 (NIS_Upscale.hlsl:98); NVSharpen 32x32 (NIS_Sharpen.hlsl:96).
uint32_t classify_mask(const uint32_t c[4], uint32_t r2, uint32_t outW, uint32_t outH, uint32_t gw, uint32_t gh)
{
    const uint32_t gxN = (outW + gw - 1) / gw, gyN = (outH + gh - 1) / gh;
    bool anyIn = false, anyOut = false;
    for (uint32_t gy = 0; gy < gyN; ++gy)
        for (uint32_t gx = 0; gx < gxN; ++gx) {
            const uint32_t cx = gx * gw + gw / 2, cy = gy * gh + gh / 2;
            const uint32_t ax = c[0] - cx, ay = c[1] - cy, bx = c[2] - cx, by = c[3] - cy;
            const bool in = (ax * ax + ay * ay <= r2) || (bx * bx + by * by <= r2);
            anyIn |= in;
            anyOut |= !in;
        }
    if (!anyOut) return MASK_ALL_INSIDE;
    if (!anyIn) return MASK_ALL_OUTSIDE;
    return MASK_MIXED;
}

} // namespace ovrfsr
