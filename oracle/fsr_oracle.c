/*
 * oracle/fsr_oracle.c -- CPU restatement of the FSR1 (EASU + RCAS) hot path of fholger/openvr_fsr.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (openvr_fsr_amd/csrc) never links
 * or calls anything in oracle/.
 *
 * What it restates (all file:line relative to /root/reference/src):
 *   - FsrEasuCon                      fsr/ffx_fsr1.h:156-202
 *   - FsrEasuTapF / SetF / FsrEasuF   fsr/ffx_fsr1.h:239-272, 275-313, 315-437
 *   - EASU shader entry + mask        fsr/fsr_easu.hlsl:21-64
 *   - FsrRcasCon                      fsr/ffx_fsr1.h:662-672 (+ truncating f32->f16, fsr/ffx_a.h:482-552)
 *   - FsrRcasF                        fsr/ffx_fsr1.h:684-769 (FSR_RCAS_LIMIT :654)
 *   - RCAS shader entry + mask        fsr/fsr_rcas.hlsl:18-55
 *   - approximations                  fsr/ffx_a.h:1843-1845 (APrxLoRcpF1/APrxMedRcpF1/APrxLoRsqF1)
 *   - host mask constants             postprocess/PostProcessor.cpp:296-305, 331-335, 420-430
 *
 * Arithmetic contract: IEEE fp32, every HLSL operator evaluated as written, NO fused
 * multiply-add (build with -ffp-contract=off), rcp(x) == 1.0f/x, min/max ignore NaN (D3D
 * semantics == fminf/fmaxf), saturate(NaN)==0.
 *
 * D3D11 fixed-function semantics restated here (the reference relies on the texture unit):
 *   - Gather4 at a texel corner returns the 2x2 footprint, each coordinate clamped to the image
 *     (tap map derived in DESIGN.md: b(fx,fy-1) c(fx+1,fy-1) e(fx-1,fy) f g h(fx+2,fy)
 *     i j k l(fx+2,fy+1) n(fx,fy+2) o(fx+1,fy+2)).
 *   - Texture2D.Load out of bounds returns 0 in every channel.
 *   - UNORM8 -> float is b/255 (correctly rounded); float -> UNORM8 is floor(sat(x)*255+0.5).
 *   - bilinear SampleLevel: texel-space coordinate u*W-0.5 snapped to 8 fractional bits
 *     (D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT; round to nearest), clamp addressing, weights from that
 *     fixed-point fraction.  A sample at a texel centre therefore returns the texel exactly.
 *
 * Pinning status: bit-exact against the reference's own FsrEasuF/FsrRcasF/FsrEasuCon/FsrRcasCon
 * compiled for the CPU through oracle/hlsl_shim.hpp (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle.py) and against the committed vectors in tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OVO_API __attribute__((visibility("default")))

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* fsr/ffx_a.h:1843-1845 */
static inline float prx_lo_rcp(float a) { return u2f(0x7ef07ebbu - f2u(a)); }
static inline float prx_med_rcp(float a) { float b = u2f(0x7ef19fffu - f2u(a)); return b * (-b * a + 2.0f); }
static inline float prx_lo_rsq(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }
static inline float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
/* fsr/ffx_a.h:1141,1166 : AMin3F1(x,y,z)=min(x,min(y,z)) */
static inline float min3(float x, float y, float z) { return fminf(x, fminf(y, z)); }
static inline float max3(float x, float y, float z) { return fmaxf(x, fmaxf(y, z)); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------ */
/* constants                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* fsr/ffx_fsr1.h:156-202.  con[0..3]=con0, [4..7]=con1, [8..11]=con2, [12..15]=con3 */
OVO_API void ovo_easu_con(uint32_t con[16], float inVpW, float inVpH, float inW, float inH,
                          float outW, float outH)
{
    con[0] = f2u(inVpW * (1.0f / outW));
    con[1] = f2u(inVpH * (1.0f / outH));
    con[2] = f2u(0.5f * inVpW * (1.0f / outW) - 0.5f);
    con[3] = f2u(0.5f * inVpH * (1.0f / outH) - 0.5f);
    con[4] = f2u(1.0f / inW);
    con[5] = f2u(1.0f / inH);
    con[6] = f2u(1.0f * (1.0f / inW));
    con[7] = f2u(-1.0f * (1.0f / inH));
    con[8] = f2u(-1.0f * (1.0f / inW));
    con[9] = f2u(2.0f * (1.0f / inH));
    con[10] = f2u(1.0f * (1.0f / inW));
    con[11] = f2u(2.0f * (1.0f / inH));
    con[12] = f2u(0.0f * (1.0f / inW));
    con[13] = f2u(4.0f * (1.0f / inH));
    con[14] = 0;
    con[15] = 0;
}

/* fsr/ffx_a.h:482-549: table-driven TRUNCATING f32->f16 (+-inf/nan -> +-65504, denormals kept).
 * Restated arithmetically: the tables are base[e] / shift[e] indexed by sign|exponent. */
OVO_API uint32_t ovo_f32_to_f16_trunc(float f)
{
    uint32_t u = f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t e = (u >> 23) & 0xffu;
    uint32_t m = u & 0x7fffffu;
    uint32_t base, shift;
    if (e < 103u) { base = 0; shift = 24; }
    else if (e < 113u) { base = 1u << (e - 103u); shift = 126u - e; }
    else if (e < 143u) { base = (e - 112u) << 10; shift = 13; }
    else { base = 0x7bffu; shift = 24; }
    return (sign | base) + (m >> shift);
}

/* fsr/ffx_fsr1.h:662-672 */
OVO_API void ovo_rcas_con(uint32_t con[4], float sharpnessStops)
{
    float s = exp2f(-sharpnessStops);
    con[0] = f2u(s);
    con[1] = ovo_f32_to_f16_trunc(s) + (ovo_f32_to_f16_trunc(s) << 16);
    con[2] = 0;
    con[3] = 0;
}

/* postprocess/PostProcessor.cpp:420-421: stops = 2 - 2*clamp(sharpness,0,1) */
OVO_API float ovo_rcas_stops_from_sharpness(float sharpness)
{
    float s = fmaxf(0.0f, fminf(sharpness, 1.0f)); /* AClampF1 = max(n,min(x,m)) ffx_a.h:353 */
    return 2.f - 2 * s;
}

/* float -> uint32: the reference's `(uint32_t)x` where that is defined (0 <= x < 2^32), total elsewhere (NaN / negative -> 0,
 * >= 2^32 -> 0xffffffff): the C cast is undefined behaviour outside the range, and the checker must not depend on the host compiler's
 * choice there.  The product converts the same way (csrc/constants.cpp f2u_sat); for every in-range value the known answers hold. */
static uint32_t f2u_sat(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

/* postprocess/PostProcessor.cpp:298-305 (shared / first-eye buffer) and :331-335 (right-eye
 * buffer when each eye has its own texture).  proj = {Lx, Ly, Rx, Ry} in [0,1].
 * Every store is float -> uint32 truncation; radius[1] is a uint32 multiply. */
OVO_API void ovo_mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t outW, uint32_t outH,
                                const float proj[4], float cfgRadius, int textureContainsOnlyOneEye,
                                int eye)
{
    if (textureContainsOnlyOneEye && eye == 1) {
        centre[0] = f2u_sat(outW * proj[2]);
        centre[1] = f2u_sat(outH * proj[3]);
        centre[2] = f2u_sat(outW * proj[2]);
        centre[3] = f2u_sat(outH * proj[3]);
    } else {
        centre[0] = f2u_sat(textureContainsOnlyOneEye ? outW * proj[0] : outW / 2 * proj[0]);
        centre[1] = f2u_sat(outH * proj[1]);
        centre[2] = f2u_sat(textureContainsOnlyOneEye ? outW * proj[0] : outW / 2 * (1 + proj[2]));
        centre[3] = f2u_sat(outH * (textureContainsOnlyOneEye ? proj[1] : proj[3]));
    }
    radius[0] = f2u_sat(0.5f * cfgRadius * outH);
    radius[1] = radius[0] * radius[0];
    radius[2] = outW;
    radius[3] = outH;
}

/* ------------------------------------------------------------------------------------------ */
/* UNORM8 conversions (D3D11 functional spec; the FSR header documents the same store rule at   */
/* fsr/ffx_fsr1.h:1075-1080)                                                                    */
/* ------------------------------------------------------------------------------------------ */
OVO_API void ovo_unorm8_to_float(const uint8_t *src, size_t n, float *dst)
{
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i] / 255.0f;
}

OVO_API void ovo_float_to_unorm8(const float *src, size_t n, uint8_t *dst)
{
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)floorf(sat(src[i]) * 255.0f + 0.5f);
}

/* ------------------------------------------------------------------------------------------ */
/* EASU                                                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const float *px; int w, h; } image_t; /* RGBA fp32, row-major, 4 floats/texel */

static inline const float *texel_clamp(const image_t *im, int x, int y)
{
    return im->px + 4 * ((size_t)clampi(y, 0, im->h - 1) * im->w + clampi(x, 0, im->w - 1));
}

/* fsr/ffx_fsr1.h:239-272 */
static inline void easu_tap(float aC[3], float *aW, float offx, float offy, float dirx, float diry,
                            float lenx, float leny, float lob, float clp, const float *c)
{
    float vx = (offx * (dirx)) + (offy * diry);
    float vy = (offx * (-diry)) + (offy * dirx);
    vx *= lenx;
    vy *= leny;
    float d2 = vx * vx + vy * vy;
    d2 = fminf(d2, clp);
    float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
    float wA = lob * d2 + -1.0f;
    wB *= wB;
    wA *= wA;
    wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
    float w = wB * wA;
    aC[0] += c[0] * w;
    aC[1] += c[1] * w;
    aC[2] += c[2] * w;
    *aW += w;
}

/* fsr/ffx_fsr1.h:275-313 */
static inline void easu_set(float dir[2], float *len, float w, float lA, float lB, float lC, float lD,
                            float lE)
{
    float dc = lD - lC;
    float cb = lC - lB;
    float lenX = fmaxf(fabsf(dc), fabsf(cb));
    lenX = prx_lo_rcp(lenX);
    float dirX = lD - lB;
    dir[0] += dirX * w;
    lenX = sat(fabsf(dirX) * lenX);
    lenX *= lenX;
    *len += lenX * w;
    float ec = lE - lC;
    float ca = lC - lA;
    float lenY = fmaxf(fabsf(ec), fabsf(ca));
    lenY = prx_lo_rcp(lenY);
    float dirY = lE - lA;
    dir[1] += dirY * w;
    lenY = sat(fabsf(dirY) * lenY);
    lenY *= lenY;
    *len += lenY * w;
}

static inline float luma2(const float *c) { return c[2] * 0.5f + (c[0] * 0.5f + c[1]); } /* :363 */

/* fsr/ffx_fsr1.h:315-437 with the Gather4 callbacks of fsr/fsr_easu.hlsl:21-23 */
static void easu_pixel(float pix[3], int ipx, int ipy, const uint32_t con[16], const image_t *im)
{
    float ppx = (float)ipx * u2f(con[0]) + u2f(con[2]);
    float ppy = (float)ipy * u2f(con[1]) + u2f(con[3]);
    float fpx = floorf(ppx), fpy = floorf(ppy);
    ppx -= fpx;
    ppy -= fpy;
    int fx = (int)fpx, fy = (int)fpy;

    const float *b = texel_clamp(im, fx, fy - 1), *c = texel_clamp(im, fx + 1, fy - 1);
    const float *e = texel_clamp(im, fx - 1, fy), *f = texel_clamp(im, fx, fy);
    const float *g = texel_clamp(im, fx + 1, fy), *h = texel_clamp(im, fx + 2, fy);
    const float *i = texel_clamp(im, fx - 1, fy + 1), *j = texel_clamp(im, fx, fy + 1);
    const float *k = texel_clamp(im, fx + 1, fy + 1), *l = texel_clamp(im, fx + 2, fy + 1);
    const float *n = texel_clamp(im, fx, fy + 2), *o = texel_clamp(im, fx + 1, fy + 2);

    float bL = luma2(b), cL = luma2(c), iL = luma2(i), jL = luma2(j), fL = luma2(f), eL = luma2(e);
    float kL = luma2(k), lL = luma2(l), hL = luma2(h), gL = luma2(g), oL = luma2(o), nL = luma2(n);

    float dir[2] = {0.0f, 0.0f};
    float len = 0.0f;
    easu_set(dir, &len, (1.0f - ppx) * (1.0f - ppy), bL, eL, fL, gL, jL);
    easu_set(dir, &len, ppx * (1.0f - ppy), cL, fL, gL, hL, kL);
    easu_set(dir, &len, (1.0f - ppx) * ppy, fL, iL, jL, kL, nL);
    easu_set(dir, &len, ppx * ppy, gL, jL, kL, lL, oL);

    float dir2x = dir[0] * dir[0], dir2y = dir[1] * dir[1];
    float dirR = dir2x + dir2y;
    int zro = dirR < (float)(1.0 / 32768.0);
    dirR = prx_lo_rsq(dirR);
    dirR = zro ? 1.0f : dirR;
    dir[0] = zro ? 1.0f : dir[0];
    dir[0] *= dirR;
    dir[1] *= dirR;
    len = len * 0.5f;
    len *= len;
    float stretch = (dir[0] * dir[0] + dir[1] * dir[1]) * prx_lo_rcp(fmaxf(fabsf(dir[0]), fabsf(dir[1])));
    float len2x = 1.0f + (stretch - 1.0f) * len;
    float len2y = 1.0f + -0.5f * len;
    float lob = 0.5f + (float)((1.0 / 4.0 - 0.04) - 0.5) * len;
    float clp = prx_lo_rcp(lob);

    float mn[3], mx[3];
    for (int ch = 0; ch < 3; ++ch) {
        mn[ch] = fminf(min3(f[ch], g[ch], j[ch]), k[ch]);
        mx[ch] = fmaxf(max3(f[ch], g[ch], j[ch]), k[ch]);
    }
    float aC[3] = {0.0f, 0.0f, 0.0f};
    float aW = 0.0f;
    easu_tap(aC, &aW, 0.0f - ppx, -1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, b);
    easu_tap(aC, &aW, 1.0f - ppx, -1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, c);
    easu_tap(aC, &aW, -1.0f - ppx, 1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, i);
    easu_tap(aC, &aW, 0.0f - ppx, 1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, j);
    easu_tap(aC, &aW, 0.0f - ppx, 0.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, f);
    easu_tap(aC, &aW, -1.0f - ppx, 0.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, e);
    easu_tap(aC, &aW, 1.0f - ppx, 1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, k);
    easu_tap(aC, &aW, 2.0f - ppx, 1.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, l);
    easu_tap(aC, &aW, 2.0f - ppx, 0.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, h);
    easu_tap(aC, &aW, 1.0f - ppx, 0.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, g);
    easu_tap(aC, &aW, 1.0f - ppx, 2.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, o);
    easu_tap(aC, &aW, 0.0f - ppx, 2.0f - ppy, dir[0], dir[1], len2x, len2y, lob, clp, n);
    float rW = 1.0f / aW;
    for (int ch = 0; ch < 3; ++ch) pix[ch] = fminf(mx[ch], fmaxf(mn[ch], aC[ch] * rW));
}

/* D3D11 fixed-point texel addressing: texel-space coordinate snapped to 8 fractional bits
 * (D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT), round to nearest. */
static inline void fixed8(float t, int *i0, float *frac)
{
    float s = floorf(t * 256.0f + 0.5f);
    float f = floorf(s * (1.0f / 256.0f));
    *i0 = (int)f;
    *frac = (s - f * 256.0f) * (1.0f / 256.0f);
}

/* The sampler model.  Default = the D3D11 functional spec as oracle/hlsl_shim.hpp restates it: 8 fractional bits, round to nearest.
 * Nothing in the reference can pin this (it is fixed-function hardware behaviour), so the EXPOSURE to it is measured instead
 * (tests/debug/sampler_exposure.py): ovo_set_sampler_model(bits, truncate) switches this library -- and nothing
 * else: not the product, not oracle/_ref -- to `bits` fractional bits (0 = exact float weights, no snap) with a rounding or truncating
 * snap.  Test infrastructure; every test and fixture runs with the default. */
int g_ovo_sampler_bits = 8, g_ovo_sampler_trunc = 0;
OVO_API void ovo_set_sampler_model(int bits, int truncate) { g_ovo_sampler_bits = bits; g_ovo_sampler_trunc = truncate; }
static inline void fixed_model(float t, int *i0, float *frac)
{
    if (g_ovo_sampler_bits <= 0) { float f = floorf(t); *i0 = (int)f; *frac = t - f; return; }
    const float one = (float)(1 << g_ovo_sampler_bits);
    float s = g_ovo_sampler_trunc ? floorf(t * one) : floorf(t * one + 0.5f);
    float f = floorf(s / one);
    *i0 = (int)f;
    *frac = (s - f * one) / one;
}

/* bilinear SampleLevel with linear/clamp sampler at normalised (u,v); fsr_easu.hlsl:33-36 */
static void sample_bilinear(float out[4], const image_t *im, float u, float v)
{
    float tx = u * (float)im->w - 0.5f;
    float ty = v * (float)im->h - 0.5f;
    int x0, y0;
    float fx, fy;
    if (g_ovo_sampler_bits == 8 && !g_ovo_sampler_trunc) { fixed8(tx, &x0, &fx); fixed8(ty, &y0, &fy); }
    else { fixed_model(tx, &x0, &fx); fixed_model(ty, &y0, &fy); }
    const float *c00 = texel_clamp(im, x0, y0), *c10 = texel_clamp(im, x0 + 1, y0);
    const float *c01 = texel_clamp(im, x0, y0 + 1), *c11 = texel_clamp(im, x0 + 1, y0 + 1);
    float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy);
    float w01 = (1.0f - fx) * fy, w11 = fx * fy;
    for (int ch = 0; ch < 4; ++ch)
        out[ch] = ((c00[ch] * w00 + c10[ch] * w10) + c01[ch] * w01) + c11[ch] * w11;
}

/* group-granular radius test shared by fsr_easu.hlsl:41-45 and fsr_rcas.hlsl:32-36 (uint32
 * wrap-around arithmetic, tile = 16x16, centre = +8) */
static inline int group_inside(uint32_t gx, uint32_t gy, const uint32_t centre[4], const uint32_t radius[4])
{
    uint32_t cx = (gx << 4) + 8u, cy = (gy << 4) + 8u;
    uint32_t d1x = centre[0] - cx, d1y = centre[1] - cy;
    uint32_t d2x = centre[2] - cx, d2y = centre[3] - cy;
    uint32_t r1 = d1x * d1x + d1y * d1y;
    uint32_t r2 = d2x * d2x + d2y * d2y;
    return r1 <= radius[1] || r2 <= radius[1];
}

/* Whole-image EASU dispatch: fsr/fsr_easu.hlsl:38-64 over the grid of PostProcessor.cpp:399.
 * in: RGBA fp32 inW x inH; out: RGBA fp32 outW x outH (alpha = 1). */
OVO_API void ovo_easu(const float *in, int inW, int inH, float *out, int outW, int outH,
                      const uint32_t con[16], const uint32_t centre[4], const uint32_t radius[4],
                      int nthreads)
{
    image_t im = {in, inW, inH};
    int groupsX = (outW + 15) >> 4, groupsY = (outH + 15) >> 4;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int gy = 0; gy < groupsY; ++gy) {
        for (int gx = 0; gx < groupsX; ++gx) {
            int inside = group_inside((uint32_t)gx, (uint32_t)gy, centre, radius);
            for (int ly = 0; ly < 16; ++ly) {
                int y = gy * 16 + ly;
                if (y >= outH) break;
                for (int lx = 0; lx < 16; ++lx) {
                    int x = gx * 16 + lx;
                    if (x >= outW) break;
                    float *o = out + 4 * ((size_t)y * outW + x);
                    if (inside) {
                        easu_pixel(o, x, y, con, &im);
                    } else {
                        float c[4];
                        /* float2(pos) / Radius.zw : uint -> float, true division */
                        sample_bilinear(c, &im, (float)x / (float)radius[2], (float)y / (float)radius[3]);
                        o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
                    }
                    o[3] = 1.0f;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* RCAS                                                                                        */
/* ------------------------------------------------------------------------------------------ */
static const float kZero4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
static inline const float *texel_load(const image_t *im, int x, int y) /* Texture2D.Load: OOB -> 0 */
{
    if (x < 0 || y < 0 || x >= im->w || y >= im->h) return kZero4;
    return im->px + 4 * ((size_t)y * im->w + x);
}

#define FSR_RCAS_LIMIT_F ((float)(0.25 - (1.0 / 16.0))) /* fsr/ffx_fsr1.h:654 */

/* fsr/ffx_fsr1.h:684-769 (FSR_RCAS_DENOISE and FSR_RCAS_PASSTHROUGH_ALPHA undefined, as in
 * fsr/fsr_rcas.hlsl:1-4; the nz term at :737-739 is dead code and omitted) */
static void rcas_pixel(float pix[3], int x, int y, const uint32_t con[4], const image_t *im)
{
    const float *b = texel_load(im, x, y - 1), *d = texel_load(im, x - 1, y);
    const float *e = texel_load(im, x, y);
    const float *f = texel_load(im, x + 1, y), *h = texel_load(im, x, y + 1);
    float lobeC[3];
    for (int ch = 0; ch < 3; ++ch) {
        float mn4 = fminf(min3(b[ch], d[ch], f[ch]), h[ch]);
        float mx4 = fmaxf(max3(b[ch], d[ch], f[ch]), h[ch]);
        float hitMin = mn4 * (1.0f / (4.0f * mx4));
        float hitMax = (1.0f - mx4) * (1.0f / (4.0f * mn4 + -4.0f));
        lobeC[ch] = fmaxf(-hitMin, hitMax);
    }
    float lobe = fmaxf(-FSR_RCAS_LIMIT_F, fminf(max3(lobeC[0], lobeC[1], lobeC[2]), 0.0f)) * u2f(con[0]);
    float rcpL = prx_med_rcp(4.0f * lobe + 1.0f);
    for (int ch = 0; ch < 3; ++ch)
        pix[ch] = (lobe * b[ch] + lobe * d[ch] + lobe * h[ch] + lobe * f[ch] + e[ch]) * rcpL;
}

/* Whole-image RCAS dispatch: fsr/fsr_rcas.hlsl:29-55 over the grid of PostProcessor.cpp:494.
 * con[3] carries debugMode as a uint (PostProcessor.cpp:430); outside the radius the pixel is
 * copied times (1,1-.3*dbg,1-.3*dbg,1) INCLUDING alpha (fsr_rcas.hlsl:46-47). */
OVO_API void ovo_rcas(const float *in, int W, int H, float *out, const uint32_t con[4],
                      const uint32_t centre[4], const uint32_t radius[4], int nthreads)
{
    image_t im = {in, W, H};
    int groupsX = (W + 15) >> 4, groupsY = (H + 15) >> 4;
    float dbg = (float)con[3];
    float mul[4] = {1.0f - dbg * 0.0f, 1.0f - dbg * 0.3f, 1.0f - dbg * 0.3f, 1.0f - dbg * 0.0f};
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int gy = 0; gy < groupsY; ++gy) {
        for (int gx = 0; gx < groupsX; ++gx) {
            int inside = group_inside((uint32_t)gx, (uint32_t)gy, centre, radius);
            for (int ly = 0; ly < 16; ++ly) {
                int y = gy * 16 + ly;
                if (y >= H) break;
                for (int lx = 0; lx < 16; ++lx) {
                    int x = gx * 16 + lx;
                    if (x >= W) break;
                    float *o = out + 4 * ((size_t)y * W + x);
                    if (inside) {
                        rcas_pixel(o, x, y, con, &im);
                        o[3] = 1.0f;
                    } else {
                        const float *s = in + 4 * ((size_t)y * W + x);
                        for (int ch = 0; ch < 4; ++ch) o[ch] = mul[ch] * s[ch];
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Whole pipeline on UNORM8 images, as PostProcessor::ApplyPostProcess runs it                 */
/* (PostProcessor.cpp:586-594): EASU -> R8G8B8A8_UNORM texture -> RCAS -> R8G8B8A8_UNORM.       */
/* quantize_intermediate=0 gives the float-intermediate variant (no 8-bit store between the     */
/* two passes).  stages: bit0 = EASU, bit1 = RCAS.  out8 (may be NULL) receives the final       */
/* UNORM8 image, outf (may be NULL) the final float image before the last store.                */
/* ------------------------------------------------------------------------------------------ */
OVO_API int ovo_fsr_pipeline_u8(const uint8_t *in8, int inW, int inH, uint8_t *out8, float *outf,
                                int outW, int outH, const uint32_t easuCon[16], const uint32_t rcasCon[4],
                                const uint32_t centre[4], const uint32_t radius[4], int stages,
                                int quantize_intermediate, int nthreads)
{
    size_t nin = (size_t)inW * inH * 4, nout = (size_t)outW * outH * 4;
    float *fin = (float *)malloc(nin * sizeof(float));
    float *a = (float *)malloc(nout * sizeof(float));
    float *b2 = (float *)malloc(nout * sizeof(float));
    uint8_t *q = (uint8_t *)malloc(nout);
    if (!fin || !a || !b2 || !q) { free(fin); free(a); free(b2); free(q); return -1; }
    ovo_unorm8_to_float(in8, nin, fin);
    float *cur;
    if (stages & 1) {
        ovo_easu(fin, inW, inH, a, outW, outH, easuCon, centre, radius, nthreads);
        cur = a;
    } else {
        if (inW != outW || inH != outH) { free(fin); free(a); free(b2); free(q); return -2; }
        memcpy(a, fin, nout * sizeof(float));
        cur = a;
    }
    if (stages & 2) {
        if ((stages & 1) && quantize_intermediate) {
            ovo_float_to_unorm8(cur, nout, q);
            ovo_unorm8_to_float(q, nout, cur);
        }
        ovo_rcas(cur, outW, outH, b2, rcasCon, centre, radius, nthreads);
        cur = b2;
    }
    if (outf) memcpy(outf, cur, nout * sizeof(float));
    if (out8) ovo_float_to_unorm8(cur, nout, out8);
    free(fin); free(a); free(b2); free(q);
    return 0;
}

OVO_API int ovo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
