/*
 * oracle/nis_oracle.c -- CPU restatement of the NVIDIA Image Scaling path of fholger/openvr_fsr.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/fsr_oracle.c for the rules).
 *
 * What it restates (file:line relative to /root/reference/src):
 *   - getY                               nis/NIS_Scaler.h:160-169 (NIS_HDR_MODE_NONE)
 *   - GetEdgeMap                         nis/NIS_Scaler.h:176-293
 *   - CalcLTI / EvalPoly6 / FilterNormal nis/NIS_Scaler.h:343-375, 399-434, 436-453
 *   - GetInterpEdgeMap / GetDirFilters   nis/NIS_Scaler.h:377-397, 455-583
 *   - NVScaler                           nis/NIS_Scaler.h:589-770
 *   - CalcLTIFast / EvalUSM / GetDirUSM / NVSharpen   nis/NIS_Scaler.h:790-971
 *   - DirectCopy + main (radius mask)    nis/NIS_Upscale.hlsl:77-107, nis/NIS_Sharpen.hlsl:75-105
 * Compile-time configuration as the reference builds it: NIS_HDR_MODE 0, NIS_VIEWPORT_SUPPORT 0,
 * NIS_USE_HALF_PRECISION 0 (NIS_SCALE_INT 255, NIS_SCALE_FLOAT 255.0), NIS_TEXTURE_GATHER 0,
 * blocks 32x24 (scaler) / 32x32 (sharpen), 256 threads.
 *
 * Formulation.  The shader stages a luma tile and an edge-map tile per thread group in groupshared
 * memory.  Both are pure functions of the (clamped) input texel they belong to -- tile cell c of a
 * group starting at srcBlockStart holds texel srcBlockStart + c - 2 -- so this restatement computes
 * them once per texel for the whole image and indexes them globally; results are identical as long
 * as the shader's float tile-index arithmetic (NIS_Scaler.h:615-616, floor(i * 1/numPixelsX)) is exact,
 * which holds for every tile width <= 80 (valid scales give <= 40; checked exhaustively).  The luma
 * fetches are SampleLevel calls at texel centres: with D3D11's 8-bit sub-texel addressing they
 * return the texel exactly.
 *
 * Arithmetic contract: fp32, every operator as written and in source order, no FMA, HLSL literals
 * are float, lerp(x,y,s) = x + s*(y-x), `unorm` UAV stores clamp to [0,1].
 *
 * Pinned bit-for-bit against the reference's own NIS_Scaler.h compiled through oracle/hlsl_shim.hpp
 * (oracle/_ref; tests/test_oracle_nis.py) and against tests/golden/nis_vectors.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OVO_API __attribute__((visibility("default")))

typedef struct {
    float kDetectRatio, kDetectThres, kMinContrastRatio, kRatioNorm;
    float kContrastBoost, kEps, kSharpStartY, kSharpScaleY;
    float kSharpStrengthMin, kSharpStrengthScale, kSharpLimitMin, kSharpLimitScale;
    float kScaleX, kScaleY, kDstNormX, kDstNormY;
    float kSrcNormX, kSrcNormY;
    uint32_t kInputViewportOriginX, kInputViewportOriginY, kInputViewportWidth, kInputViewportHeight;
    uint32_t kOutputViewportOriginX, kOutputViewportOriginY, kOutputViewportWidth, kOutputViewportHeight;
    float reserved0, reserved1;
    uint32_t centre[4];
    uint32_t radius[4];
    uint32_t pad[28];
} nis_cb_t; /* NISConfig, nis/NIS_Config.h:37-77, as the HLSL cbuffer reads it (NIS_Upscale.hlsl:28-68) */

typedef struct { const float *px; int w, h; } image_t;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float lerpf(float x, float y, float s) { return x + s * (y - x); }
static inline const float *texel_clamp(const image_t *im, int x, int y)
{
    return im->px + 4 * ((size_t)clampi(y, 0, im->h - 1) * im->w + clampi(x, 0, im->w - 1));
}
static inline float getY(const float *c) { return 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]; }

/* The reference fills its luma tiles with SampleLevel at TEXEL CENTRES (NIS_Scaler.h:653, 902; NIS_TEXTURE_GATHER is off in this build): the
 * snapped coordinate has fraction 0, so the four taps carry the weights 1, 0, 0, 0 -- the texel itself for every finite image, but a NaN / Inf
 * neighbour to the right or below still enters as 0 x NaN.  Restated as written (round 6: the pin campaign's non-finite family). */
static inline float luma_centre(const image_t *im, int x, int y)
{
    const float *c00 = texel_clamp(im, x, y), *c10 = texel_clamp(im, x + 1, y), *c01 = texel_clamp(im, x, y + 1), *c11 = texel_clamp(im, x + 1, y + 1);
    float s[3];
    for (int ch = 0; ch < 3; ++ch) s[ch] = ((c00[ch] * 1.0f + c10[ch] * 0.0f) + c01[ch] * 0.0f) + c11[ch] * 0.0f;
    return getY(s);
}

static inline void fixed8(float t, int *i0, float *frac)
{
    float s = floorf(t * 256.0f + 0.5f);
    float f = floorf(s * (1.0f / 256.0f));
    *i0 = (int)f;
    *frac = (s - f * 256.0f) * (1.0f / 256.0f);
}
extern int g_ovo_sampler_bits, g_ovo_sampler_trunc; /* fsr_oracle.c: the sampler model of the exposure study (default 8 bits, rounding) */
static inline void fixed_model(float t, int *i0, float *frac)
{
    if (g_ovo_sampler_bits <= 0) { float f = floorf(t); *i0 = (int)f; *frac = t - f; return; }
    const float one = (float)(1 << g_ovo_sampler_bits);
    float s = g_ovo_sampler_trunc ? floorf(t * one) : floorf(t * one + 0.5f);
    float f = floorf(s / one);
    *i0 = (int)f;
    *frac = (s - f * one) / one;
}
static void sample_bilinear(float out[4], const image_t *im, float u, float v)
{
    float tx = u * (float)im->w - 0.5f, ty = v * (float)im->h - 0.5f;
    int x0, y0;
    float fx, fy;
    if (g_ovo_sampler_bits == 8 && !g_ovo_sampler_trunc) { fixed8(tx, &x0, &fx); fixed8(ty, &y0, &fy); }
    else { fixed_model(tx, &x0, &fx); fixed_model(ty, &y0, &fy); }
    const float *c00 = texel_clamp(im, x0, y0), *c10 = texel_clamp(im, x0 + 1, y0);
    const float *c01 = texel_clamp(im, x0, y0 + 1), *c11 = texel_clamp(im, x0 + 1, y0 + 1);
    float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    for (int ch = 0; ch < 4; ++ch) out[ch] = ((c00[ch] * w00 + c10[ch] * w10) + c01[ch] * w01) + c11[ch] * w11;
}

/* NIS_Scaler.h:176-293 on the 3x3 block q[r][c] = p[r+i][c+j] */
static void edge_map(float w[4], float q[3][3], const nis_cb_t *cb)
{
    const float g_0 = fabsf(q[0][0] + q[0][1] + q[0][2] - q[2][0] - q[2][1] - q[2][2]);
    const float g_45 = fabsf(q[1][0] + q[0][0] + q[0][1] - q[2][1] - q[2][2] - q[1][2]);
    const float g_90 = fabsf(q[0][0] + q[1][0] + q[2][0] - q[0][2] - q[1][2] - q[2][2]);
    const float g_135 = fabsf(q[1][0] + q[2][0] + q[2][1] - q[0][1] - q[0][2] - q[1][2]);
    const float g_0_90_max = fmaxf(g_0, g_90), g_0_90_min = fminf(g_0, g_90);
    const float g_45_135_max = fmaxf(g_45, g_135), g_45_135_min = fminf(g_45, g_135);
    float e_0_90 = 0, e_45_135 = 0;
    float edge_0 = 0, edge_45 = 0, edge_90 = 0, edge_135 = 0;
    if ((g_0_90_max + g_45_135_max) == 0) {
        e_0_90 = 0;
        e_45_135 = 0;
    } else {
        e_0_90 = g_0_90_max / (g_0_90_max + g_45_135_max);
        e_0_90 = fminf(e_0_90, 1.0f);
        e_45_135 = 1.0f - e_0_90;
    }
    if ((g_0_90_max > (g_0_90_min * cb->kDetectRatio)) && (g_0_90_max > cb->kDetectThres) && (g_0_90_max > g_45_135_min)) {
        if (g_0_90_max == g_0) { edge_0 = 1.0f; edge_90 = 0; }
        else { edge_0 = 0; edge_90 = 1.0f; }
    } else {
        edge_0 = 0;
        edge_90 = 0;
    }
    if ((g_45_135_max > (g_45_135_min * cb->kDetectRatio)) && (g_45_135_max > cb->kDetectThres) && (g_45_135_max > g_0_90_min)) {
        if (g_45_135_max == g_45) { edge_45 = 1.0f; edge_135 = 0; }
        else { edge_45 = 0; edge_135 = 1.0f; }
    } else {
        edge_45 = 0;
        edge_135 = 0;
    }
    float weight_0, weight_90, weight_45, weight_135;
    if ((edge_0 + edge_90 + edge_45 + edge_135) >= 2.0f) {
        if (edge_0 == 1.0f) { weight_0 = e_0_90; weight_90 = 0; }
        else { weight_0 = 0; weight_90 = e_0_90; }
        if (edge_45 == 1.0f) { weight_45 = e_45_135; weight_135 = 0; }
        else { weight_45 = 0; weight_135 = e_45_135; }
    } else if ((edge_0 + edge_90 + edge_45 + edge_135) >= 1.0f) {
        weight_0 = edge_0;
        weight_90 = edge_90;
        weight_45 = edge_45;
        weight_135 = edge_135;
    } else {
        weight_0 = 0;
        weight_90 = 0;
        weight_45 = 0;
        weight_135 = 0;
    }
    w[0] = weight_0; w[1] = weight_90; w[2] = weight_45; w[3] = weight_135;
}

/* NIS_Scaler.h:343-375 */
static float calc_lti(float p0, float p1, float p2, float p3, float p4, float p5, int phase_index, const nis_cb_t *cb)
{
    float y0, y1, y2, y3, y4;
    if (phase_index <= 64 / 2) { y0 = p0; y1 = p1; y2 = p2; y3 = p3; y4 = p4; }
    else { y0 = p1; y1 = p2; y2 = p3; y3 = p4; y4 = p5; }
    const float a_min = fminf(fminf(y0, y1), y2), a_max = fmaxf(fmaxf(y0, y1), y2);
    const float b_min = fminf(fminf(y2, y3), y4), b_max = fmaxf(fmaxf(y2, y3), y4);
    const float a_cont = a_max - a_min, b_cont = b_max - b_min;
    const float cont_ratio = fmaxf(a_cont, b_cont) / (fminf(a_cont, b_cont) + cb->kEps);
    return (1.0f - sat((cont_ratio - cb->kMinContrastRatio) * cb->kRatioNorm)) * cb->kContrastBoost;
}

/* NIS_Scaler.h:399-434; coefficient banks are [64][8] (LoadFilterBanksSh copies taps 0..5) */
static float eval_poly6(const float pxl[6], int phase_int, const nis_cb_t *cb, const float *cs, const float *cu)
{
    float y = 0.f;
    for (int i = 0; i < 6; ++i) y += cs[phase_int * 8 + i] * pxl[i];
    float y_usm = 0.f;
    for (int i = 0; i < 6; ++i) y_usm += cu[phase_int * 8 + i] * pxl[i];
    const float y_scale = 1.0f - sat((y * (1.0f / 255) - cb->kSharpStartY) * cb->kSharpScaleY);
    const float y_sharpness = y_scale * cb->kSharpStrengthScale + cb->kSharpStrengthMin;
    y_usm *= y_sharpness;
    const float y_sharpness_limit = (y_scale * cb->kSharpLimitScale + cb->kSharpLimitMin) * y;
    y_usm = fminf(y_sharpness_limit, fmaxf(-y_sharpness_limit, y_usm));
    y_usm *= calc_lti(pxl[0], pxl[1], pxl[2], pxl[3], pxl[4], pxl[5], phase_int, cb);
    return y + y_usm;
}

/* NIS_Scaler.h:436-453 */
static float filter_normal(float p[6][6], int phx, int phy, const float *cs)
{
    float h_acc = 0.0f;
    for (int j = 0; j < 6; ++j) {
        float v_acc = 0.0f;
        for (int i = 0; i < 6; ++i) v_acc += p[i][j] * cs[phy * 8 + i];
        h_acc += v_acc * cs[phx * 8 + j];
    }
    return h_acc;
}

/* NIS_Scaler.h:455-583 */
static void dir_filters(float f[4], float p[6][6], float phx, float phy, int phxi, int phyi, const nis_cb_t *cb,
                        const float *cs, const float *cu)
{
    float interp0Deg[6];
    for (int i = 0; i < 6; ++i) interp0Deg[i] = lerpf(p[i][2], p[i][3], phx);
    f[0] = eval_poly6(interp0Deg, phyi, cb, cs, cu);

    float interp90Deg[6];
    for (int i = 0; i < 6; ++i) interp90Deg[i] = lerpf(p[2][i], p[3][i], phy);
    f[1] = eval_poly6(interp90Deg, phxi, cb, cs, cu);

    float pphase_b45 = 0.5f + 0.5f * (phx - phy);
    float t45[7];
    t45[1] = lerpf(p[2][1], p[1][2], pphase_b45);
    t45[3] = lerpf(p[3][2], p[2][3], pphase_b45);
    t45[5] = lerpf(p[4][3], p[3][4], pphase_b45);
    if (pphase_b45 >= 0.5f) {
        pphase_b45 = pphase_b45 - 0.5f;
        t45[0] = lerpf(p[1][1], p[0][2], pphase_b45);
        t45[2] = lerpf(p[2][2], p[1][3], pphase_b45);
        t45[4] = lerpf(p[3][3], p[2][4], pphase_b45);
        t45[6] = lerpf(p[4][4], p[3][5], pphase_b45);
    } else {
        pphase_b45 = 0.5f - pphase_b45;
        t45[0] = lerpf(p[1][1], p[2][0], pphase_b45);
        t45[2] = lerpf(p[2][2], p[3][1], pphase_b45);
        t45[4] = lerpf(p[3][3], p[4][2], pphase_b45);
        t45[6] = lerpf(p[4][4], p[5][3], pphase_b45);
    }
    float interp45Deg[6];
    float pphase_p45 = phx + phy;
    if (pphase_p45 >= 1) {
        for (int i = 0; i < 6; i++) interp45Deg[i] = t45[i + 1];
        pphase_p45 = pphase_p45 - 1;
    } else {
        for (int i = 0; i < 6; i++) interp45Deg[i] = t45[i];
    }
    f[2] = eval_poly6(interp45Deg, (int)(pphase_p45 * 64), cb, cs, cu);

    float pphase_b135 = 0.5f * (phx + phy);
    float t135[7];
    t135[1] = lerpf(p[3][1], p[4][2], pphase_b135);
    t135[3] = lerpf(p[2][2], p[3][3], pphase_b135);
    t135[5] = lerpf(p[1][3], p[2][4], pphase_b135);
    if (pphase_b135 >= 0.5f) {
        pphase_b135 = pphase_b135 - 0.5f;
        t135[0] = lerpf(p[4][1], p[5][2], pphase_b135);
        t135[2] = lerpf(p[3][2], p[4][3], pphase_b135);
        t135[4] = lerpf(p[2][3], p[3][4], pphase_b135);
        t135[6] = lerpf(p[1][4], p[2][5], pphase_b135);
    } else {
        pphase_b135 = 0.5f - pphase_b135;
        t135[0] = lerpf(p[4][1], p[3][0], pphase_b135);
        t135[2] = lerpf(p[3][2], p[2][1], pphase_b135);
        t135[4] = lerpf(p[2][3], p[1][2], pphase_b135);
        t135[6] = lerpf(p[1][4], p[0][3], pphase_b135);
    }
    float interp135Deg[6];
    float pphase_p135 = 1 + (phx - phy);
    if (pphase_p135 >= 1) {
        for (int i = 0; i < 6; ++i) interp135Deg[i] = t135[i + 1];
        pphase_p135 = pphase_p135 - 1;
    } else {
        for (int i = 0; i < 6; ++i) interp135Deg[i] = t135[i];
    }
    f[3] = eval_poly6(interp135Deg, (int)(pphase_p135 * 64), cb, cs, cu);
}

/* group-granular radius test, NIS_Upscale.hlsl:98-101 / NIS_Sharpen.hlsl:96-99 */
static inline int group_inside(uint32_t bx, uint32_t by, uint32_t bw, uint32_t bh, const nis_cb_t *cb)
{
    uint32_t cx = bx * bw + bw / 2, cy = by * bh + bh / 2;
    uint32_t d1x = cb->centre[0] - cx, d1y = cb->centre[1] - cy, d2x = cb->centre[2] - cx, d2y = cb->centre[3] - cy;
    return (d1x * d1x + d1y * d1y <= cb->radius[1]) || (d2x * d2x + d2y * d2y <= cb->radius[1]);
}

/* DirectCopy of the scaler, NIS_Upscale.hlsl:77-90: bilinear sample at pos/outSize */
static void direct_copy_px(float o[4], const image_t *im, int dstX, int dstY, const nis_cb_t *cb)
{
    float c[4];
    sample_bilinear(c, im, (float)dstX / (float)cb->radius[2], (float)dstY / (float)cb->radius[3]);
    const float mul[4] = {1 - cb->reserved1 * 0, 1 - cb->reserved1 * 0.3f, 1 - cb->reserved1 * 0.3f, 1 - cb->reserved1 * 0};
    o[0] = sat(c[0] * mul[0]);
    o[1] = sat(c[1] * mul[1]);
    o[2] = sat(c[2] * mul[2]);
    o[3] = sat(1.0f * mul[3]);
}

/* Whole-image NVScaler dispatch (NIS_Upscale.hlsl:95-107 over the grid of PostProcessor.cpp:397).
 * in: RGBA fp32 inW x inH; out: RGBA fp32 outW x outH; cfg256: NISConfig with centre/radius filled in;
 * coefScale/coefUsm: [64][8]. */
OVO_API int ovo_nis_upscale(const float *in, int inW, int inH, float *out, int outW, int outH, const void *cfg256,
                            const float *coefScale, const float *coefUsm, int nthreads)
{
    nis_cb_t cb;
    memcpy(&cb, cfg256, 256);
    image_t im = {in, inW, inH};
    /* per-texel luma (x255) and edge map, over the image extended by the 3-texel support ring that the
     * clamped fetches can reach: index (x+PAD, y+PAD) */
    const int PAD = 8;
    const int PW = inW + 2 * PAD, PH = inH + 2 * PAD;
    float *Y = (float *)malloc(sizeof(float) * (size_t)PW * PH);
    float *E = (float *)malloc(sizeof(float) * 4 * (size_t)PW * PH);
    if (!Y || !E) { free(Y); free(E); return -1; }
    (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int y = 0; y < PH; ++y)
        for (int x = 0; x < PW; ++x) {
            float q[3][3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) q[r][c] = luma_centre(&im, x - PAD - 1 + c, y - PAD - 1 + r);
            edge_map(E + 4 * ((size_t)y * PW + x), q, &cb);
            Y[(size_t)y * PW + x] = q[1][1] * 255.0f;
        }
    const int gxN = (int)ceilf(outW / 32.f), gyN = (int)ceilf(outH / 24.f);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int by = 0; by < gyN; ++by)
        for (int bx = 0; bx < gxN; ++bx) {
            const int inside = group_inside((uint32_t)bx, (uint32_t)by, 32, 24, &cb);
            for (int ly = 0; ly < 24; ++ly)
                for (int lx = 0; lx < 32; ++lx) {
                    const int dstX = bx * 32 + lx, dstY = by * 24 + ly;
                    if (dstX >= outW || dstY >= outH) continue;
                    float *o = out + 4 * ((size_t)dstY * outW + dstX);
                    if (!inside) { direct_copy_px(o, &im, dstX, dstY, &cb); continue; }
                    const float srcX = (0.5f + dstX) * cb.kScaleX - 0.5f;
                    const float srcY = (0.5f + dstY) * cb.kScaleY - 0.5f;
                    const int fxi = (int)floorf(srcX), fyi = (int)floorf(srcY);
                    float p[6][6];
                    for (int i = 0; i < 6; ++i)
                        for (int j = 0; j < 6; ++j) {
                            int tx = clampi(fxi - 2 + j, -PAD, inW - 1 + PAD), ty = clampi(fyi - 2 + i, -PAD, inH - 1 + PAD);
                            p[i][j] = Y[(size_t)(ty + PAD) * PW + (tx + PAD)];
                        }
                    const float fx = srcX - floorf(srcX), fy = srcY - floorf(srcY);
                    const int fx_int = (int)(fx * 64), fy_int = (int)(fy * 64);
                    const float pixel_n = filter_normal(p, fx_int, fy_int, coefScale);
                    float opDirYU[4];
                    dir_filters(opDirYU, p, fx, fy, fx_int, fy_int, &cb, coefScale, coefUsm);
                    /* 2x2 edge maps centred in the 6x6 support: texels (fxi+j, fyi+i) */
                    const float *e[2][2];
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j) {
                            int tx = clampi(fxi + j, -PAD, inW - 1 + PAD), ty = clampi(fyi + i, -PAD, inH - 1 + PAD);
                            e[i][j] = E + 4 * ((size_t)(ty + PAD) * PW + (tx + PAD));
                        }
                    float w[4];
                    for (int c = 0; c < 4; ++c) {
                        float h0 = lerpf(e[0][0][c], e[0][1][c], fx), h1 = lerpf(e[1][0][c], e[1][1][c], fx);
                        w[c] = lerpf(h0, h1, fy) * 255;
                    }
                    const float opY = (opDirYU[0] * w[0] + opDirYU[1] * w[1] + opDirYU[2] * w[2] + opDirYU[3] * w[3] +
                                       pixel_n * (255.0f - w[0] - w[1] - w[2] - w[3])) * (1.0f / 255.0f);
                    float op[4];
                    sample_bilinear(op, &im, (dstX + 0.5f) * cb.kDstNormX, (dstY + 0.5f) * cb.kDstNormY);
                    const float corr = opY * (1.0f / 255.0f) - getY(op);
                    op[0] += corr;
                    op[1] += corr;
                    op[2] += corr;
                    o[0] = sat(op[0]); o[1] = sat(op[1]); o[2] = sat(op[2]); o[3] = sat(op[3]);
                }
        }
    free(Y);
    free(E);
    return 0;
}

/* DirectCopy of the sharpener, NIS_Sharpen.hlsl:75-88: plain texel load (same size in and out) */
static void direct_copy_load_px(float o[4], const image_t *im, int dstX, int dstY, const nis_cb_t *cb)
{
    const float *c = im->px + 4 * ((size_t)dstY * im->w + dstX);
    const float mul[4] = {1 - cb->reserved1 * 0, 1 - cb->reserved1 * 0.3f, 1 - cb->reserved1 * 0.3f, 1 - cb->reserved1 * 0};
    o[0] = sat(c[0] * mul[0]);
    o[1] = sat(c[1] * mul[1]);
    o[2] = sat(c[2] * mul[2]);
    o[3] = sat(1.0f * mul[3]);
}

/* ---- NVSharpen, NIS_Scaler.h:783-971 ---------------------------------------------------------- */
static float calc_lti_fast(const float y[5], const nis_cb_t *cb)
{
    const float a_min = fminf(fminf(y[0], y[1]), y[2]), a_max = fmaxf(fmaxf(y[0], y[1]), y[2]);
    const float b_min = fminf(fminf(y[2], y[3]), y[4]), b_max = fmaxf(fmaxf(y[2], y[3]), y[4]);
    const float a_cont = a_max - a_min, b_cont = b_max - b_min;
    const float cont_ratio = fmaxf(a_cont, b_cont) / (fminf(a_cont, b_cont) + cb->kEps * (1.0f / 255.0f));
    return (1.0f - sat((cont_ratio - cb->kMinContrastRatio) * cb->kRatioNorm)) * cb->kContrastBoost;
}
static float eval_usm(const float pxl[5], float strength, float limit, const nis_cb_t *cb)
{
    float y_usm = -0.6001f * pxl[1] + 1.2002f * pxl[2] - 0.6001f * pxl[3];
    y_usm *= strength;
    y_usm = fminf(limit, fmaxf(-limit, y_usm));
    y_usm *= calc_lti_fast(pxl, cb);
    return y_usm;
}

OVO_API int ovo_nis_sharpen(const float *in, int W, int H, float *out, const void *cfg256, int nthreads)
{
    nis_cb_t cb;
    memcpy(&cb, cfg256, 256);
    image_t im = {in, W, H};
    const int gxN = (int)ceilf(W / 32.f), gyN = (int)ceilf(H / 32.f);
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int by = 0; by < gyN; ++by)
        for (int bx = 0; bx < gxN; ++bx) {
            const int inside = group_inside((uint32_t)bx, (uint32_t)by, 32, 32, &cb);
            for (int ly = 0; ly < 32; ++ly)
                for (int lx = 0; lx < 32; ++lx) {
                    const int dstX = bx * 32 + lx, dstY = by * 32 + ly;
                    if (dstX >= W || dstY >= H) continue;
                    float *o = out + 4 * ((size_t)dstY * W + dstX);
                    if (!inside) { direct_copy_load_px(o, &im, dstX, dstY, &cb); continue; }
                    /* tile cell (c) of the group holds texel dstBlock + c - 2 (kShift = 0.5 - 5/2 = -1.5);
                     * 5x5 support of pixel = cells pos..pos+4 = texels dst-2..dst+2 */
                    float p[5][5];
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j) p[i][j] = luma_centre(&im, dstX - 2 + j, dstY - 2 + i);
                    const float scaleY = 1.0f - sat((p[2][2] - cb.kSharpStartY) * cb.kSharpScaleY);
                    const float strength = scaleY * cb.kSharpStrengthScale + cb.kSharpStrengthMin;
                    const float limit = (scaleY * cb.kSharpLimitScale + cb.kSharpLimitMin) * p[2][2];
                    float d0[5], d90[5], d45[5], d135[5], rv[4];
                    for (int i = 0; i < 5; ++i) { d0[i] = p[i][2]; d90[i] = p[2][i]; }
                    rv[0] = eval_usm(d0, strength, limit, &cb);
                    rv[1] = eval_usm(d90, strength, limit, &cb);
                    d45[0] = p[1][1]; d45[1] = lerpf(p[2][1], p[1][2], 0.5f); d45[2] = p[2][2];
                    d45[3] = lerpf(p[3][2], p[2][3], 0.5f); d45[4] = p[3][3];
                    rv[2] = eval_usm(d45, strength, limit, &cb);
                    d135[0] = p[3][1]; d135[1] = lerpf(p[3][2], p[2][1], 0.5f); d135[2] = p[2][2];
                    d135[3] = lerpf(p[2][3], p[1][2], 0.5f); d135[4] = p[1][3];
                    rv[3] = eval_usm(d135, strength, limit, &cb);
                    float q[3][3], w[4];
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) q[r][c] = p[r + 1][c + 1]; /* GetEdgeMap(p, 1, 1) */
                    edge_map(w, q, &cb);
                    const float usmY = (rv[0] * w[0] + rv[1] * w[1] + rv[2] * w[2] + rv[3] * w[3]);
                    float op[4];
                    sample_bilinear(op, &im, (dstX + 0.5f) * cb.kDstNormX, (dstY + 0.5f) * cb.kDstNormY);
                    op[0] += usmY;
                    op[1] += usmY;
                    op[2] += usmY;
                    o[0] = sat(op[0]); o[1] = sat(op[1]); o[2] = sat(op[2]); o[3] = sat(op[3]);
                }
        }
    return 0;
}
