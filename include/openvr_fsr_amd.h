/*
 * openvr_fsr_amd.h -- C ABI of the MI355X-native per-eye upscaler (FSR1 EASU + RCAS, NIS).
 *
 * This is the drop-in boundary for the reference's post-process hot path: everything
 * vr::PostProcessor does between "game submitted an eye texture" and "upscaled texture handed to
 * the compositor" (reference: src/postprocess/PostProcessor.{h,cpp}), with the D3D11 compute
 * dispatches replaced by HIP kernels for gfx950.  Plain pointers and sizes only; no C++/torch
 * types.  All image pointers are DEVICE pointers (the reference's input is an ID3D11Texture2D
 * already resident on the GPU -- PostProcessor.cpp:133).  Nothing here throws; every entry point
 * returns an ovrfsr_status and leaves outputs untouched on failure (PostProcessor.cpp:145-152).
 *
 * Reference interface each declaration replaces is cited as file:line under /root/reference/.
 * The reference-side binding a maintainer would write is in INTEGRATION.md.
 */
#ifndef OPENVR_FSR_AMD_H
#define OPENVR_FSR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define OVRFSR_API __attribute__((visibility("default")))
#else
#define OVRFSR_API
#endif

#define OVRFSR_ABI_VERSION 5u /* 5: ovrfsr_pair_pending added; pair_submit pairs by arrival order; create / set_config validate the float fields.  4: ovrfsr_config::pair_submit (was reserved[0]; same struct size).  3: ovrfsr_apply_batch_shared added.  2: ovrfsr_average_gpu_time_ms added; precision value 1 (never built) removed */

typedef enum ovrfsr_status {
    OVRFSR_OK = 0,
    OVRFSR_ERR_INVALID_ARGUMENT = 1, /* null / misaligned / inconsistent descriptor            */
    OVRFSR_ERR_UNSUPPORTED = 2,      /* format or scale the path does not cover                 */
    OVRFSR_ERR_HIP = 3,              /* a HIP runtime call failed; see ovrfsr_last_error()      */
    OVRFSR_ERR_NO_DEVICE = 4,        /* no gfx950 device / kernels not loadable                 */
    OVRFSR_ERR_DISABLED = 5,         /* ctx disabled itself after a failed (re)build, like
                                        PostProcessor::enabled=false (PostProcessor.cpp:148-151)*/
    OVRFSR_ERR_OUT_OF_MEMORY = 6
} ovrfsr_status;

/* vr::EVREye, headers/openvr.h:149-153 */
typedef enum ovrfsr_eye { OVRFSR_EYE_LEFT = 0, OVRFSR_EYE_RIGHT = 1 } ovrfsr_eye;

/* Pixel formats of the linear device buffers that stand in for ID3D11Texture2D.
 * RGBA8_UNORM is what the reference allocates for its outputs (DetermineOutputFormat,
 * PostProcessor.cpp:63-74); RGBA16F is the packed-half I/O of BASELINE config C5; RGBA32F exists
 * so that parity can be measured on un-quantised results.  RGB10A2_UNORM (DXGI_FORMAT_R10G10B10A2_UNORM
 * bit layout: R 0-9, G 10-19, B 20-29, A 30-31) is the other format DetermineOutputFormat returns: a
 * 10-bit submission keeps 10-bit intermediate and output textures.  It is accepted as input with a
 * RGB10A2 (or, for parity measurements, RGBA32F) output, two-kernel pipeline only (cfg.fused = 1 is
 * rejected for it).  BGRA8_UNORM is input-only: the B8G8R8A8 submissions the reference views through a
 * typed SRV (TranslateTypelessFormats / MakeSrgbFormatsTypeless, PostProcessor.cpp:30-61) while its own
 * textures stay R8G8B8A8; here the channels are re-ordered by a copy kernel in front of the pipeline and
 * everything after it is the RGBA8 path. */
typedef enum ovrfsr_format {
    OVRFSR_FORMAT_RGBA8_UNORM = 0,
    OVRFSR_FORMAT_RGBA16F = 1,
    OVRFSR_FORMAT_RGBA32F = 2,
    OVRFSR_FORMAT_RGB10A2_UNORM = 3,
    OVRFSR_FORMAT_BGRA8_UNORM = 4
} ovrfsr_format;

/* Arithmetic the kernels run in.  The reference only ever compiles the fp32 bodies
 * (`//#define A_HALF`, src/fsr/fsr_easu.hlsl:3).
 *   FP32         fp32 math, FMA contraction allowed, hardware rcp (<= 1 ulp): the product build.  Its quantised EASU stores (the UNORM8 /
 *                half intermediate of a pipeline, an EASU-only UNORM8 output) are nevertheless the STRICT build's, bit for bit: a pixel whose
 *                re-associated result lies within 2^-9 byte of a UNORM8 rounding boundary -- for half stores within 2^-6 of a half spacing,
 *                that band scaled with the largest texel of the tile's input footprint where it exceeds 1 (HDR: the re-association error
 *                follows the largest tap, not the output), for values >= xmin = 0.25 / 0.5 derived from the sharpness (a flipped half-ulp
 *                below xmin stays under 1e-3 behind RCAS's largest gain) -- is re-resolved in the reference's operator order (near-tie
 *                guard).  This is AUDITED, not derived: an audit build of the same kernels (-DOVRFSR_TIE_AUDIT) re-resolves every
 *                pixel in reference order on the device and counts the unlisted pixels whose stored value differs -- 0 in 1.2e12 pixels over
 *                every configuration, structured / random / extreme / natural content, RGBA16F content up to 40x the unit range, and the
 *                candidates of a directed search that maximises the distance between the two evaluations; the largest distance met is
 *                3.5e-4 byte, 0.18 of the band (profiles/r05_tie_audit.txt, profiles/r06_tie_audit_campaigns.txt,
 *                tests/test_gpu_adversarial.py).  A first-order worst-case bound of that distance is
 *                two orders of magnitude above the band (same file, section 5): the filter's one ill-conditioned step, the direction blend, is
 *                evaluated in the reference's order for that reason, the rest is rounding noise that no norm bound captures.  A violation,
 *                should one exist, is one LSB (one half-ulp) of one intermediate pixel.  Float outputs differ by <= 3e-6, UNORM8 pipeline
 *                outputs by <= 1 LSB, half pipeline outputs by <= 1e-3 on unit-range images and by <= one half-ulp of the value beyond
 *   FP32_STRICT  fp32, every operator evaluated as written (no FMA), IEEE division: bit-identical
 *                to the CPU oracle; a validation build, not a fast one
 * Pixels OUTSIDE the radius (the reference's `Bilinear` fallback, fsr_easu.hlsl:33-36, and NIS `DirectCopy`, NIS_Upscale.hlsl:77-90 --
 * 78 % of a frame at the shipped radius 0.5) and NVScaler's chroma tap are one SampleLevel through D3D11's default linear-clamp sampler.
 * Both builds evaluate it as the D3D11 functional spec describes that sampler -- texel coordinate snapped to 8 fractional bits, round to
 * nearest, weights applied unfused -- and are bit-identical to the oracle there.  That reading is a restatement of the spec, not of anything
 * in the reference (it is fixed-function hardware), so its exposure is stated instead of its proof (profiles/r06_sampler_exposure.txt): if
 * hardware keeps MORE sub-texel bits (10, 12, exact float weights), pixels outside the radius move by <= 1 LSB -- none at all at BASELINE's
 * C2 / C3 shape, whose scale is exactly 3/4 (every coordinate a multiple of 1/4), 1.9 % of the bytes at x1.3 (C4 shape) on structured and
 * natural content, 10.8 % on uniform noise; a TRUNCATING snap would move 0.6 % / 3.9 % of them by <= 2 LSB.
 * TEXEL VALUES the statements above cover (tests/test_gpu_formats.py::test_texel_value_domain): every finite value whose fp32 products do
 * not overflow -- the whole RGBA16F range, negative values, denormals, RGBA32F magnitudes up to 1e18.  One thing IEEE 754 leaves open shows
 * through: min / max of a +0 and a -0.  An image holding zeros of BOTH signs can make a result that is a zero carry the other sign than the
 * oracle's (x86 and gfx950 choose differently); colour images (values >= +0) never meet it.  NaN / +-Inf texels, and magnitudes whose products
 * overflow (1e30), are outside the parity contract and inside the memory-safety one: what the pixels whose taps reach such a texel hold is
 * implementation-defined (EASU / RCAS: the same pixels are NaN in every build and in the oracle), every other pixel is unaffected, nothing is
 * read or written out of place (checked build, profiles/r06_bounds.txt).
 * There is no packed-half arithmetic mode (value 1 was reserved for one in ABI 1 and is rejected with
 * OVRFSR_ERR_INVALID_ARGUMENT by ovrfsr_create / ovrfsr_set_config): on gfx950 v_pk_*_f16 issues at the rate of
 * v_pk_*_f32 (profiles/r02_valu_issue_rates.txt), the fp32 kernels already process two taps per packed instruction, and
 * half accumulation of 12 taps misses the 1e-3 tolerance (measured: 3.9e-3, profiles/r03_half_acc.txt) -- the "fp16" of BASELINE configs C2/C3/C5 is served by fp32
 * arithmetic with RGBA16F images and a half intermediate where the config asks for packed-half I/O. */
typedef enum ovrfsr_precision {
    OVRFSR_PRECISION_FP32 = 0,
    OVRFSR_PRECISION_FP32_STRICT = 2
} ovrfsr_precision;

/* Stands in for vr::Texture_t{handle,eType,eColorSpace} (headers/openvr.h:177-182) plus the
 * D3D11_TEXTURE2D_DESC the reference queries from the handle (PostProcessor.cpp:137-139). */
typedef struct ovrfsr_image {
    void *data;           /* device pointer to texel (0,0); aligned to the texel size (4 / 8 / 16 B)    */
    uint32_t width;       /* texels                                                              */
    uint32_t height;      /* texels                                                              */
    uint32_t pitch_bytes; /* distance between rows; multiple of the texel size; >= width*texel   */
    uint32_t format;      /* ovrfsr_format                                                       */
} ovrfsr_image;

/* vr::VRTextureBounds_t, headers/openvr.h:609-613.  Only |uMax-uMin| > 0.5 is consulted
 * ("texture contains only one eye", PostProcessor.cpp:146). */
typedef struct ovrfsr_bounds { float uMin, vMin, uMax, vMax; } ovrfsr_bounds;

/* The numeric fields of the reference's Config singleton that reach the GPU
 * (src/postprocess/Config.h:10-17), plus the values the reference obtains from the OpenVR runtime
 * (projection centres, PostProcessor.cpp:104-121) and implementation knobs of this library.
 *
 * Configuration values.  The mask centre and radius are float expressions converted to uint32 (PostProcessor.cpp:298-305); the
 * reference validates nothing -- its one rule is `if (sharpness < 0) sharpness = 0` (Config.h:40) -- and that conversion is undefined
 * behaviour in C++ for NaN, negative or huge values.  This library never evaluates it on such a value:
 *   - ovrfsr_create / ovrfsr_set_config return OVRFSR_ERR_INVALID_ARGUMENT (no ctx is built / the old cfg stays) for a radius that is
 *     not finite or is negative, a sharpness or render_scale that is not finite, or a proj_centre component outside [-1, 2] (or NaN);
 *   - sharpness is otherwise clamped to [0, 1] where the reference clamps it (PostProcessor.cpp:420, NIS_Config.h);
 *   - ovrfsr_mask_constants is total: for 0 <= x < 2^32 the conversion is the reference's truncation (every known answer holds), NaN
 *     and negative values give 0, values >= 2^32 give 0xffffffff (the saturating float->uint of the shader model that reads the
 *     cbuffer); radius^2 keeps the reference's uint32 wrap-around (radius * outH / 2 >= 65536: defined, if useless, there too);
 *   - ovrfsr_config_from_json clamps a negative radius to 0 exactly as the reference clamps a negative sharpness, and treats a
 *     number that overflows float as "Could not read config file" (defaults, OVRFSR_ERR_INVALID_ARGUMENT); numbers are read in the "C"
 *     locale whatever locale the host process has set (a comma-decimal LC_NUMERIC would make atof("0.77") return 0). */
typedef struct ovrfsr_config {
    uint32_t struct_size;    /* = sizeof(ovrfsr_config); ABI guard                               */
    int32_t fsr_enabled;     /* Config::fsrEnabled: 0 -> apply() is a pass-through (output = input
                                handle untouched, PostProcessor.cpp:135)                         */
    int32_t use_nis;         /* Config::useNis.  DEPARTURE FROM THE REFERENCE, NVScaler only: the reference ignores
                                NVScalerUpdateConfig's `false` for a scale outside [0.5, 1] (in/out; i.e. upscales
                                beyond 2x, or any render_scale > 1) and dispatches with a half-initialised NISConfig
                                block (PostProcessor.cpp:308, NIS_Config.h:144-160: the result is undefined).  Here
                                such a configuration fails the (re)build: ovrfsr_apply* returns
                                OVRFSR_ERR_UNSUPPORTED, the ctx disables itself like any failed PrepareResources
                                (PostProcessor.cpp:148-151) and the caller's texture goes through untouched       */
    int32_t debug_mode;      /* Config::debugMode: tints pixels outside the radius               */
    float render_scale;      /* Config::renderScale; <1: out = in / scale, >=1: out = in * scale,
                                both truncated to uint (PostProcessor.cpp:512-518)               */
    float sharpness;         /* Config::sharpness, [0,1] (clamped where the reference clamps)    */
    float radius;            /* Config::radius, in units of outH/2; 2.0 disables the mask; finite, >= 0 */
    float proj_centre[4];    /* {Lx, Ly, Rx, Ry} in [0,1]: what CalculateProjectionCenter()
                                returned for each eye; default 0.5; accepted range [-1, 2]       */
    uint32_t out_width;      /* explicit output size; 0,0 = derive from render_scale             */
    uint32_t out_height;
    int32_t precision;       /* ovrfsr_precision                                                 */
    int32_t quantize_intermediate; /* 1 = EASU result is stored as UNORM8 before RCAS reads it,
                                as the reference's R8G8B8A8_UNORM intermediate texture does
                                (PostProcessor.cpp:348); 0 = intermediate kept in float          */
    int32_t fused;           /* 0 two kernels through an HBM intermediate, 1 one kernel with the
                                intermediate in LDS (same results for the same quantize_intermediate
                                up to the product build's rounding), -1 auto: whichever is faster
                                for the configuration (two kernels unmasked; with a radius mask the
                                tiles outside the radius are written in final form by a concurrent
                                kernel and the tiles touching it take the two-kernel form for RGBA8
                                pipelines, the fused one for half / float)                       */
    int32_t stage_mask;      /* 0 = the reference's stage selection (upscale iff scale != 1, sharpen
                                iff !use_nis || scale == 1; PostProcessor.cpp:586-594).  1 = upscale
                                stage only (BASELINE config C1 "EASU-only"), 2 = sharpen stage only  */
    int32_t pair_submit;     /* 0 = every ovrfsr_apply processes its eye at once: the reference's pattern, two dispatches per Submit,
                                four dependent launches per frame (PostProcessor.cpp:586-594).
                                1 = deferred pair, for submissions with one texture per eye: the ovrfsr_apply of the FIRST eye of a frame
                                -- LEFT or RIGHT, games submit in either order -- only RECORDS the submission and returns, in *out, the
                                image that eye's result will be written to (ovrfsr_pair_pending() then returns 1); the ovrfsr_apply of
                                the OTHER eye then launches BOTH as one batch of two -- two launches per frame.  The first eye's output is complete, in stream order, once the
                                second call has returned: a host must hold back whatever consumes it (the forwarded Submit,
                                INTEGRATION.md) until then, and the first eye's INPUT texture must stay unmodified until then too (it is
                                read when the pair is launched, not when it is recorded).  Same pixels, bit for bit.
                                One frame costs 0.097 instead of 0.110 ms of GPU time at 1683x1869 -> 2244x2492 (ramp-up, the partly
                                filled last round of workgroups and the drain are paid per launch), 0.05 instead of 0.07 ms at the
                                shipped radius 0.5.  The two images of a pair may sit anywhere in memory relative to each other (inputs
                                and outputs independently).  A recorded eye whose other eye does not follow -- the same eye again, a
                                texture of another size or format, one texture submitted for both eyes, outputs that overlap,
                                ovrfsr_apply_batch* -- is processed on its own by that next call.  A disturbed sequence does not become
                                a standing lag: after the same eye twice in a row nothing is recorded until the other eye is seen again
                                (a host that submits ONE eye per frame pays one late result, once), and a frame's second eye that finds
                                nothing recorded (its partner went out with a batch call or a size change) is processed at once; only a
                                host that REVERSES its eye order mid-stream should call ovrfsr_reset.  ovrfsr_reset /
                                ovrfsr_set_config / ovrfsr_destroy DROP a recorded eye and forget the learned order.  Both eyes must get their own output image:
                                caller-owned ones, or the two ctx-owned images of this mode (a ctx-owned image handed out for a recorded
                                eye stays valid across the rebuild a size change triggers, until the next reset)                */
    int32_t reserved[2];
} ovrfsr_config;

typedef struct ovrfsr_ctx ovrfsr_ctx; /* one per device; not thread-safe; distinct ctxs are independent */

/* Config.h:10-17 defaults (fsrEnabled=false, renderScale=1, sharpness=0.75, radius=0.5), centres 0.5 */
OVRFSR_API void ovrfsr_config_default(ovrfsr_config *cfg);

/* Stands in for the implicit construction of the global `postProcessor` (VrHooks.cpp:19) +
 * Config::Instance().  `device` is a HIP device ordinal. */
OVRFSR_API int ovrfsr_create(int device, const ovrfsr_config *cfg, ovrfsr_ctx **out_ctx);
OVRFSR_API void ovrfsr_destroy(ovrfsr_ctx *ctx);

/* What the reference's hotkeys do: mutate Config then Reset() (PostProcessor.cpp:659-709). */
OVRFSR_API int ovrfsr_set_config(ovrfsr_ctx *ctx, const ovrfsr_config *cfg);
OVRFSR_API int ovrfsr_get_config(const ovrfsr_ctx *ctx, ovrfsr_config *cfg);

/* PostProcessor::Reset() (PostProcessor.h:13, .cpp:166-194): drop cached constants and ctx-owned
 * resources, re-enable after a failure. */
OVRFSR_API int ovrfsr_reset(ovrfsr_ctx *ctx);

/* Output size rule of PrepareResources (PostProcessor.cpp:512-518). */
OVRFSR_API int ovrfsr_output_size(const ovrfsr_config *cfg, uint32_t in_width, uint32_t in_height,
                                  uint32_t *out_width, uint32_t *out_height);

/* PostProcessor::Apply(EVREye, const Texture_t*, const VRTextureBounds_t*, EVRSubmitFlags)
 * (PostProcessor.h:12, .cpp:123-164).
 *   in      the submitted eye texture (or the shared side-by-side texture).
 *   bounds  may be NULL (= {0,0,1,1}, PostProcessor.cpp:128-131).
 *   out     in/out.  If out->data is NULL the ctx-owned output image is used and *out is filled in
 *           -- the reference's behaviour of swapping Texture_t::handle for its own texture
 *           (PostProcessor.cpp:161), valid until the next apply on this ctx.  If out->data is set
 *           the result is written there (width/height must equal the output size; format
 *           selects the store conversion; its bytes must not overlap the input image's: the sharpen
 *           stages read neighbour texels other workgroups write -- OVRFSR_ERR_INVALID_ARGUMENT).  When the ctx is a pass-through (fsr disabled, or the
 *           second Submit of a shared texture, PostProcessor.cpp:155-158) *out describes the
 *           image the compositor should receive and nothing is launched.
 *   stream  hipStream_t (NULL = default stream).  Launches are asynchronous on it.  A ctx owns scratch device
 *           memory (the EASU->RCAS intermediate, its output image) and an auxiliary stream that is forked from
 *           and joined back into `stream` with events: consecutive calls on one ctx must be ordered by the
 *           caller -- same stream, or synchronised streams -- exactly like draws on one D3D11 immediate context.
 *           After the first call for a given input size a call allocates nothing and can be captured into a
 *           HIP graph.  A call that WOULD have to build something while `stream` is being captured (the first call for an
 *           input size, a larger batch than any before, the first use of the ctx-owned output) is refused with
 *           OVRFSR_ERR_INVALID_ARGUMENT before it touches anything -- the capture stays valid, the ctx stays enabled: make
 *           the same call once outside the capture.  A captured call records no debug-mode timestamps.
 * (Re)builds constants on first use and whenever the input size changes (PostProcessor.cpp:136-153). */
OVRFSR_API int ovrfsr_apply(ovrfsr_ctx *ctx, int eye, const ovrfsr_image *in, const ovrfsr_bounds *bounds,
                            ovrfsr_image *out, void *stream);

/* Batch form for headless throughput: n eye images of identical shape, image i at
 * base + i*stride_bytes, one launch over the whole batch.  Eye of image i is
 * first_eye ^ (alternate_eyes ? (i & 1) : 0): a batch of stereo pairs is laid out L,R,L,R,...
 * Each image is its own texture (textureContainsOnlyOneEye), outputs are caller-owned and must not overlap the inputs. */
OVRFSR_API int ovrfsr_apply_batch(ovrfsr_ctx *ctx, uint32_t n, int first_eye, int alternate_eyes,
                                  const ovrfsr_image *in0, size_t in_stride_bytes,
                                  const ovrfsr_image *out0, size_t out_stride_bytes, void *stream);

/* The same for SHARED side-by-side textures -- one image holds both eyes, what games that submit a single texture with
 * half-width bounds do (|uMax-uMin| <= 0.5, PostProcessor.cpp:146): every image is processed once with the two mask
 * centres of the shared layout (imageCentre of the left eye and width/2 * (1 + projX_R) for the right one,
 * PostProcessor.cpp:155-158,298-301), exactly as the first ovrfsr_apply of such a texture does. */
OVRFSR_API int ovrfsr_apply_batch_shared(ovrfsr_ctx *ctx, uint32_t n, const ovrfsr_image *in0, size_t in_stride_bytes,
                                         const ovrfsr_image *out0, size_t out_stride_bytes, void *stream);

OVRFSR_API const char *ovrfsr_last_error(const ovrfsr_ctx *ctx);

/* Averaged GPU time of the launches of the last apply, like the debug-mode timestamp queries
 * (PostProcessor.cpp:579-628).  Blocks until that work is done.  Only recorded when
 * cfg.debug_mode != 0. */
OVRFSR_API int ovrfsr_last_gpu_time_ms(ovrfsr_ctx *ctx, float *ms);

/* The reference's debug-mode log line "Average GPU processing time for upscale" (PostProcessor.cpp:605-626): a ring of
 * 6 timestamp pairs, the oldest read back after every apply, the mean of 500 readings published -- doubled when each eye
 * has its own texture (one reading = one eye, the figure = one frame).  *ms = the last published mean (0 before the
 * first), *reports = how many have been published; with OVRFSR_LOG=1 in the environment each one is also printed to
 * stderr in the reference's wording.  Only recorded when cfg.debug_mode != 0. */
OVRFSR_API int ovrfsr_average_gpu_time_ms(ovrfsr_ctx *ctx, float *ms, uint32_t *reports);

/* cfg.pair_submit (ABI 5): 1 when the most recent ovrfsr_apply on this ctx only RECORDED its submission (nothing was launched for it; its
 * output image will be complete once the next ovrfsr_apply has returned), 0 otherwise (also for a NULL ctx and with pair_submit off).
 * What a Submit detour needs to decide whether to hold the forwarded Submit back.  The reference has no counterpart:
 * it processes every eye inside its own Submit (VrHooks.cpp:50-62). */
OVRFSR_API int ovrfsr_pair_pending(const ovrfsr_ctx *ctx);

/* ---- constants-only entry points (known-answer tests; no GPU needed) ------------------------- */

/* FsrEasuCon (src/fsr/ffx_fsr1.h:156-202); con = con0|con1|con2|con3 */
OVRFSR_API void ovrfsr_easu_con(uint32_t con[16], float in_viewport_w, float in_viewport_h, float in_w,
                                float in_h, float out_w, float out_h);
/* FsrRcasCon (src/fsr/ffx_fsr1.h:662-672); `stops` = 2 - 2*clamp(sharpness,0,1) (PostProcessor.cpp:420-421) */
OVRFSR_API void ovrfsr_rcas_con(uint32_t con[4], float stops);
/* imageCentre / radius of UpscaleConstants and SharpenConstants (PostProcessor.cpp:298-305, 331-335) */
OVRFSR_API void ovrfsr_mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t out_w, uint32_t out_h,
                                      const float proj_centre[4], float cfg_radius,
                                      int texture_contains_only_one_eye, int eye);
/* NVScalerUpdateConfig / NVSharpenUpdateConfig (src/nis/NIS_Config.h:144-255) as PostProcessor.cpp:308,:433
 * call them; cfg256 receives the 256-byte NISConfig; returns 1/0 like the reference's bool. */
OVRFSR_API int ovrfsr_nis_scaler_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h,
                                        uint32_t out_w, uint32_t out_h);
OVRFSR_API int ovrfsr_nis_sharpen_config(void *cfg256, float sharpness, uint32_t in_w, uint32_t in_h);
/* coef_scale / coef_usm, 64 phases x 8 taps (src/nis/NIS_Config.h:261-393) */
OVRFSR_API const float *ovrfsr_nis_coef_scale(void);
OVRFSR_API const float *ovrfsr_nis_coef_usm(void);

/* ---- data formats either side of the path --------------------------------- */

/* Config::Load (src/postprocess/Config.h:30-63): parse the text of an openvr_mod.cfg (JSON with // comments,
 * src/openvr_mod.cfg) into cfg with the reference's defaulting rules.  On a parse error cfg holds the struct
 * defaults and INVALID_ARGUMENT is returned ("Could not read config file").  Hotkey keys are ignored. */
OVRFSR_API int ovrfsr_config_from_json(const char *text, size_t len, ovrfsr_config *cfg);

/* Stand-in for the F7 capture (PostProcessor.cpp:640-657): dump a device image as a binary PPM. */
OVRFSR_API int ovrfsr_save_ppm(const ovrfsr_image *img, const char *path, void *stream);
/* The capture in the reference's own container: a DDS file (SaveDDSTextureToFile, PostProcessor.cpp:652; DX10 header extension, one mip)
 * holding the texels exactly as they sit in the image -- R8G8B8A8_UNORM, R16G16B16A16_FLOAT, R32G32B32A32_FLOAT, R10G10B10A2_UNORM or
 * B8G8R8A8_UNORM --, rows tightly packed.  (ABI 4) */
OVRFSR_API int ovrfsr_save_dds(const ovrfsr_image *img, const char *path, void *stream);

OVRFSR_API uint32_t ovrfsr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OPENVR_FSR_AMD_H */
