// postprocessor.cpp -- see postprocessor.hpp.  Reference: src/postprocess/PostProcessor.cpp.
#include "postprocessor.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fsr_launch.h"

namespace ovrfsr {

static uint32_t texel_bytes(uint32_t fmt)
{
    return fmt == OVRFSR_FORMAT_RGBA8_UNORM || fmt == OVRFSR_FORMAT_RGB10A2_UNORM || fmt == OVRFSR_FORMAT_BGRA8_UNORM ? 4u
         : fmt == OVRFSR_FORMAT_RGBA16F ? 8u : 16u;
}

// hipSetDevice for the duration of a call, then back to whatever the caller had selected
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int dev)
    {
        (void)hipGetDevice(&prev);
        err = hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Every device allocation and every stream / event the launch manager creates goes through these.  Product build: the HIP call.  Checked builds
// (-DOVRFSR_BOUNDS): FAULT INJECTION -- ovrfsr_debug_fail_resource(n) arms a countdown and the n-th creation from then on fails with
// hipErrorOutOfMemory, once; tools/debug/fault_campaign.py walks n over every configuration and asserts what the header promises about failures
// (status codes, outputs untouched, a failed (re)build disables the ctx until reset: PostProcessor.cpp:145-152).  The reference has no fault
// injection; its failure handling has never been exercised either.
#ifdef OVRFSR_BOUNDS
static int g_fail_countdown = 0; // 0 = disarmed
void debug_fail_resource(int nth) { g_fail_countdown = nth > 0 ? nth : 0; }
static bool inject_failure() { return g_fail_countdown > 0 && --g_fail_countdown == 0; }
#else
static inline bool inject_failure() { return false; }
#endif
static inline hipError_t dev_malloc(void **p, size_t bytes) { return inject_failure() ? hipErrorOutOfMemory : hipMalloc(p, bytes); }
static inline hipError_t event_create(hipEvent_t *e, unsigned flags) { return inject_failure() ? hipErrorOutOfMemory : hipEventCreateWithFlags(e, flags); }
static inline hipError_t stream_create(hipStream_t *s, int prioMode, int least, int greatest)
{
    if (inject_failure()) return hipErrorOutOfMemory;
    return prioMode == 0 ? hipStreamCreateWithFlags(s, hipStreamNonBlocking) : hipStreamCreateWithPriority(s, hipStreamNonBlocking, prioMode == 1 ? least : greatest);
}

// A stream that is being captured into a HIP graph takes launches and event fork / join only: an allocation or a table upload fails with
// "operation not permitted when stream is capturing", INVALIDATES the caller's capture and, as a failed rebuild, used to leave the ctx
// disabled.  A call that would have to (re)build under capture is refused before it touches anything (round 6; header, `stream`).
// (the legacy stream cannot be captured, and asking about it while another stream captures would itself invalidate that capture)
static bool stream_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
static const char *const kCaptureRefusal = "this call has to build device resources (the first call for an input size, a larger batch, a new output image), "
                                           "which a capturing stream does not permit: make the same call once outside the capture";

// Offsets and strides of a batch of two are taken modulo 2^64 (cfg.pair_submit: the second eye's image may lie BELOW the first one's): that is
// integer arithmetic on the address, not pointer arithmetic -- `p += huge` is undefined beyond the object (UBSan on the R,L job of
// tests/debug/thread_stress.c, round 6), the wrap of an unsigned sum is not.  The kernels add `i * stride` to a 64-bit base the same way.
template <class T>
static inline T *at_offset(T *p, size_t off) { return reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(p) + off); }

// a*b+c with two roundings, identical to the kernels' mad_unfused (separate statements)
static inline float mad2(float a, float b, float c)
{
    volatile float t = a * b;
    return t + c;
}

PostProcessor::PostProcessor(int device, const ovrfsr_config &cfg) : device_(device), cfg_(cfg) {}

// The tiles outside the radius of a masked pass are independent of the tiles touching it: they can run on the ctx's auxiliary
// stream beside the main kernel.  Whether that pays was re-measured in round 5 (profiles/r05_frame.txt):
//   * RGBA8 sources (outside_staged_kernel: LDS-staged, persistent, streams at ~5 TB/s): NO.  In order on the caller's stream the
//     step is faster at every batch size -- C2r 28.2 k against 27.9 k pairs/s at 128 images per call, 25.7 k against 24.3 k at 8 --
//     and one eye image per call, the reference's own call pattern (one Apply per Submit), takes 35 us of GPU time instead of 44:
//     the two cross-queue event waits of a fork / join pair cost ~10 us each, and two VALU- / HBM-hungry kernels sharing the CUs
//     finish no sooner than one after the other.  (Rounds 2-3 overlapped them because the outside-tile kernel of that time was
//     latency-bound; round 3 rebuilt it.)
//   * other sources (easu_outside_kernel / nis_outside_kernel: per-pixel, latency-bound, 8 workgroups per CU of spare issue slots
//     beside the fused kernel's 3): YES -- C5 13.9 k against 10.3 k pairs/s at 128 images per call, 12.8 k against 10.6 k at 2.
// `overlap` is that decision (OverlapOutside); OVRFSR_SERIAL=1 / 0 forces in-order / forked launches (diagnostic).
hipStream_t PostProcessor::Fork(hipStream_t user, bool overlap)
{
    static const int serial = [] { const char *e = std::getenv("OVRFSR_SERIAL"); return e && e[0] == '1' ? 1 : e && e[0] == '0' ? 0 : -1; }();
    if (serial == 1 || (serial < 0 && !overlap)) return user;
    if (!auxStream_) {
        // The auxiliary stream is a LOW-priority queue: what runs on it (the outside-tile kernel of half / float passes) fills the wave slots
        // the main kernel leaves, never the other way round -- C5 13.28 k / 13.30 k pairs/s at default priority, 13.32 k / 13.37 k low,
        // 11.26 k / 11.29 k high (round 5).  OVRFSR_AUX_PRIORITY=default|high: tuning
        static const int prio = [] { const char *e = std::getenv("OVRFSR_AUX_PRIORITY"); return !e ? 1 : e[0] == 'd' ? 0 : e[0] == 'h' ? 2 : 1; }();
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const hipError_t ce = stream_create(&auxStream_, prio, least, greatest);
        if (ce != hipSuccess ||
            event_create(&evFork_, hipEventDisableTiming) != hipSuccess ||
            event_create(&evJoin_, hipEventDisableTiming) != hipSuccess) {
            // no second stream: run everything in order on the caller's stream (and do not keep half of the set: a later Fork would use it)
            if (ce == hipSuccess && auxStream_) (void)hipStreamDestroy(auxStream_);
            if (evFork_) { (void)hipEventDestroy(evFork_); evFork_ = nullptr; }
            if (evJoin_) { (void)hipEventDestroy(evJoin_); evJoin_ = nullptr; }
            auxStream_ = nullptr;
            return user;
        }
    }
    (void)hipEventRecord(evFork_, user);
    (void)hipStreamWaitEvent(auxStream_, evFork_, 0);
    return auxStream_;
}

void PostProcessor::Join(hipStream_t user, hipStream_t aux)
{
    if (!auxStream_ || aux == user) return; // nothing was forked
    (void)hipEventRecord(evJoin_, auxStream_);
#ifndef OVRFSR_MUTATE_NO_JOIN /* mutation build (never shipped): proves tests/test_gpu_back_to_back.py notices a missing join edge */
    (void)hipStreamWaitEvent(user, evJoin_, 0);
#endif
}

PostProcessor::~PostProcessor()
{
    Reset(); // (frees a retired pair_submit image too)
    DeviceGuard guard(device_);
    if (auxStream_) (void)hipStreamDestroy(auxStream_);
    if (evFork_) (void)hipEventDestroy(evFork_);
    if (evJoin_) (void)hipEventDestroy(evJoin_);
    for (ProfileQuery &q : queries_) {
        if (q.start) (void)hipEventDestroy(q.start);
        if (q.end) (void)hipEventDestroy(q.end);
    }
}

bool PostProcessor::OverlapOutside(const ovrfsr_image &in) const
{
    // launch_easu_outside / launch_nis_outside take the LDS-staged kernel for RGBA8 sources when upscaling (outside_staged_ok)
    const bool staged = in.format == OVRFSR_FORMAT_RGBA8_UNORM && in.width <= outputWidth_ && in.height <= outputHeight_;
    return !staged;
}

int PostProcessor::Fail(int status, const std::string &what)
{
    lastError_ = what;
    return status;
}

void PostProcessor::Reset() { ResetKeeping(false); }

// keepRetired: the implicit Reset of a size change (Apply) must not free the ctx-owned image a just-flushed pair_submit eye was handed in
void PostProcessor::ResetKeeping(bool keepRetired)
{
    DeviceGuard guard(device_); // resources live on the ctx's device, whatever the caller has selected
    if (!keepRetired && retired_) { (void)hipFree(retired_); retired_ = nullptr; }
    havePending_ = false;       // a recorded first eye of cfg.pair_submit is dropped (header)
    lastApplyRecorded_ = false;
    if (!keepRetired) { pairFirstEye_ = -1; pairDefer_ = true; lastEye_ = -1; } // an explicit reset forgets the learned submission order
    enabled_ = true;
    initialized_ = false;
    if (swizzled_) (void)hipFree(swizzled_);
    swizzled_ = nullptr; swizzledBytes_ = 0;
    if (upscaled_) (void)hipFree(upscaled_);
    if (sharpened_) (void)hipFree(sharpened_);
    if (nisCoefDev_) (void)hipFree(nisCoefDev_);
    if (bilinDev_) (void)hipFree(bilinDev_);
    if (tileListDev_) (void)hipFree(tileListDev_);
    tileListDev_ = nullptr;
    tileRecDev_ = nullptr;
    spanRecDev_ = nullptr;
    nSpans_[0] = nSpans_[1] = 0;
    nInside_[0] = nInside_[1] = nOutside_[0] = nOutside_[1] = nRing_[0] = nRing_[1] = 0;
    nisCoefDev_ = nullptr;
    bilinDev_ = nullptr;
    bilinHost_.clear(); // never let taps of a previous configuration reach PrepareTileLists' footprint records
    bilYOff_ = 0;
    upscaled_ = sharpened_ = nullptr;
    upscaledBytes_ = sharpenedBytes_ = 0;
    lastSubmittedTexture_ = nullptr;
    outputTexture_ = ovrfsr_image{};
    eyeCount_ = 0;
    for (ProfileQuery &q : queries_) q.pending = false;
    lastQuery_ = -1;
    // summedGpuTime / countedQueries survive a Reset in the reference (plain members, PostProcessor.h:81-82): kept
}

int PostProcessor::SetConfig(const ovrfsr_config &cfg)
{
    // the reference's hotkeys mutate Config and then Reset() (PostProcessor.cpp:670-704)
    cfg_ = cfg;
    Reset();
    return OVRFSR_OK;
}

int PostProcessor::CheckImage(const ovrfsr_image *img, const char *name)
{
    if (!img || !img->data) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, std::string(name) + ": null image");
    if (img->format > OVRFSR_FORMAT_BGRA8_UNORM) return Fail(OVRFSR_ERR_UNSUPPORTED, std::string(name) + ": unknown format");
    const uint32_t tb = texel_bytes(img->format);
    if (img->width == 0 || img->height == 0 || img->width > 16384 || img->height > 16384)
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, std::string(name) + ": bad size");
    if (img->pitch_bytes < img->width * tb || (img->pitch_bytes % tb) != 0)
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, std::string(name) + ": bad pitch");
    // the fast paths address texels with 32-bit byte offsets off the image base (natural size: at most 16384^2 x 16 B = 4 GiB)
    if ((uint64_t)img->pitch_bytes * img->height > 0x100000000ull)
        return Fail(OVRFSR_ERR_UNSUPPORTED, std::string(name) + ": image spans more than 4 GiB (row pitch too large)");
    if ((reinterpret_cast<uintptr_t>(img->data) % tb) != 0)
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, std::string(name) + ": data not texel-aligned");
    return OVRFSR_OK;
}

int PostProcessor::EnsureBuffer(void **buf, size_t *have, size_t need)
{
    if (*have >= need) return OVRFSR_OK;
    if (capturing_) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, kCaptureRefusal);
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    hipError_t e = dev_malloc(buf, need);
    if (e != hipSuccess) return Fail(OVRFSR_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    *have = need;
    return OVRFSR_OK;
}

uint32_t PostProcessor::IntermediateFormat() const
{
    // the reference's upscaledTexture has the output format: R8G8B8A8_UNORM (PostProcessor.cpp:348 via :63-74).
    // Half-float pipelines (BASELINE C5) keep a half-float intermediate; quantize_intermediate=0 keeps fp32.
    if (!cfg_.quantize_intermediate) return OVRFSR_FORMAT_RGBA32F;
    // a 10-bit submission keeps 10-bit resources (DetermineOutputFormat, :63-74)
    return inputFormat_ == OVRFSR_FORMAT_RGBA8_UNORM || inputFormat_ == OVRFSR_FORMAT_BGRA8_UNORM ? OVRFSR_FORMAT_RGBA8_UNORM
         : inputFormat_ == OVRFSR_FORMAT_RGBA16F ? OVRFSR_FORMAT_RGBA16F
         : inputFormat_ == OVRFSR_FORMAT_RGB10A2_UNORM ? OVRFSR_FORMAT_RGB10A2_UNORM : OVRFSR_FORMAT_RGBA32F;
}

void PostProcessor::PrepareUpscalingResources()
{
    easu_con(easuCon_, (float)inputWidth_, (float)inputHeight_, (float)inputWidth_, (float)inputHeight_,
             (float)outputWidth_, (float)outputHeight_);
    float sx, sy, cx, cy;
    std::memcpy(&sx, &easuCon_[0], 4); std::memcpy(&sy, &easuCon_[1], 4);
    std::memcpy(&cx, &easuCon_[2], 4); std::memcpy(&cy, &easuCon_[3], 4);
    // LDS footprint of one 32x32 output tile: f-texel of first and last pixel, +1/+2 apron.  `pairs`: the product EASU kernel resolves rows in
    // PAIRS (ly, ly + 1), ly even, and evaluates the second pixel of the last pair even when its row lies behind the image (only its store is
    // guarded): the footprint covers that row too.  (Until round 6 it did not: where the LAST, partial tile row defines the extent -- an image of
    // a single tile row with an odd height -- the discarded pixel read up to two cell rows past the colour / analysis planes, into the next
    // plane of the same workgroup.  Found by the fuzz seeds run against the checked build, seed 162312: profiles/r06_bounds.txt.)
    auto extent = [](uint32_t outN, int tile, float s, float c, bool pairs) {
        int best = 0;
        for (uint32_t o0 = 0; o0 < outN; o0 += tile) {
            uint32_t o1 = o0 + tile - 1 < outN ? o0 + tile - 1 : outN - 1;
            if (pairs) o1 |= 1u; // the partner row of the last pair (inside the tile: tile heights are even)
            int f0 = (int)std::floor(mad2((float)o0, s, c)), f1 = (int)std::floor(mad2((float)o1, s, c));
            best = f1 - f0 + 4 > best ? f1 - f0 + 4 : best;
        }
        return best;
    };
    cellsW_ = extent(outputWidth_, kTileW, sx, cx, false);
    cellsH_ = extent(outputHeight_, kTileH, sy, cy, true);
    // fused kernel: EASU runs on the tile plus a 1-pixel ring, origin at pixel (o0 - 1)
    auto extentRing = [](uint32_t outN, int tile, float s, float c) {
        int best = 0;
        for (uint32_t o0 = 0; o0 < outN; o0 += tile) {
            uint32_t o1 = o0 + tile < outN ? o0 + tile : outN - 1;
            int f0 = (int)std::floor(mad2((float)o0 - 1.0f, s, c)), f1 = (int)std::floor(mad2((float)o1, s, c));
            best = f1 - f0 + 4 > best ? f1 - f0 + 4 : best;
        }
        return best;
    };
    fusedCellsW_ = extentRing(outputWidth_, kTileW, sx, cx);
    fusedCellsH_ = extentRing(outputHeight_, kTileH, sy, cy);
}

void PostProcessor::PrepareSharpeningResources()
{
    float s = cfg_.sharpness;
    s = s < 1.0f ? s : 1.0f; // AClampF1(x,0,1) = max(0,min(x,1)), PostProcessor.cpp:420
    s = s > 0.0f ? s : 0.0f;
    rcas_con(rcasCon_, 2.f - 2 * s);
    rcasCon_[3] = cfg_.debug_mode ? 1u : 0u; // :430
}

int PostProcessor::PrepareResources(const ovrfsr_image &submitted)
{
    inputWidth_ = submitted.width;
    inputHeight_ = submitted.height;
    inputFormat_ = submitted.format;
    ovrfsr_image in = submitted; // what the kernels will see: a BGRA8 submission is re-ordered to RGBA8 first (ApplyPostProcess)
    if (in.format == OVRFSR_FORMAT_BGRA8_UNORM) in.format = OVRFSR_FORMAT_RGBA8_UNORM;
    uint32_t ow = 0, oh = 0;
    if (ovrfsr_output_size(&cfg_, in.width, in.height, &ow, &oh) != OVRFSR_OK || ow == 0 || oh == 0)
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "output size is zero or beyond 16384 texels (render_scale must be finite and > 0)");
    outputWidth_ = ow;
    outputHeight_ = oh;

    const bool explicitSize = cfg_.out_width != 0 && cfg_.out_height != 0;
    const bool scaleNotOne = explicitSize ? (ow != in.width || oh != in.height) : (cfg_.render_scale != 1.f);
    doUpscale_ = cfg_.fsr_enabled && scaleNotOne;                       // :586
    doSharpen_ = cfg_.fsr_enabled && (!cfg_.use_nis || !scaleNotOne);   // :591
    if (cfg_.stage_mask == 1) doSharpen_ = false;                       // "EASU-only" (BASELINE C1)
    if (cfg_.stage_mask == 2) doUpscale_ = false;
    if (cfg_.stage_mask < 0 || cfg_.stage_mask > 2) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "bad stage_mask");
    if (!doUpscale_ && (ow != in.width || oh != in.height))
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "sharpen-only needs output size == input size");
    if (cfg_.precision != OVRFSR_PRECISION_FP32 && cfg_.precision != OVRFSR_PRECISION_FP32_STRICT)
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "unknown precision");

    for (int eye = 0; eye < 2; ++eye) {
        mask_constants(centre_[eye], radius_, ow, oh, cfg_.proj_centre, cfg_.radius, textureContainsOnlyOneEye_ ? 1 : 0, eye);
        const uint32_t gw = cfg_.use_nis ? 32u : 16u, gh = cfg_.use_nis ? (scaleNotOne ? 24u : 32u) : 16u;
        maskMode_[eye] = classify_mask(centre_[eye], radius_[1], ow, oh, gw, gh);
    }
    if (cfg_.use_nis) {
        int rc = PrepareNisResources();
        if (rc != OVRFSR_OK) return rc;
    } else if (doUpscale_) {
        PrepareUpscalingResources();
        const size_t lds = easu_lds_bytes(cfg_.precision, (int)in.format, cellsW_, cellsH_);
        if (lds > 64 * 1024) return Fail(OVRFSR_ERR_UNSUPPORTED, "scale ratio needs more LDS than one tile may use");
    }
    {
        // o/outW, o/outH of the bilinear fallback as a multiply and two FMAs: verify against IEEE division for every o
        auto check = [](uint32_t n, float &rn) {
            volatile float r = 1.0f / (float)n;
            rn = r;
            for (uint32_t o = 0; o < n; ++o) {
                const float q0 = (float)o * rn;
                const float rem = std::fma(-q0, (float)n, (float)o);
                volatile float want = (float)o / (float)n;
                if (std::fma(rem, rn, q0) != want) return false;
            }
            return true;
        };
        const bool okW = check(ow, rcpOut_[0]), okH = check(oh, rcpOut_[1]);
        rcpExact_ = okW && okH;
    }
    if (doUpscale_) {
        // column / row taps of the bilinear fallback / NIS DirectCopy (SampleLevel at pos/outSize, 8-bit sub-texel snap): same IEEE
        // operations as fsr_device.inc's bilinear_uv / fixed8, evaluated once per column and row instead of per pixel
        // Layout: [ow column taps | copies of the last column tap up to a multiple of the tile width | oh row taps | 64 spare entries].
        // The staged outside-tile kernel loads the column taps of every pixel QUAD of a 32-wide tile (outside_staged_kernel::issue_taps);
        // the quads on and behind the last column must find VALID taps -- their pixels are never stored, but the taps index the kernel's
        // LDS plane.  (Until round 6
        // the row taps followed the column taps directly and that quad read row taps as column taps: out-of-plane LDS reads whose values
        // were discarded -- found by the checked build, profiles/r06_bounds.txt.)
        std::vector<BilinTap> &taps = bilinHost_;
        bilYOff_ = (ow + (uint32_t)kTileW - 1u) & ~((uint32_t)kTileW - 1u);
        taps.assign((size_t)bilYOff_ + oh + 64, BilinTap{0, 0.0f});
        auto fill = [](BilinTap *t, uint32_t outN, uint32_t inN) {
            for (uint32_t o = 0; o < outN; ++o) {
                volatile float u = (float)o / (float)outN;
                const float tt = mad2(u, (float)inN, -0.5f);
                const float s = std::floor(mad2(tt, 256.0f, 0.5f));
                volatile float q = s * (1.0f / 256.0f);
                const float f = std::floor(q);
                t[o].i0 = (int32_t)f;
                t[o].frac = mad2(f, -256.0f, s) * (1.0f / 256.0f);
            }
        };
        fill(taps.data(), ow, in.width);
        for (uint32_t o = ow; o < bilYOff_; ++o) taps[o] = taps[ow - 1];
        fill(taps.data() + bilYOff_, oh, in.height);
        // largest [first tap, last tap + 1] span of a tile: the LDS plane of the staged outside-tile kernel
        auto span = [](const BilinTap *t, uint32_t outN, uint32_t tile) {
            int best = 2;
            for (uint32_t o0 = 0; o0 < outN; o0 += tile) {
                const uint32_t o1 = o0 + tile - 1 < outN ? o0 + tile - 1 : outN - 1;
                best = std::max(best, t[o1].i0 + 2 - t[o0].i0);
            }
            return (uint32_t)best;
        };
        outsideCols_ = span(taps.data(), ow, 32);
        outsideRows_[0] = span(taps.data() + bilYOff_, oh, 32);
        outsideRows_[1] = span(taps.data() + bilYOff_, oh, 24);
        hipError_t e = dev_malloc(reinterpret_cast<void **>(&bilinDev_), taps.size() * sizeof(BilinTap));
        if (e == hipSuccess) e = hipMemcpy(bilinDev_, taps.data(), taps.size() * sizeof(BilinTap), hipMemcpyHostToDevice);
        if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("bilinear tap tables: ") + hipGetErrorString(e));
    }
    if (doUpscale_ && !cfg_.use_nis && cfg_.precision == OVRFSR_PRECISION_FP32 &&
        (maskMode_[0] == MASK_MIXED || maskMode_[1] == MASK_MIXED)) {
        int rc = PrepareTileLists(kTileW, kTileH, 16, 16);
        if (rc != OVRFSR_OK) return rc;
    }
    if (doUpscale_ && cfg_.use_nis && cfg_.precision == OVRFSR_PRECISION_FP32 &&
        (maskMode_[0] == MASK_MIXED || maskMode_[1] == MASK_MIXED)) {
        int rc = PrepareTileLists(32, 24, 32, 24); // NVScaler: one workgroup per 32x24 mask group
        if (rc != OVRFSR_OK) return rc;
    }
    if (doSharpen_ && !cfg_.use_nis) PrepareSharpeningResources();
    // one launch with the intermediate in LDS only on request: on this chip both stages are VALU-bound and the ring
    // recompute costs more than the HBM round trip saves, so auto (-1) means two kernels
    useFused_ = false;
    // auto: masked product-build pipelines run fused + mask-sorted (most of their pixels are plain bilinear copies, and tiles outside the
    // radius need no intermediate at all); unmasked ones stay two-pass (VALU-bound, the ring recompute costs 9 %)
    const bool tenBit = in.format == OVRFSR_FORMAT_RGB10A2_UNORM; // two-kernel pipeline only (header)
    if (tenBit && cfg_.fused == 1) return Fail(OVRFSR_ERR_UNSUPPORTED, "the fused kernel is not built for RGB10A2 images");
    const bool autoFused = !tenBit && cfg_.fused == -1 && tileListDev_ != nullptr && fusedCellsW_ <= 40 &&
                           fused_lds_bytes(cfg_.precision, (int)in.format, (int)IntermediateFormat(), fusedCellsW_, fusedCellsH_) <= kFusedLdsMax;
    // auto on a masked product-build EASU+RCAS pipeline: the two-pass kernels on the tiles touching the radius, tiles
    // outside written in final form (ApplySorted); cfg.fused = 1 keeps the single fused kernel on those tiles
    // Measured: with 4-byte pixels the sorted two-pass form wins (C2 shape, radius 0.5: +13 %); with 8/16-byte
    // pixels the outside kernel dominates the frame, and the three dependent launches of the sorted form lose to the
    // fused kernel (C5: -15 %), so those keep it.
    useSorted_ = cfg_.fused == -1 && tileListDev_ != nullptr && doUpscale_ && doSharpen_ && !cfg_.use_nis &&
                 in.format == OVRFSR_FORMAT_RGBA8_UNORM && IntermediateFormat() == OVRFSR_FORMAT_RGBA8_UNORM;
    if ((cfg_.fused == 1 || (autoFused && !useSorted_)) && doUpscale_ && doSharpen_ && !cfg_.use_nis) {
        const bool pitchOk = cfg_.precision == OVRFSR_PRECISION_FP32_STRICT || fusedCellsW_ <= 40;
        if (!pitchOk || fused_lds_bytes(cfg_.precision, (int)in.format, (int)IntermediateFormat(), fusedCellsW_, fusedCellsH_) > kFusedLdsMax)
            return Fail(OVRFSR_ERR_UNSUPPORTED, "fused kernel: tile footprint does not fit LDS at this scale");
        useFused_ = true;
    }
    if (cfg_.debug_mode) {
        // every slot is checked on its own: a creation that failed half-way through the ring (found by fault injection, round 6: slot 0 existed,
        // a later one did not, and the rebuild after reset skipped the whole ring -- hipEventRecord on a null event) is completed by the next build
        for (ProfileQuery &q : queries_)
            if ((!q.start && event_create(&q.start, hipEventDefault) != hipSuccess) || (!q.end && event_create(&q.end, hipEventDefault) != hipSuccess))
                return Fail(OVRFSR_ERR_HIP, "hipEventCreate failed");
    }
    initialized_ = true;
    return OVRFSR_OK;
}

// Block b of a launch runs on XCD b % 8, each XCD with a private L2: entry b of a work list of n row-major items is item
// xcd_source(b, n), which hands every XCD a contiguous raster run of the list (the n % 8 trailing entries keep their place).
static inline uint32_t xcd_source(uint32_t b, uint32_t n)
{
    const uint32_t full = n & ~7u;
    return b < full ? (b & 7u) * (full >> 3) + (b >> 3) : b;
}

// The radius mask is static per eye, so the tiles are sorted once on the host: tiles with at least one 16x16 group
// inside the radius (EASU kernel, LDS-staged) and tiles entirely outside (bilinear only, LDS-free kernel).
int PostProcessor::PrepareTileLists(uint32_t tileW, uint32_t tileH, uint32_t groupW, uint32_t groupH)
{
    const uint32_t tx = (outputWidth_ + tileW - 1) / tileW, ty = (outputHeight_ + tileH - 1) / tileH;
    const uint32_t gpx = tileW / groupW, gpy = tileH / groupH; // mask groups per tile
    std::vector<uint32_t> lists;
    std::vector<uint32_t> in[2], outl[2];
    for (int eye = 0; eye < 2; ++eye) {
        for (uint32_t t = 0; t < tx * ty; ++t) {
            const uint32_t tyi = t / tx, txi = t - tyi * tx;
            bool any = false;
            for (uint32_t g = 0; g < gpx * gpy && !any; ++g) {
                const uint32_t gx = gpx * txi + (g % gpx), gy = gpy * tyi + (g / gpx);
                const uint32_t cx = gx * groupW + groupW / 2, cy = gy * groupH + groupH / 2;
                const uint32_t ax = centre_[eye][0] - cx, ay = centre_[eye][1] - cy, bx = centre_[eye][2] - cx, by = centre_[eye][3] - cy;
                any = (ax * ax + ay * ay <= radius_[1]) || (bx * bx + by * by <= radius_[1]);
            }
            (any ? in[eye] : outl[eye]).push_back(t);
        }
    }
    std::vector<uint32_t> ring[2];
    for (int eye = 0; eye < 2; ++eye) {
        std::vector<uint8_t> isIn(tx * ty, 0);
        for (uint32_t t : in[eye]) isIn[t] = 1;
        for (uint32_t t : outl[eye]) {
            const uint32_t tyi = t / tx, txi = t - tyi * tx;
            const bool adj = (txi > 0 && isIn[t - 1]) || (txi + 1 < tx && isIn[t + 1]) || (tyi > 0 && isIn[t - tx]) || (tyi + 1 < ty && isIn[t + tx]);
            if (adj) ring[eye].push_back(t);
        }
    }
    listsShared_ = in[0] == in[1];
    // RCAS on the mask-sorted form (RGBA8): per 32-row band, runs of adjacent tiles touching the radius, cut into the DPP
    // kernel's 62-column segments (rcas_dpp_kernel<.., true>): {x0 | tileY << 16, xEnd}
    std::vector<uint32_t> spans;
    nSpans_[0] = nSpans_[1] = 0;
    if (tileW == kTileW && tileH == kRcasDppTileH && outputWidth_ < 65536u && ty < 65536u) {
        for (int eye = 0; eye < 2; ++eye) {
            spanOff_[eye] = spans.size() / 2;
            for (size_t i = 0; i < in[eye].size();) { // in[] is row-major here
                size_t j = i;
                while (j + 1 < in[eye].size() && in[eye][j + 1] == in[eye][j] + 1 && (in[eye][j + 1] / tx) == (in[eye][i] / tx)) ++j;
                const uint32_t tyi = in[eye][i] / tx, xa = (in[eye][i] - tyi * tx) * tileW;
                const uint32_t xb = std::min((in[eye][j] - tyi * tx + 1) * tileW, outputWidth_);
                for (uint32_t x0 = xa; x0 < xb; x0 += kRcasDppTileW) {
                    spans.push_back(x0 | (tyi << 16));
                    spans.push_back(std::min(x0 + (uint32_t)kRcasDppTileW, xb));
                }
                i = j + 1;
            }
            nSpans_[eye] = (uint32_t)(spans.size() / 2 - spanOff_[eye]);
            // every XCD a contiguous raster run of segments (xcd_source), so that the 128-byte lines two neighbouring
            // segments share, and the rows two bands share, are fetched once
            const uint32_t n = nSpans_[eye];
            std::vector<uint32_t> r(2 * (size_t)n);
            for (uint32_t b = 0; b < n; ++b) {
                const size_t src = spanOff_[eye] + xcd_source(b, n);
                r[2 * (size_t)b] = spans[2 * src]; r[2 * (size_t)b + 1] = spans[2 * src + 1];
            }
            std::copy(r.begin(), r.end(), spans.begin() + 2 * spanOff_[eye]);
        }
    }
    auto xcd_order = [](std::vector<uint32_t> &v) {
        const uint32_t n = (uint32_t)v.size();
        std::vector<uint32_t> r(n);
        for (uint32_t b = 0; b < n; ++b) r[b] = v[xcd_source(b, n)];
        v.swap(r);
    };
    for (int eye = 0; eye < 2; ++eye) {
        xcd_order(in[eye]); xcd_order(outl[eye]);
        nInside_[eye] = (uint32_t)in[eye].size(); nOutside_[eye] = (uint32_t)outl[eye].size();
        // inside | ring | outside: the ring tiles follow the inside tiles so that ONE EASU launch over nInside + nRing entries
        // also writes the bilinear intermediate of the ring (its all-outside path), while RCAS walks the first nInside only
        listOffInside_[eye] = lists.size(); lists.insert(lists.end(), in[eye].begin(), in[eye].end());
        nRing_[eye] = (uint32_t)ring[eye].size();
        listOffRing_[eye] = lists.size(); lists.insert(lists.end(), ring[eye].begin(), ring[eye].end());
        listOffOutside_[eye] = lists.size(); lists.insert(lists.end(), outl[eye].begin(), outl[eye].end());
    }
    if (lists.empty()) return OVRFSR_OK;
    // One record per list entry for the persistent outside-tile kernel (outside_staged_kernel): tile origin, footprint origin
    // (first column / row tap) and extent ([first tap, last tap + 1], what the kernel used to fetch through a chain of
    // dependent scalar loads: tile index -> tap tables)
    std::vector<uint32_t> recs(lists.size() * 4, 0u);
    if (bilinHost_.size() >= (size_t)bilYOff_ + outputHeight_) {
        const BilinTap *bx = bilinHost_.data(), *by = bilinHost_.data() + bilYOff_;
        const uint32_t rowsCap = outsideRows_[tileH == 24 ? 1 : 0];
        for (size_t i = 0; i < lists.size(); ++i) {
            const uint32_t t = lists[i], tyi = t / tx, txi = t - tyi * tx;
            const uint32_t ox0 = txi * tileW, oy0 = tyi * tileH;
            const int X0 = bx[ox0].i0, Y0 = by[oy0].i0;
            const int colsN = std::min<int>((int)outsideCols_, bx[std::min(ox0 + tileW - 1, outputWidth_ - 1)].i0 + 2 - X0);
            const int rowsN = std::min<int>((int)rowsCap, by[std::min(oy0 + tileH - 1, outputHeight_ - 1)].i0 + 2 - Y0);
            recs[4 * i + 0] = ox0 | (oy0 << 16);
            recs[4 * i + 1] = (uint32_t)(X0 + 1) | ((uint32_t)(Y0 + 1) << 16);
            recs[4 * i + 2] = (uint32_t)colsN | ((uint32_t)rowsN << 8);
        }
    }
    const size_t listDwords = (lists.size() + 3) & ~(size_t)3; // the records follow the lists, 16-byte aligned
    hipError_t e = dev_malloc(reinterpret_cast<void **>(&tileListDev_), (listDwords + recs.size() + spans.size()) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(tileListDev_, lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        tileRecDev_ = tileListDev_ + listDwords;
        e = hipMemcpy(tileRecDev_, recs.data(), recs.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !spans.empty()) {
        spanRecDev_ = tileRecDev_ + recs.size();
        e = hipMemcpy(spanRecDev_, spans.data(), spans.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("tile lists: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

int PostProcessor::PrepareNisResources()
{
    std::memset(&nisConfig_, 0, sizeof(nisConfig_));
    // NVScalerUpdateConfig (scale != 1) or NVSharpenUpdateConfig (out == in), PostProcessor.cpp:308,:433.
    // The reference ignores a `false` result and dispatches with a half-filled block; that is undefined
    // there, so it is an error here.
    if (!nis_scaler_config(&nisConfig_, cfg_.sharpness, inputWidth_, inputHeight_, outputWidth_, outputHeight_))
        return Fail(OVRFSR_ERR_UNSUPPORTED, "NIS scales 1x..2x only (NVScalerUpdateConfig returned false)");
    nisConfig_.reserved1 = cfg_.debug_mode ? 1.f : 0.f; // :309
    if (doUpscale_) {
        auto extent = [](uint32_t outN, int blk, float s) {
            int best = 0;
            for (uint32_t o0 = 0; o0 < outN; o0 += blk) {
                uint32_t o1 = o0 + blk - 1 < outN ? o0 + blk - 1 : outN - 1;
                int f0 = (int)std::floor(mad2(0.5f + (float)o0, s, -0.5f)), f1 = (int)std::floor(mad2(0.5f + (float)o1, s, -0.5f));
                best = f1 - f0 + 7 > best ? f1 - f0 + 7 : best; // 6-tap support (+2/+3) and the edge-map ring (+1)
            }
            return best;
        };
        nisCellsW_ = extent(outputWidth_, 32, nisConfig_.kScaleX);
        nisCellsH_ = extent(outputHeight_, 24, nisConfig_.kScaleY);
        if (nis_pitch(nisCellsW_) == 0 || nis_scaler_lds_bytes(nisCellsW_, nisCellsH_) > 64 * 1024)
            return Fail(OVRFSR_ERR_UNSUPPORTED, "NIS tile does not fit LDS");
    }
    hipError_t e = dev_malloc(reinterpret_cast<void **>(&nisCoefDev_), 2 * 512 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(nisCoefDev_, nis_coef_scale(), 512 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(nisCoefDev_ + 512, nis_coef_usm(), 512 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("NIS coefficient upload: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

void PostProcessor::FillNis(NisArgs &a, int firstEye, int alternate) const
{
    const NisConstants &c = nisConfig_;
    a.kDetectRatio = c.kDetectRatio; a.kDetectThres = c.kDetectThres; a.kMinContrastRatio = c.kMinContrastRatio; a.kRatioNorm = c.kRatioNorm;
    a.kContrastBoost = c.kContrastBoost; a.kEps = c.kEps; a.kSharpStartY = c.kSharpStartY; a.kSharpScaleY = c.kSharpScaleY;
    a.kSharpStrengthMin = c.kSharpStrengthMin; a.kSharpStrengthScale = c.kSharpStrengthScale;
    a.kSharpLimitMin = c.kSharpLimitMin; a.kSharpLimitScale = c.kSharpLimitScale;
    a.kScaleX = c.kScaleX; a.kScaleY = c.kScaleY; a.kDstNormX = c.kDstNormX; a.kDstNormY = c.kDstNormY;
    a.kSrcNormX = c.kSrcNormX; a.kSrcNormY = c.kSrcNormY;
    a.reserved1 = c.reserved1;
    FillMask(a.m, firstEye, alternate);
    a.tileRec = nullptr;
    a.coefScale = nisCoefDev_;
    a.coefUsm = nisCoefDev_ + 512;
    a.cellsW = nisCellsW_; a.cellsH = nisCellsH_;
    a.bilX = bilinDev_; a.bilY = bilinDev_ ? bilinDev_ + bilYOff_ : nullptr;
    a.outsideCols = outsideCols_; a.outsideRows = outsideRows_[1];
}

void PostProcessor::FillMask(MaskArgs &m, int firstEye, int alternate) const
{
    std::memcpy(m.centre, centre_, sizeof(centre_));
    m.r2 = radius_[1];
    m.mode[0] = maskMode_[0];
    m.mode[1] = maskMode_[1];
    m.first_eye = (uint32_t)(firstEye & 1);
    m.alternate = alternate ? 1u : 0u;
}

static BatchView make_view(const ovrfsr_image &in, size_t inStride, const ovrfsr_image &out, size_t outStride)
{
    BatchView v;
    v.in = static_cast<const uint8_t *>(in.data);
    v.out = static_cast<uint8_t *>(out.data);
    v.in_stride = inStride;
    v.out_stride = outStride;
    v.in_pitch = in.pitch_bytes;
    v.out_pitch = out.pitch_bytes;
    v.inW = (int32_t)in.width; v.inH = (int32_t)in.height;
    v.outW = (int32_t)out.width; v.outH = (int32_t)out.height;
    return v;
}

int PostProcessor::ApplyUpscaling(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                                  const ovrfsr_image &out, size_t outStride, hipStream_t stream)
{
    if (cfg_.use_nis) {
        NisArgs na;
        na.v = make_view(in, inStride, out, outStride);
        FillNis(na, firstEye, alternate);
        na.tilesX = (out.width + 31) / 32;   // Dispatch(ceil(outW/32), ceil(outH/24)), :397
        na.tilesY = (out.height + 23) / 24;
        na.tileList = nullptr;
        hipError_t e = hipSuccess;
        if (!tileListDev_) {
            e = launch_nis_scaler(cfg_.precision, (int)in.format, (int)out.format, na, n, stream);
        } else {
            EyePass passes[2];
            const int np = EyePasses(n, firstEye, alternate, inStride, outStride, passes);
            hipStream_t aux = Fork(stream, OverlapOutside(in));
            for (int p = 0; p < np && e == hipSuccess; ++p) {
                NisArgs b = na;
                const EyePass &ps = passes[p];
                b.v.in = at_offset(b.v.in, ps.inOff); b.v.out = at_offset(b.v.out, ps.outOff); b.v.in_stride = ps.inStride; b.v.out_stride = ps.outStride;
                if (ps.split) { b.m.first_eye = (uint32_t)ps.eye; b.m.alternate = 0; }
                if (nInside_[ps.eye]) {
                    b.tileList = tileListDev_ + listOffInside_[ps.eye];
                    e = launch_nis_scaler(cfg_.precision, (int)in.format, (int)out.format, b, ps.cnt, stream, nInside_[ps.eye]);
                }
                if (e == hipSuccess && nOutside_[ps.eye]) {
                    b.tileList = tileListDev_ + listOffOutside_[ps.eye];
                    b.tileRec = tileRecDev_ + 4 * listOffOutside_[ps.eye];
                    e = launch_nis_outside((int)in.format, (int)out.format, b, nOutside_[ps.eye], ps.cnt, aux);
                }
            }
            Join(stream, aux);
        }
        if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("NVScaler launch: ") + hipGetErrorString(e));
        return OVRFSR_OK;
    }
    EasuArgs a;
    FillEasu(a, in, inStride, out, outStride, firstEye, alternate);
    hipError_t e = hipSuccess;
    if (!tileListDev_) {
        e = launch_easu(cfg_.precision, (int)in.format, (int)out.format, a, n, stream);
    } else {
        EyePass passes[2];
        const int np = EyePasses(n, firstEye, alternate, inStride, outStride, passes);
        hipStream_t aux = Fork(stream, OverlapOutside(in));
        for (int p = 0; p < np && e == hipSuccess; ++p) {
            EasuArgs b = a;
            const EyePass &ps = passes[p];
            b.v.in = at_offset(b.v.in, ps.inOff); b.v.out = at_offset(b.v.out, ps.outOff); b.v.in_stride = ps.inStride; b.v.out_stride = ps.outStride;
            if (ps.split) { b.m.first_eye = (uint32_t)ps.eye; b.m.alternate = 0; }
            if (nInside_[ps.eye]) {
                b.tileList = tileListDev_ + listOffInside_[ps.eye];
                e = launch_easu(cfg_.precision, (int)in.format, (int)out.format, b, ps.cnt, stream, nInside_[ps.eye]);
            }
            if (e == hipSuccess && nOutside_[ps.eye]) {
                b.tileList = tileListDev_ + listOffOutside_[ps.eye];
                b.tileRec = tileRecDev_ + 4 * listOffOutside_[ps.eye];
                e = launch_easu_outside((int)in.format, -1, (int)out.format, b, nOutside_[ps.eye], ps.cnt, aux);
            }
        }
        Join(stream, aux);
    }
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("EASU launch: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

void PostProcessor::FillEasu(EasuArgs &a, const ovrfsr_image &in, size_t inStride, const ovrfsr_image &out, size_t outStride,
                             int firstEye, int alternate) const
{
    a.v = make_view(in, inStride, out, outStride);
    std::memcpy(&a.sx, &easuCon_[0], 4); std::memcpy(&a.sy, &easuCon_[1], 4);
    std::memcpy(&a.cx, &easuCon_[2], 4); std::memcpy(&a.cy, &easuCon_[3], 4);
    FillMask(a.m, firstEye, alternate);
    a.cellsW = cellsW_; a.cellsH = cellsH_;
    a.bilX = bilinDev_; a.bilY = bilinDev_ + bilYOff_;
    a.tileList = nullptr;
    a.tileRec = nullptr;
    a.tieHalfMin = TieHalfMin();
    a.ringStrips = 0;
    a.debug = rcasCon_[3];
    a.rcpOutW = rcpOut_[0]; a.rcpOutH = rcpOut_[1]; a.rcpExact = rcpExact_ ? 1u : 0u;
    a.outsideCols = outsideCols_; a.outsideRows = outsideRows_[0];
    a.tilesX = (out.width + kTileW - 1) / kTileW;   // the reference dispatches 16x16 groups (PostProcessor.cpp:399);
    a.tilesY = (out.height + kTileH - 1) / kTileH;  // a tile here is 2x2 of those
}

// Near-tie guard of a half-float intermediate: the smallest value whose flipped half rounding could exceed the 1e-3 tolerance
// behind RCAS.  RCAS's gain on its centre tap is at most 1 / (1 - 4 * 0.1875 * sharp) (lobe >= -FSR_RCAS_LIMIT * sharp,
// ffx_fsr1.h:654,757-765); a half in [b, 2b) moves in steps of b * 2^-10.  Binades whose step times that gain stays under
// 9e-4 are left alone; +inf (guard off) when no sharpening pass follows the upscale.
float PostProcessor::TieHalfMin() const
{
#ifdef OVRFSR_TIE_AUDIT
    // AUDIT builds only (round 6; the shipped library never reads it: a stray environment variable must not change product numerics):
    // OVRFSR_TIE_HALF_MIN=<x> guards half stores from x upwards whatever follows the upscale -- lets tools/debug/tie_audit.py and
    // tests/test_gpu_formats.py read the guarded half intermediate directly as the output of an EASU-only pass.  Read once.
    static const float forced = [] { const char *e = std::getenv("OVRFSR_TIE_HALF_MIN"); return e ? (float)std::atof(e) : 0.0f; }();
    if (forced > 0.0f) return forced;
#endif
    if (!(doUpscale_ && doSharpen_) || cfg_.use_nis) return INFINITY;
    float sharp;
    std::memcpy(&sharp, &rcasCon_[0], 4);
    const float gain = 1.0f / (1.0f - 0.75f * sharp);
    float b = 1.0f / 16384.0f; // half's smallest normal binade
    while (b < 65536.0f && gain * b * (1.0f / 1024.0f) < 9e-4f) b *= 2.0f;
    return b;
}

// Mask-sorted launches use per-eye tile lists.  Both eyes share the lists when their mask centres coincide;
// otherwise the images of each eye form their own stride-2 sub-batch (images p, p+2, ... have the same eye).
int PostProcessor::EyePasses(uint32_t n, int firstEye, int alternate, size_t inStride, size_t outStride, EyePass out[2]) const
{
    const bool twoEyes = alternate && n > 1 && !listsShared_;
    if (!twoEyes) {
        out[0] = EyePass{firstEye & 1, n, 0, 0, inStride, outStride, false};
        return 1;
    }
    for (int p = 0; p < 2; ++p)
        out[p] = EyePass{(firstEye & 1) ^ p, (n - p + 1) / 2, (size_t)p * inStride, (size_t)p * outStride, 2 * inStride, 2 * outStride, true};
    return 2;
}

// Masked EASU+RCAS, product build.  Main stream: EASU on the tiles touching the radius -> intermediate; the bilinear
// intermediate of the outside tiles 4-adjacent to them (RCAS taps reach one pixel across a tile edge); RCAS on the same
// tile list.  Auxiliary stream: every outside tile in final form.  The intermediate buffer is only ever touched at
// inside + ring tiles.
int PostProcessor::ApplySorted(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                               const ovrfsr_image &out, size_t outStride, hipStream_t stream)
{
    ovrfsr_image mid;
    mid.width = outputWidth_; mid.height = outputHeight_;
    mid.format = IntermediateFormat();
    mid.pitch_bytes = outputWidth_ * texel_bytes(mid.format);
    const size_t midStride = (size_t)mid.pitch_bytes * outputHeight_;
    int rc = EnsureBuffer(&upscaled_, &upscaledBytes_, midStride * n);
    if (rc != OVRFSR_OK) return rc;
    mid.data = upscaled_;

    EasuArgs toMid, toOut;
    FillEasu(toMid, in, inStride, mid, midStride, firstEye, alternate);
    FillEasu(toOut, in, inStride, out, outStride, firstEye, alternate);
    RcasArgs ra;
    ra.v = make_view(mid, midStride, out, outStride);
    std::memcpy(&ra.sharp, &rcasCon_[0], 4);
    ra.debug = rcasCon_[3];
    FillMask(ra.m, firstEye, alternate);
    ra.tilesX = toOut.tilesX; ra.tilesY = toOut.tilesY;
    ra.tileList = nullptr;
    ra.spanRec = nullptr; ra.nSpans = 0;

    EyePass passes[2], midPasses[2];
    const int np = EyePasses(n, firstEye, alternate, inStride, outStride, passes);
    EyePasses(n, firstEye, alternate, midStride, midStride, midPasses); // same split, strides of the intermediate
    hipError_t e = hipSuccess;
    hipStream_t aux = Fork(stream, OverlapOutside(in));
    for (int p = 0; p < np && e == hipSuccess; ++p) {
        const EyePass &ps = passes[p];
        const EyePass &ms = midPasses[p];
        EasuArgs em = toMid, eo = toOut;
        RcasArgs rb = ra;
        em.v.in = at_offset(em.v.in, ps.inOff); em.v.in_stride = ps.inStride; em.v.out = at_offset(em.v.out, ms.outOff); em.v.out_stride = ms.outStride;
        eo.v.in = at_offset(eo.v.in, ps.inOff); eo.v.in_stride = ps.inStride; eo.v.out = at_offset(eo.v.out, ps.outOff); eo.v.out_stride = ps.outStride;
        rb.v.in = at_offset(rb.v.in, ms.inOff); rb.v.in_stride = ms.inStride; rb.v.out = at_offset(rb.v.out, ps.outOff); rb.v.out_stride = ps.outStride;
        if (ps.split) {
            em.m.first_eye = eo.m.first_eye = rb.m.first_eye = (uint32_t)ps.eye;
            em.m.alternate = eo.m.alternate = rb.m.alternate = 0;
        }
        const int eye = ps.eye;
        // launch order matters for how the two hardware queues share the chip: the VALU-bound kernel first.  The EASU launch
        // covers the inside tiles AND the ring (outside tiles 4-adjacent to them, listed right behind: RCAS taps reach one
        // pixel across a tile edge); its all-outside path writes their bilinear intermediate, the same bytes the separate
        // ring launch of rounds 1-2 wrote
        if (nInside_[eye]) {
            em.tileList = tileListDev_ + listOffInside_[eye];
            em.ringStrips = 1; // of a ring tile, RCAS only reads the pixels next to an inside tile
            e = launch_easu(cfg_.precision, (int)in.format, (int)mid.format, em, ps.cnt, stream, nInside_[eye] + nRing_[eye]);
        }
        if (e == hipSuccess && nOutside_[eye]) {
            eo.tileList = tileListDev_ + listOffOutside_[eye];
            eo.tileRec = tileRecDev_ + 4 * listOffOutside_[eye];
            e = launch_easu_outside((int)in.format, (int)mid.format, (int)out.format, eo, nOutside_[eye], ps.cnt, aux);
        }
        if (e == hipSuccess && nInside_[eye]) {
            rb.tileList = tileListDev_ + listOffInside_[eye];
            if (spanRecDev_ && nSpans_[eye]) { rb.spanRec = spanRecDev_ + 2 * spanOff_[eye]; rb.nSpans = nSpans_[eye]; }
            e = launch_rcas(cfg_.precision, (int)mid.format, (int)out.format, rb, ps.cnt, stream, nInside_[eye]);
        }
    }
    Join(stream, aux);
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("mask-sorted EASU+RCAS launch: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

int PostProcessor::ApplyFused(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                              const ovrfsr_image &out, size_t outStride, hipStream_t stream)
{
    FusedArgs a;
    a.v = make_view(in, inStride, out, outStride);
    std::memcpy(&a.sx, &easuCon_[0], 4); std::memcpy(&a.sy, &easuCon_[1], 4);
    std::memcpy(&a.cx, &easuCon_[2], 4); std::memcpy(&a.cy, &easuCon_[3], 4);
    std::memcpy(&a.sharp, &rcasCon_[0], 4);
    a.debug = rcasCon_[3];
    FillMask(a.m, firstEye, alternate);
    a.cellsW = fusedCellsW_; a.cellsH = fusedCellsH_;
    a.tilesX = (out.width + kTileW - 1) / kTileW;
    a.tilesY = (out.height + kTileH - 1) / kTileH;
    a.tileList = nullptr;
    a.tieHalfMin = TieHalfMin();
    a.bilX = bilinDev_; a.bilY = bilinDev_ + bilYOff_;
    hipError_t e = hipSuccess;
    if (!tileListDev_) {
        e = launch_fused(cfg_.precision, (int)in.format, (int)IntermediateFormat(), (int)out.format, a, n, stream);
    } else {
        // masked: tiles entirely outside the radius never need an intermediate (RCAS there is a tinted copy), they are
        // written in final form by the LDS-free bilinear kernel; only tiles touching the radius run the fused kernel
        EasuArgs ea;
        FillEasu(ea, in, inStride, out, outStride, firstEye, alternate);
        EyePass passes[2];
        const int np = EyePasses(n, firstEye, alternate, inStride, outStride, passes);
        hipStream_t aux = Fork(stream, OverlapOutside(in));
        for (int p = 0; p < np && e == hipSuccess; ++p) {
            const EyePass &ps = passes[p];
            FusedArgs fb = a;
            EasuArgs eb = ea;
            fb.v.in = at_offset(fb.v.in, ps.inOff); fb.v.out = at_offset(fb.v.out, ps.outOff); fb.v.in_stride = ps.inStride; fb.v.out_stride = ps.outStride;
            eb.v = fb.v;
            if (ps.split) { fb.m.first_eye = eb.m.first_eye = (uint32_t)ps.eye; fb.m.alternate = eb.m.alternate = 0; }
            if (nInside_[ps.eye]) {
                fb.tileList = tileListDev_ + listOffInside_[ps.eye];
                e = launch_fused(cfg_.precision, (int)in.format, (int)IntermediateFormat(), (int)out.format, fb, ps.cnt, stream, nInside_[ps.eye]);
            }
            if (e == hipSuccess && nOutside_[ps.eye]) {
                eb.tileList = tileListDev_ + listOffOutside_[ps.eye];
                eb.tileRec = tileRecDev_ + 4 * listOffOutside_[ps.eye];
                e = launch_easu_outside((int)in.format, (int)IntermediateFormat(), (int)out.format, eb, nOutside_[ps.eye], ps.cnt, aux);
            }
        }
        Join(stream, aux);
    }
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("fused EASU+RCAS launch: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

int PostProcessor::ApplySharpening(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                                   const ovrfsr_image &out, size_t outStride, hipStream_t stream)
{
    if (cfg_.use_nis) {
        NisArgs na;
        na.v = make_view(in, inStride, out, outStride);
        FillNis(na, firstEye, alternate);
        na.tilesX = (out.width + 31) / 32;   // Dispatch(ceil(outW/32), ceil(outH/32)), :492
        na.tilesY = (out.height + 31) / 32;
        na.tileList = nullptr;
        hipError_t e = launch_nis_sharpen(cfg_.precision, (int)in.format, (int)out.format, na, n, stream);
        if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("NVSharpen launch: ") + hipGetErrorString(e));
        return OVRFSR_OK;
    }
    RcasArgs a;
    a.v = make_view(in, inStride, out, outStride);
    std::memcpy(&a.sharp, &rcasCon_[0], 4);
    a.debug = rcasCon_[3];
    FillMask(a.m, firstEye, alternate);
    a.tilesX = (out.width + kTileW - 1) / kTileW;
    a.tilesY = (out.height + kTileH - 1) / kTileH;
    a.tileList = nullptr;
    a.spanRec = nullptr; a.nSpans = 0;
    hipError_t e = launch_rcas(cfg_.precision, (int)in.format, (int)out.format, a, n, stream);
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("RCAS launch: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

// `in`/`out` describe image 0 of a batch of n; `out` is the FINAL destination.
int PostProcessor::ApplyPostProcess(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                                    const ovrfsr_image &out, size_t outStride, hipStream_t stream)
{
    if (out.format == OVRFSR_FORMAT_BGRA8_UNORM) return Fail(OVRFSR_ERR_UNSUPPORTED, "BGRA8 is an input-only format");
    if (in.format == OVRFSR_FORMAT_BGRA8_UNORM) {
        // the reference reads B8G8R8A8 submissions through a typed view and writes R8G8B8A8 (PostProcessor.cpp:30-61,63-74):
        // re-order the channels once, then run the RGBA8 pipeline on the copy
        const size_t tight = (size_t)in.width * in.height * 4;
        int rcs = EnsureBuffer(&swizzled_, &swizzledBytes_, tight * n);
        if (rcs != OVRFSR_OK) return rcs;
        hipError_t e = launch_bgra_to_rgba(static_cast<const uint8_t *>(in.data), in.pitch_bytes, inStride, static_cast<uint8_t *>(swizzled_),
                                           in.width, in.height, n, stream);
        if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("BGRA8 re-order launch: ") + hipGetErrorString(e));
        ovrfsr_image rgba = in;
        rgba.data = swizzled_; rgba.format = OVRFSR_FORMAT_RGBA8_UNORM; rgba.pitch_bytes = in.width * 4;
        return ApplyPostProcess(n, firstEye, alternate, rgba, tight, out, outStride, stream);
    }
    // R10G10B10A2 exists for the reference's 10-bit path: 10-bit in -> 10-bit out (or float, to measure parity)
    const bool inTen = in.format == OVRFSR_FORMAT_RGB10A2_UNORM, outTen = out.format == OVRFSR_FORMAT_RGB10A2_UNORM;
    if ((outTen && !inTen) || (inTen && !outTen && out.format != OVRFSR_FORMAT_RGBA32F))
        return Fail(OVRFSR_ERR_UNSUPPORTED, "RGB10A2 images pair with an RGB10A2 (or RGBA32F) destination only");
    // (a float intermediate in front of a 10-bit destination is a kernel pair nobody builds: refused here, before anything is launched,
    // instead of surfacing as a launch error behind the EASU pass -- found by the round-6 format sweep)
    if (outTen && doUpscale_ && doSharpen_ && IntermediateFormat() != OVRFSR_FORMAT_RGB10A2_UNORM)
        return Fail(OVRFSR_ERR_UNSUPPORTED, "RGB10A2 pipelines keep a 10-bit intermediate (quantize_intermediate = 1)");
    // (the ring is complete: slots are created in order; a call that is being captured into a graph is not timed: its events could never be read back)
    const bool timing = cfg_.debug_mode && !capturing_ && queries_[kQueryCount - 1].end;
    if (timing) (void)hipEventRecord(queries_[currentQuery_].start, stream);
    int rc = OVRFSR_OK;
    if (useSorted_) {
        rc = ApplySorted(n, firstEye, alternate, in, inStride, out, outStride, stream);
    } else if (useFused_) {
        rc = ApplyFused(n, firstEye, alternate, in, inStride, out, outStride, stream);
    } else if (doUpscale_ && doSharpen_) {
        ovrfsr_image mid;
        mid.width = outputWidth_; mid.height = outputHeight_;
        mid.format = IntermediateFormat();
        mid.pitch_bytes = outputWidth_ * texel_bytes(mid.format);
        const size_t midStride = (size_t)mid.pitch_bytes * outputHeight_;
        rc = EnsureBuffer(&upscaled_, &upscaledBytes_, midStride * n);
        if (rc != OVRFSR_OK) return rc;
        mid.data = upscaled_;
        rc = ApplyUpscaling(n, firstEye, alternate, in, inStride, mid, midStride, stream);
        if (rc == OVRFSR_OK) rc = ApplySharpening(n, firstEye, alternate, mid, midStride, out, outStride, stream);
    } else if (doUpscale_) {
        rc = ApplyUpscaling(n, firstEye, alternate, in, inStride, out, outStride, stream);
    } else if (doSharpen_) {
        rc = ApplySharpening(n, firstEye, alternate, in, inStride, out, outStride, stream);
    }
    if (timing) {
        (void)hipEventRecord(queries_[currentQuery_].end, stream);
        queries_[currentQuery_].pending = true;
        queries_[currentQuery_].images = n;
        lastQuery_ = currentQuery_;
        CollectQuery(stream);
    }
    return rc;
}

// the kernels read neighbours of what other workgroups write: the byte ranges of input and output images must be disjoint
bool PostProcessor::RangesOverlap(const ovrfsr_image &in0, size_t inStride, const ovrfsr_image &out0, size_t outStride, uint32_t n)
{
    const uintptr_t i0 = reinterpret_cast<uintptr_t>(in0.data), o0 = reinterpret_cast<uintptr_t>(out0.data);
    const uintptr_t i1 = i0 + (n - 1) * inStride + (size_t)in0.pitch_bytes * in0.height;
    const uintptr_t o1 = o0 + (n - 1) * outStride + (size_t)out0.pitch_bytes * out0.height;
    return i0 < o1 && o0 < i1;
}

int PostProcessor::Apply(int eye, const ovrfsr_image *in, const ovrfsr_bounds *bounds, ovrfsr_image *out, hipStream_t stream)
{
    if (!enabled_) return Fail(OVRFSR_ERR_DISABLED, "post-processing disabled after an earlier failure; call reset");
    if (!out) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "out is null");
    int rc = CheckImage(in, "in");
    if (rc != OVRFSR_OK) return rc;
    if (eye != OVRFSR_EYE_LEFT && eye != OVRFSR_EYE_RIGHT) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "bad eye");
    static const ovrfsr_bounds defaultBounds = {0, 0, 1, 1};
    if (!bounds) bounds = &defaultBounds;

    if (!cfg_.fsr_enabled) { // PostProcessor.cpp:135: texture is forwarded untouched
        *out = *in;
        return OVRFSR_OK;
    }
    DeviceGuard guard(device_);
    if (guard.err != hipSuccess) return Fail(OVRFSR_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));

    capturing_ = stream_capturing(stream);
    if (capturing_ && (!initialized_ || in->width != inputWidth_ || in->height != inputHeight_ || in->format != inputFormat_))
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, kCaptureRefusal); // (nothing touched: the capture stays valid, the ctx stays enabled)
    if (initialized_ && (in->width != inputWidth_ || in->height != inputHeight_ || in->format != inputFormat_)) {
        bool keep = false;
        if (havePending_) { // the recorded eye belongs to the old resources
            const uint8_t *po = static_cast<const uint8_t *>(pendingOut_.data), *sb = static_cast<const uint8_t *>(sharpened_);
            const bool owned = sb && po >= sb && po < sb + sharpenedBytes_;
            rc = FlushPending(stream);
            if (rc != OVRFSR_OK) return rc;
            if (owned) { // its result was written to a ctx-owned image the caller already holds: that image outlives the rebuild
                if (retired_) (void)hipFree(retired_);
                retired_ = sharpened_; sharpened_ = nullptr; sharpenedBytes_ = 0;
                keep = true;
            }
        }
        ResetKeeping(keep); // "Texture size changed, recreating resources" (:139-142)
    }
    if (!initialized_) {
        textureContainsOnlyOneEye_ = std::fabs(bounds->uMax - bounds->uMin) > .5f; // :146
        rc = PrepareResources(*in);
        if (rc != OVRFSR_OK) { enabled_ = false; return rc; } // :148-151
    }

    // caller-owned or ctx-owned final image
    ovrfsr_image dst;
    const bool stages = doUpscale_ || doSharpen_;
    // cfg.pair_submit (header): one texture per eye and something to launch -> the first eye of a frame is recorded, the other eye launches both
    const bool pairMode = cfg_.pair_submit != 0 && textureContainsOnlyOneEye_ && stages;
    if (out->data) {
        rc = CheckImage(out, "out");
        if (rc != OVRFSR_OK) return rc;
        if (out->width != outputWidth_ || out->height != outputHeight_) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "out has the wrong size");
        dst = *out;
    } else {
        dst.width = outputWidth_; dst.height = outputHeight_;
        dst.format = in->format == OVRFSR_FORMAT_BGRA8_UNORM ? OVRFSR_FORMAT_RGBA8_UNORM : in->format; // DetermineOutputFormat (:63-74)
        dst.pitch_bytes = dst.width * texel_bytes(dst.format);
        // (pair mode: both eyes' results are alive at once -- two ctx-owned images, left first)
        const size_t one = (size_t)dst.pitch_bytes * dst.height;
        if (pairMode && havePending_ && sharpenedBytes_ < 2 * one) { rc = FlushPending(stream); if (rc != OVRFSR_OK) return rc; } // (cannot grow under a recorded LEFT)
        rc = EnsureBuffer(&sharpened_, &sharpenedBytes_, pairMode ? 2 * one : one);
        if (rc != OVRFSR_OK) return rc;
        dst.data = pairMode && eye == OVRFSR_EYE_RIGHT ? static_cast<uint8_t *>(sharpened_) + one : sharpened_;
    }
    // an in-place call would race: RCAS / NVSharpen / EASU read neighbour texels other workgroups overwrite.  Checked against the
    // RESOLVED destination -- the caller's buffer or the ctx-owned one: chaining the previous ctx-owned result back in as `in`
    // (sharpen-only mode, where the sizes agree) is the same race
    if (stages && RangesOverlap(*in, 0, dst, 0, 1)) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "input and output images overlap");

    lastApplyRecorded_ = false;
    if (pairMode) {
        // Pairing is by ARRIVAL order (round 6: games submit L,R or R,L; until then only LEFT was recorded and an R,L game had every LEFT
        // batched with the NEXT frame's RIGHT).  A small state machine keeps a disturbed sequence from turning into a standing one-frame lag:
        //   pending, other eye arrives   -> both as one batch of two; that pair's first eye is remembered as the frame's first eye
        //   pending, SAME eye again      -> the older one alone, this one alone, and no more recording until the other eye shows up
        //                                   (a host that submits one eye per frame, or a frame that lost an eye)
        //   nothing pending              -> recorded if it is (or may be) a frame's first eye; a frame's SECOND eye with nothing pending
        //                                   (its partner was flushed by a batch call or a size change) is processed at once
        const int prevEye = lastEye_;
        lastEye_ = eye;
        if (havePending_ && pendingEye_ == eye) {
            rc = FlushPending(stream);
            if (rc != OVRFSR_OK) return rc;
            pairDefer_ = false; // falls through to the single launch below
        } else if (havePending_) {
            // image 0 = the recorded eye, image 1 = this one.  Image 1 lives at base + stride with the strides taken modulo 2^64 (the kernels
            // add `i * stride` to a 64-bit base, i in {0, 1}): it may lie above or below image 0, inputs and outputs independently.  Images
            // that differ in size / pitch / format, one texture submitted for both eyes, or outputs that overlap an input or each other
            // take two single launches instead.
            const ovrfsr_image &fi = pendingIn_, &fo = pendingOut_, &si = *in, &so = dst;
            const bool same = fi.width == si.width && fi.height == si.height && fi.pitch_bytes == si.pitch_bytes && fi.format == si.format &&
                              fo.width == so.width && fo.height == so.height && fo.pitch_bytes == so.pitch_bytes && fo.format == so.format;
            const size_t inStride = (size_t)(reinterpret_cast<uintptr_t>(si.data) - reinterpret_cast<uintptr_t>(fi.data));
            const size_t outStride = (size_t)(reinterpret_cast<uintptr_t>(so.data) - reinterpret_cast<uintptr_t>(fo.data));
            const bool disjoint = fi.data != si.data && !RangesOverlap(fo, 0, so, 0, 1) && !RangesOverlap(fi, 0, so, 0, 1) && !RangesOverlap(si, 0, fo, 0, 1) &&
                                  !RangesOverlap(fi, 0, fo, 0, 1) && !RangesOverlap(si, 0, so, 0, 1);
            pairFirstEye_ = pendingEye_;
            pairDefer_ = true;
            if (same && disjoint && inStride % texel_bytes(fi.format) == 0 && outStride % texel_bytes(fo.format) == 0) {
                havePending_ = false;
                rc = ApplyPostProcess(2, pendingEye_, 1, fi, inStride, fo, outStride, stream);
                if (rc != OVRFSR_OK) return rc;
                lastSubmittedTexture_ = in->data;
                eyeCount_ = (eyeCount_ + 1) % 2;
                outputTexture_ = dst;
                *out = dst;
                return OVRFSR_OK;
            }
            rc = FlushPending(stream); // then this eye alone, below
            if (rc != OVRFSR_OK) return rc;
        } else if (!pairDefer_) {
            if (prevEye >= 0 && prevEye != eye) pairDefer_ = true; // both eyes are back: pairs again from the next call on
        } else if (pairFirstEye_ < 0 || pairFirstEye_ == eye) {
            // the first eye of a frame: record it and hand out where its result will be
            pendingIn_ = *in; pendingOut_ = dst; pendingEye_ = eye; havePending_ = true;
            lastApplyRecorded_ = true;
            lastSubmittedTexture_ = in->data;
            eyeCount_ = (eyeCount_ + 1) % 2;
            outputTexture_ = dst;
            *out = dst;
            return OVRFSR_OK;
        }
    }
    // a shared side-by-side texture is processed once, on the first Submit (:155-158)
    if (eyeCount_ == 0 || textureContainsOnlyOneEye_ || in->data != lastSubmittedTexture_) {
        if (stages) {
            rc = ApplyPostProcess(1, textureContainsOnlyOneEye_ ? eye : OVRFSR_EYE_LEFT, 0, *in, 0, dst, 0, stream);
            if (rc != OVRFSR_OK) return rc;
            outputTexture_ = dst;
        } else {
            outputTexture_ = *in;
        }
    }
    lastSubmittedTexture_ = in->data;
    eyeCount_ = (eyeCount_ + 1) % 2;
    *out = outputTexture_;
    return OVRFSR_OK;
}

int PostProcessor::ApplyBatch(uint32_t n, int firstEye, int alternate, const ovrfsr_image *in0, size_t inStride,
                              const ovrfsr_image *out0, size_t outStride, hipStream_t stream, bool sharedTextures)
{
    if (!enabled_) return Fail(OVRFSR_ERR_DISABLED, "post-processing disabled after an earlier failure; call reset");
    if (n == 0) return OVRFSR_OK;
    if (n > 65535) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "batch too large for one launch (65535)");
    int rc = CheckImage(in0, "in0");
    if (rc != OVRFSR_OK) return rc;
    rc = CheckImage(out0, "out0");
    if (rc != OVRFSR_OK) return rc;
    if (!cfg_.fsr_enabled) return Fail(OVRFSR_ERR_DISABLED, "fsr_enabled is 0: nothing to launch");
    if (n > 1 && (inStride < (size_t)in0->pitch_bytes * in0->height || outStride < (size_t)out0->pitch_bytes * out0->height))
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "batch stride smaller than one image");
    if (n > 1 && (inStride % texel_bytes(in0->format) != 0 || outStride % texel_bytes(out0->format) != 0))
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "batch stride is not a multiple of the texel size (images i > 0 would be misaligned)");
    if (RangesOverlap(*in0, inStride, *out0, outStride, n)) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "input and output batches overlap");
    DeviceGuard guard(device_);
    if (guard.err != hipSuccess) return Fail(OVRFSR_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    capturing_ = stream_capturing(stream);
    if (capturing_ && (!initialized_ || in0->width != inputWidth_ || in0->height != inputHeight_ || in0->format != inputFormat_ || textureContainsOnlyOneEye_ == sharedTextures))
        return Fail(OVRFSR_ERR_INVALID_ARGUMENT, kCaptureRefusal);
    if (havePending_) { rc = FlushPending(stream); if (rc != OVRFSR_OK) return rc; } // cfg.pair_submit: a recorded LEFT goes first
    // shared side-by-side textures: both mask centres per image, processed once each (PostProcessor.cpp:146,155-158,298-301)
    if (initialized_ && (in0->width != inputWidth_ || in0->height != inputHeight_ || in0->format != inputFormat_ || textureContainsOnlyOneEye_ == sharedTextures))
        Reset();
    if (!initialized_) {
        textureContainsOnlyOneEye_ = !sharedTextures;
        rc = PrepareResources(*in0);
        if (rc != OVRFSR_OK) { enabled_ = false; return rc; }
    }
    if (out0->width != outputWidth_ || out0->height != outputHeight_) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "out has the wrong size");
    if (!(doUpscale_ || doSharpen_)) return Fail(OVRFSR_ERR_UNSUPPORTED, "no stage selected (render_scale == 1 with NIS off would still sharpen)");
    return ApplyPostProcess(n, firstEye, alternate, *in0, inStride, *out0, outStride, stream);
}

// cfg.pair_submit: the recorded submission on its own (its other eye did not follow)
int PostProcessor::FlushPending(hipStream_t stream)
{
    if (!havePending_) return OVRFSR_OK;
    havePending_ = false;
    return ApplyPostProcess(1, pendingEye_, 0, pendingIn_, 0, pendingOut_, 0, stream);
}

// PostProcessor.cpp:608-626: advance the ring and read the slot recorded kQueryCount-1 applies ago
void PostProcessor::CollectQuery(hipStream_t stream)
{
    currentQuery_ = (currentQuery_ + 1) % kQueryCount;
    ProfileQuery &q = queries_[currentQuery_];
    if (!q.pending) return;
    // no host waits while a graph is captured.  (The legacy NULL stream cannot be captured, and querying it while ANOTHER
    // stream is in global-mode capture would invalidate that capture: skipped.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr && (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return;
    float ms = 0.0f;
    if (hipEventSynchronize(q.end) != hipSuccess || hipEventElapsedTime(&ms, q.start, q.end) != hipSuccess) return; // "disjoint": reading dropped
    q.pending = false;
    // the reference times one Apply = one eye image (x2 below when each eye has its own texture: a frame); a batched apply
    // covered q.images of them, so its reading counts per image
    summedGpuTime_ += ms * 1e-3f / (float)(q.images ? q.images : 1u);
    if (++countedQueries_ >= 500) {
        float avgTimeMs = 1000.f / countedQueries_ * summedGpuTime_;
        if (textureContainsOnlyOneEye_) avgTimeMs *= 2;
        avgGpuTimeMs_ = avgTimeMs;
        ++avgReports_;
        static const bool log = [] { const char *e = std::getenv("OVRFSR_LOG"); return e && e[0] == '1'; }();
        if (log) std::fprintf(stderr, "Average GPU processing time for upscale: %g ms\n", avgTimeMs);
        countedQueries_ = 0;
        summedGpuTime_ = 0.f;
    }
}

int PostProcessor::LastGpuTimeMs(float *ms)
{
    if (!ms) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "ms is null");
    if (lastQuery_ < 0 || !queries_[lastQuery_].start) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "no timed apply (debug_mode off?)");
    hipError_t e = hipEventSynchronize(queries_[lastQuery_].end);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, queries_[lastQuery_].start, queries_[lastQuery_].end);
    if (e != hipSuccess) return Fail(OVRFSR_ERR_HIP, std::string("event timing: ") + hipGetErrorString(e));
    return OVRFSR_OK;
}

int PostProcessor::AverageGpuTimeMs(float *ms, uint32_t *reports)
{
    if (!ms || !reports) return Fail(OVRFSR_ERR_INVALID_ARGUMENT, "null argument");
    *ms = avgGpuTimeMs_;
    *reports = avgReports_;
    return OVRFSR_OK;
}

} // namespace ovrfsr
