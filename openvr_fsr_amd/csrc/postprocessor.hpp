// postprocessor.hpp -- HIP stream/launch manager that replaces the reference's D3D11 host pipeline
// (src/postprocess/PostProcessor.{h,cpp}).  Same public shape -- Apply(eye, texture, bounds) and
// Reset() -- same lazy (re)build rules, same stage selection; D3D11 resources become linear device
// buffers and `context->Dispatch` becomes a kernel launch on a caller-provided HIP stream.
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>
#include <vector>
#include "../../include/openvr_fsr_amd.h"
#include "fsr_params.h"
#include "nis_tables.h"

namespace ovrfsr {

// host-side constant math (restates FsrEasuCon / FsrRcasCon / NVScalerUpdateConfig; see constants.cpp)
void easu_con(uint32_t con[16], float inVpW, float inVpH, float inW, float inH, float outW, float outH);
void rcas_con(uint32_t con[4], float stops);
void mask_constants(uint32_t centre[4], uint32_t radius[4], uint32_t outW, uint32_t outH, const float proj[4],
                    float cfgRadius, int onlyOneEye, int eye);
uint32_t classify_mask(const uint32_t centre[4], uint32_t r2, uint32_t outW, uint32_t outH, uint32_t gw, uint32_t gh);
#ifdef OVRFSR_BOUNDS
void debug_fail_resource(int nth); // checked builds: the nth device allocation / stream / event creation from now on fails, once (postprocessor.cpp)
#endif

class PostProcessor {
public:
    PostProcessor(int device, const ovrfsr_config &cfg);
    ~PostProcessor();

    // PostProcessor::Apply, PostProcessor.cpp:123-164
    int Apply(int eye, const ovrfsr_image *in, const ovrfsr_bounds *bounds, ovrfsr_image *out, hipStream_t stream);
    int ApplyBatch(uint32_t n, int firstEye, int alternate, const ovrfsr_image *in0, size_t inStride,
                   const ovrfsr_image *out0, size_t outStride, hipStream_t stream, bool sharedTextures = false);
    // PostProcessor::Reset, PostProcessor.cpp:166-194
    void Reset();
    int SetConfig(const ovrfsr_config &cfg);
    const ovrfsr_config &GetConfig() const { return cfg_; }
    const char *LastError() const { return lastError_.c_str(); }
    bool PairPending() const { return lastApplyRecorded_ && havePending_; }
    int LastGpuTimeMs(float *ms);
    int AverageGpuTimeMs(float *ms, uint32_t *reports);

private:
    int device_;
    ovrfsr_config cfg_;
    std::string lastError_;

    // PostProcessor.h:16-24
    bool enabled_ = true;
    bool initialized_ = false;
    uint32_t inputWidth_ = 0, inputHeight_ = 0, outputWidth_ = 0, outputHeight_ = 0;
    uint32_t inputFormat_ = 0;
    bool textureContainsOnlyOneEye_ = true;
    // PostProcessor.h:66-68
    const void *lastSubmittedTexture_ = nullptr;
    ovrfsr_image outputTexture_ = {};
    int eyeCount_ = 0;

    // stage selection, PostProcessor.cpp:530-535 / :586-594
    bool doUpscale_ = false, doSharpen_ = false;

    // "constant buffers", one per eye: PostProcessor.cpp:296-338, :419-460
    uint32_t easuCon_[16] = {};
    uint32_t rcasCon_[4] = {};
    uint32_t centre_[2][4] = {};
    uint32_t radius_[4] = {};
    uint32_t maskMode_[2] = {};
    uint32_t outsideCols_ = 0, outsideRows_[2] = {0, 0}; // bilinear footprint bound of a 32-wide tile; rows for 32- and 24-row tiles
    float rcpOut_[2] = {0, 0}; // RN(1/outW), RN(1/outH) and whether mul+2fma reproduces o/out for every o (div_exact)
    bool rcpExact_ = false;
    int cellsW_ = 0, cellsH_ = 0;
    int fusedCellsW_ = 0, fusedCellsH_ = 0; // footprint of the 34x34 EASU block of the fused kernel
    bool useFused_ = false;
    // NIS: the 256-byte NISConfig (PostProcessor.cpp:307-310) and the coefficient "textures" (:366-381)
    NisConstants nisConfig_ = {};
    float *nisCoefDev_ = nullptr; // coef_scale[512] | coef_usm[512]
    BilinTap *bilinDev_ = nullptr; // [outW] column taps (padded to a multiple of the tile width with copies of the last one), then [outH] row taps at bilYOff_
    uint32_t bilYOff_ = 0;
    // mask-sorted EASU tile lists (product build, masked configs): per eye, tiles with any group inside the radius
    // and tiles entirely outside; the latter run through an LDS-free kernel at twice the occupancy
    uint32_t *tileListDev_ = nullptr;
    uint32_t *tileRecDev_ = nullptr;      // records of the same entries (inside the tileListDev_ allocation), 4 dwords each
    uint32_t *spanRecDev_ = nullptr;      // RCAS segments of the inside runs (inside the tileListDev_ allocation), 2 dwords each
    uint32_t nSpans_[2] = {0, 0};
    size_t spanOff_[2] = {0, 0};          // first segment of each eye (in segments)
    std::vector<BilinTap> bilinHost_;     // host copy of the column / row tap tables (bilinDev_)
    uint32_t nInside_[2] = {0, 0}, nOutside_[2] = {0, 0}, nRing_[2] = {0, 0};
    size_t listOffInside_[2] = {0, 0}, listOffOutside_[2] = {0, 0}, listOffRing_[2] = {0, 0}; // ring: outside tiles 4-adjacent to an inside tile
    bool useSorted_ = false;   // masked EASU+RCAS: two passes on the inside list, final-form outside tiles (ApplySorted)
    bool listsShared_ = false; // both eyes have identical lists
    // the memory-bound outside-tile kernel and the VALU-bound inside-tile kernel are independent: the former runs on
    // a ctx-owned auxiliary stream, forked from and joined back into the caller's stream with events
    hipStream_t auxStream_ = nullptr;
    hipEvent_t evFork_ = nullptr, evJoin_ = nullptr;
    // cfg.pair_submit: the recorded FIRST submission of the current frame (either eye; see the header) and what it takes to launch it
    bool havePending_ = false;
    int pendingEye_ = 0;
    int pairFirstEye_ = -1;          // the eye that opens a frame, learned from the last completed pair (-1: not known yet)
    bool pairDefer_ = true;          // false after the same eye came twice in a row, until the other eye is seen again
    int lastEye_ = -1;
    bool lastApplyRecorded_ = false; // the last Apply only recorded its submission (ovrfsr_pair_pending)
    bool capturing_ = false;         // the stream of the call in progress is being captured into a HIP graph: launches only, no (re)build
    ovrfsr_image pendingIn_{}, pendingOut_{};
    void *retired_ = nullptr; // a ctx-owned output image a flushed pair_submit eye was handed in, kept across the rebuild of a size change
    void ResetKeeping(bool keepRetired);
    int FlushPending(hipStream_t stream);
    bool OverlapOutside(const ovrfsr_image &in) const; // does a masked pass run its outside-tile kernel on the auxiliary stream?
    hipStream_t Fork(hipStream_t user, bool overlap);
    void Join(hipStream_t user, hipStream_t aux);
    int nisCellsW_ = 0, nisCellsH_ = 0;

    // ctx-owned device buffers: upscaledTexture / sharpenedTexture, PostProcessor.h:43-45,58-59
    void *swizzled_ = nullptr;          // RGBA8 copy of a BGRA8 submission (tight pitch), see ApplyPostProcess
    size_t swizzledBytes_ = 0;
    void *upscaled_ = nullptr;
    size_t upscaledBytes_ = 0;
    void *sharpened_ = nullptr;
    size_t sharpenedBytes_ = 0;

    // debug-mode GPU timing, PostProcessor.h:72-82: a ring of kQueryCount timestamp pairs; after every apply the OLDEST
    // slot is read back (so the wait is normally over already), durations are summed and every 500 readings the mean
    // is published -- doubled when each eye has its own texture, i.e. "per frame" (PostProcessor.cpp:605-626)
    static constexpr int kQueryCount = 6;
    struct ProfileQuery { hipEvent_t start = nullptr, end = nullptr; bool pending = false; uint32_t images = 1; };
    ProfileQuery queries_[kQueryCount];
    int currentQuery_ = 0, lastQuery_ = -1;
    float summedGpuTime_ = 0.0f;   // seconds
    int countedQueries_ = 0;
    float avgGpuTimeMs_ = 0.0f;
    uint32_t avgReports_ = 0;
    void CollectQuery(hipStream_t stream);

    int Fail(int status, const std::string &what);
    int CheckImage(const ovrfsr_image *img, const char *name);
    int PrepareResources(const ovrfsr_image &in);                         // :498-561
    void PrepareUpscalingResources();                                    // :285-383
    void PrepareSharpeningResources();                                   // :409-481
    int PrepareTileLists(uint32_t tileW, uint32_t tileH, uint32_t groupW, uint32_t groupH);
    struct EyePass { int eye; uint32_t cnt; size_t inOff, outOff, inStride, outStride; bool split; };
    int EyePasses(uint32_t n, int firstEye, int alternate, size_t inStride, size_t outStride, EyePass out[2]) const;
    void FillEasu(EasuArgs &a, const ovrfsr_image &in, size_t inStride, const ovrfsr_image &out, size_t outStride, int firstEye, int alternate) const;
    int PrepareNisResources();                                           // :307-310, :366-382, :432-435
    void FillNis(NisArgs &a, int firstEye, int alternate) const;
    int EnsureBuffer(void **buf, size_t *have, size_t need);
    static bool RangesOverlap(const ovrfsr_image &in0, size_t inStride, const ovrfsr_image &out0, size_t outStride, uint32_t n);
    uint32_t IntermediateFormat() const;
    float TieHalfMin() const;
    int ApplyPostProcess(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                         const ovrfsr_image &out, size_t outStride, hipStream_t stream); // :563-638
    int ApplyUpscaling(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                       const ovrfsr_image &out, size_t outStride, hipStream_t stream);   // :385-401
    int ApplySorted(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                    const ovrfsr_image &out, size_t outStride, hipStream_t stream);
    int ApplyFused(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                   const ovrfsr_image &out, size_t outStride, hipStream_t stream);
    int ApplySharpening(uint32_t n, int firstEye, int alternate, const ovrfsr_image &in, size_t inStride,
                        const ovrfsr_image &out, size_t outStride, hipStream_t stream);  // :483-496
    void FillMask(MaskArgs &m, int firstEye, int alternate) const;
};

} // namespace ovrfsr
