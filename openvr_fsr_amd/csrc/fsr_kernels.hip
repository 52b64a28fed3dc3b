// fsr_kernels.hip -- instantiates the gfx950 FSR1 kernels twice (product build and strict
// validation build, see fsr_kernels.inc) and exposes typed launchers to the host launch manager.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "fsr_params.h"
#include "fsr_launch.h"
#include "fsr_bounds.h"

#ifdef OVRFSR_TIE_AUDIT
// AUDIT BUILD (-DOVRFSR_TIE_AUDIT, never shipped): every pixel the product EASU resolves is resolved a second time in the reference's
// operator order, and what the product build stores is compared with what the strict build would store.  Counters, per device:
//   [0] pixels audited   [1] pixels the near-tie guard listed (re-resolved in reference order: equal by construction)
//   [2] FLIPS: pixels NOT listed whose stored UNORM8 bytes / guarded halves differ from the strict build's  -- the guard's claim is [2] == 0
//   [3] unlisted pixels whose half store differs in a channel BELOW xmin (outside the guard's contract: a flipped half-ulp there stays under 1e-3)
//   [4] max |product - strict| over all audited channels, fp32 bit pattern: bytes (UNORM8 stores) ...  [5] ... or half spacings (half stores)
__device__ unsigned long long g_ovrfsr_tie_audit[6];
#endif

namespace ovrfsr_fast {
#define OVRFSR_STRICT 0
#pragma clang fp contract(fast)
#include "fsr_device.inc"
#include "fsr_kernels.inc"
#undef OVRFSR_STRICT
} // namespace ovrfsr_fast

namespace ovrfsr_strict {
#define OVRFSR_STRICT 1
#pragma clang fp contract(off)
#include "fsr_device.inc"
#include "fsr_kernels.inc"
#undef OVRFSR_STRICT
} // namespace ovrfsr_strict
#pragma clang fp contract(on)

namespace ovrfsr {

// LDS row pitch (cells) of the product-build EASU kernel; 0 = footprint too wide, use the generic kernel
int easu_fast_pitch(int cellsW) { return cellsW <= 32 ? 32 : cellsW <= 40 ? 40 : 0; }
// the EASU-only kernel also has a 28-cell pitch: exactly the footprint of a 32-pixel tile at scale 3/4
static int easu_kernel_pitch(int cellsW) { return cellsW <= 28 ? 28 : easu_fast_pitch(cellsW); }

size_t easu_lds_bytes(int prec, int in_fmt, int cellsW, int cellsH)
{
    if (prec != PREC_FP32_STRICT && easu_fast_pitch(cellsW) != 0)
        return (size_t)easu_fast_pitch(cellsW) * cellsH * (16 + 16 + 4) + (size_t)easu_fast_pitch(cellsW) * kLumPadRows * 4; // pitch 32/40 (the fused kernel's; 28 fits inside) + kLumPadRows
    const bool wide = (prec == PREC_FP32_STRICT) || (in_fmt == FMT_RGBA32F) || (in_fmt == FMT_RGB10A2);
    const size_t ncell = (size_t)cellsW * cellsH;
    const size_t col = (ncell * (wide ? 16 : 8) + 15) & ~(size_t)15;
    return col + ncell * 16 + ncell * 4;
}

template <int I, int O, bool M>
static void easu_fast_go(int pitch, const EasuArgs &a, dim3 grid, hipStream_t s)
{
    const size_t lds = (size_t)pitch * a.cellsH * 36 + (size_t)pitch * kLumPadRows * 4; // colour + analysis (float4) + luma planes + pad rows
    if (pitch == 28) hipLaunchKernelGGL((ovrfsr_fast::easu_fast_kernel<I, O, 28, M>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    else if (pitch == 32) hipLaunchKernelGGL((ovrfsr_fast::easu_fast_kernel<I, O, 32, M>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    else hipLaunchKernelGGL((ovrfsr_fast::easu_fast_kernel<I, O, 40, M>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
}

template <int I, int O>
static hipError_t easu_go(bool strict, const EasuArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const int pitch = easu_kernel_pitch(a.cellsW);
    const bool masked = a.m.mode[0] != MASK_ALL_INSIDE || a.m.mode[1] != MASK_ALL_INSIDE;
    if (strict) hipLaunchKernelGGL((ovrfsr_strict::easu_kernel<I, O>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    // footprints wider than the fixed pitches, and the 10-bit format (a quantised destination without a near-tie guard of
    // its own): the generic kernel, whose resolve is the reference-order one in every build
    else if (pitch == 0 || I == FMT_RGB10A2 || O == FMT_RGB10A2) hipLaunchKernelGGL((ovrfsr_fast::easu_kernel<I, O>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    else if (masked) easu_fast_go<I, O, true>(pitch, a, grid, s);
    else easu_fast_go<I, O, false>(pitch, a, grid, s);
    return hipGetLastError();
}
template <int I, int O>
static hipError_t rcas_go(bool strict, const RcasArgs &a, dim3 grid, hipStream_t s)
{
    // OVRFSR_RCAS_DPP=0: A/B switch back to the per-lane-loads kernel (diagnostic)
    static const bool dpp = [] { const char *e = std::getenv("OVRFSR_RCAS_DPP"); return !(e && e[0] == '0'); }();
    const bool unmasked = !a.tileList && a.m.mode[0] == MASK_ALL_INSIDE && a.m.mode[1] == MASK_ALL_INSIDE;
    if (strict) {
        hipLaunchKernelGGL((ovrfsr_strict::rcas_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
    } else if constexpr (I == FMT_RGBA8 && O != FMT_RGB10A2) {
        if (dpp && unmasked) {
            const uint32_t tx = (uint32_t)(a.v.outW + kRcasDppTileW - 1) / kRcasDppTileW, ty = (uint32_t)(a.v.outH + kRcasDppTileH - 1) / kRcasDppTileH;
            // small launches: the grid is r = workgroups / kRcasResident rounds of the resident workgroups, the last one partly filled;
            // half-height workgroups run ceil(2r) rounds of half the length (3 % more work per pixel).  One C2 eye image: r = 1.41,
            // 2 rounds against 3 half rounds = 1.5 (14.4 instead of 15.5 us, profiles/r05_frame.txt); a batch: no difference, the
            // 8-row form wins.  OVRFSR_RCAS_TH=16|32 forces one form (tuning)
            static const int forced = [] { const char *e = std::getenv("OVRFSR_RCAS_TH"); return e ? std::atoi(e) : 0; }();
            const uint64_t wgs = (uint64_t)tx * ty * grid.z;
            const uint64_t full = (wgs + kRcasResident - 1) / kRcasResident, half = (2 * wgs + kRcasResident - 1) / kRcasResident;
            const bool small = forced ? forced == 16 : (wgs < 16 * kRcasResident && 103 * half < 200 * full);
            if (small) {
                const uint32_t ty16 = (uint32_t)(a.v.outH + 15) / 16;
                hipLaunchKernelGGL((ovrfsr_fast::rcas_dpp_kernel<O, false, 16>), dim3(tx * ty16, 1, grid.z), dim3(kThreads), 0, s, a);
            } else {
                hipLaunchKernelGGL((ovrfsr_fast::rcas_dpp_kernel<O>), dim3(tx * ty, 1, grid.z), dim3(kThreads), 0, s, a);
            }
        } else if (dpp && a.tileList && a.spanRec && a.nSpans) {
            // mask-sorted form: the DPP kernel on 62-column segments of the runs of tiles touching the radius
            hipLaunchKernelGGL((ovrfsr_fast::rcas_dpp_kernel<O, true>), dim3(a.nSpans, 1, grid.z), dim3(kThreads), 0, s, a);
        } else {
            hipLaunchKernelGGL((ovrfsr_fast::rcas_direct_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
        }
    } else {
        hipLaunchKernelGGL((ovrfsr_fast::rcas_direct_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
    }
    return hipGetLastError();
}

#define OVRFSR_DISPATCH_FMT(FN, ...)                                                                     \
    switch (in_fmt < 3 && out_fmt < 3 ? in_fmt * 3 + out_fmt : -1) {                                                                      \
    case 0: return FN<FMT_RGBA8, FMT_RGBA8>(__VA_ARGS__);                                                \
    case 1: return FN<FMT_RGBA8, FMT_RGBA16F>(__VA_ARGS__);                                              \
    case 2: return FN<FMT_RGBA8, FMT_RGBA32F>(__VA_ARGS__);                                              \
    case 3: return FN<FMT_RGBA16F, FMT_RGBA8>(__VA_ARGS__);                                              \
    case 4: return FN<FMT_RGBA16F, FMT_RGBA16F>(__VA_ARGS__);                                            \
    case 5: return FN<FMT_RGBA16F, FMT_RGBA32F>(__VA_ARGS__);                                            \
    case 6: return FN<FMT_RGBA32F, FMT_RGBA8>(__VA_ARGS__);                                              \
    case 7: return FN<FMT_RGBA32F, FMT_RGBA16F>(__VA_ARGS__);                                            \
    case 8: return FN<FMT_RGBA32F, FMT_RGBA32F>(__VA_ARGS__);                                            \
    default: break;                                                                                      \
    }                                                                                                    \
    /* R10G10B10A2: only what the reference's 10-bit path needs (10-bit in -> 10-bit out, PostProcessor.cpp:63-74) plus a \
       float destination for un-quantised parity checks */                                               \
    if (in_fmt == FMT_RGB10A2 && out_fmt == FMT_RGB10A2) return FN<FMT_RGB10A2, FMT_RGB10A2>(__VA_ARGS__); \
    if (in_fmt == FMT_RGB10A2 && out_fmt == FMT_RGBA32F) return FN<FMT_RGB10A2, FMT_RGBA32F>(__VA_ARGS__); \
    return hipErrorInvalidValue;

// the same without the 10-bit pairs: kernels that are not built for R10G10B10A2 (fused, LDS-staged outside)
#define OVRFSR_DISPATCH_FMT3(FN, ...)                                                                     \
    switch (in_fmt < 3 && out_fmt < 3 ? in_fmt * 3 + out_fmt : -1) {                                                                      \
    case 0: return FN<FMT_RGBA8, FMT_RGBA8>(__VA_ARGS__);                                                \
    case 1: return FN<FMT_RGBA8, FMT_RGBA16F>(__VA_ARGS__);                                              \
    case 2: return FN<FMT_RGBA8, FMT_RGBA32F>(__VA_ARGS__);                                              \
    case 3: return FN<FMT_RGBA16F, FMT_RGBA8>(__VA_ARGS__);                                              \
    case 4: return FN<FMT_RGBA16F, FMT_RGBA16F>(__VA_ARGS__);                                            \
    case 5: return FN<FMT_RGBA16F, FMT_RGBA32F>(__VA_ARGS__);                                            \
    case 6: return FN<FMT_RGBA32F, FMT_RGBA8>(__VA_ARGS__);                                              \
    case 7: return FN<FMT_RGBA32F, FMT_RGBA16F>(__VA_ARGS__);                                            \
    case 8: return FN<FMT_RGBA32F, FMT_RGBA32F>(__VA_ARGS__);                                            \
    default: break;                                                                                      \
    }                                                                                                    \
    return hipErrorInvalidValue;

// LDS of the fused kernel: EASU planes + 34x34 intermediate (float4 cells)
size_t fused_lds_bytes(int prec, int in_fmt, int mid_fmt, int cellsW, int cellsH)
{
    size_t e = easu_lds_bytes(prec, in_fmt, cellsW, cellsH);
    e = (e + 15) & ~(size_t)15;
    size_t midCell = 16;
    if (prec != PREC_FP32_STRICT && easu_fast_pitch(cellsW) != 0) {
        const size_t ncell = (size_t)easu_fast_pitch(cellsW) * cellsH;
        // the luma plane doubles as the near-tie list region (fused_kernel) and is at least that large
        if (ncell * 4 < kFusedTieListBytes) e += kFusedTieListBytes;
    }
    (void)mid_fmt;
    return e + (size_t)(kTileW + 2) * (kTileH + 2) * midCell;
}

template <int I, int M, int O>
static hipError_t fused_go3(bool strict, const FusedArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const int pitch = easu_fast_pitch(a.cellsW);
    // the EASU planes plus the 34x34 intermediate can exceed the 64 KiB default cap on dynamic LDS (160 KiB per CU).  The
    // attribute is per DEVICE (a one-process node driver holds one ctx per device, examples/bench_node.c): latched per
    // (kernel instantiation, device), one bit per device ordinal.
    auto raise = [](const void *fn, std::atomic<uint64_t> &done) -> hipError_t {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLdsMax);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    };
    if (strict) {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = raise(reinterpret_cast<const void *>(&ovrfsr_strict::fused_kernel<I, M, O, 0>), done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ovrfsr_strict::fused_kernel<I, M, O, 0>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    } else if (pitch == 32) {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = raise(reinterpret_cast<const void *>(&ovrfsr_fast::fused_kernel<I, M, O, 32, kFusedThreads>), done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ovrfsr_fast::fused_kernel<I, M, O, 32, kFusedThreads>), grid, dim3(kFusedThreads), lds, s, with_lds(a, lds));
    } else if (pitch == 40) {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = raise(reinterpret_cast<const void *>(&ovrfsr_fast::fused_kernel<I, M, O, 40, kFusedThreads>), done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ovrfsr_fast::fused_kernel<I, M, O, 40, kFusedThreads>), grid, dim3(kFusedThreads), lds, s, with_lds(a, lds));
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// The intermediate has the submission's format (R8G8B8A8 -> UNORM8, RGBA16F -> half: PostProcessor::IntermediateFormat, which
// mirrors DetermineOutputFormat, PostProcessor.cpp:63-74) or, with quantize_intermediate = 0, stays in float: those are the only
// (input, intermediate) pairs a ctx ever asks for, so only they are instantiated (12 of the 27 format triples of the fused and
// outside-tile kernels were dead weight in the library until round 4).
template <int I, int M> constexpr bool mid_reachable() { return M == I || M == FMT_RGBA32F; }

template <int I, int O>
static hipError_t fused_go(int mid_fmt, bool strict, const FusedArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    if (mid_fmt == (int)I) return fused_go3<I, I, O>(strict, a, grid, lds, s);
    if (mid_fmt == FMT_RGBA32F) return fused_go3<I, FMT_RGBA32F, O>(strict, a, grid, lds, s);
    return hipErrorInvalidValue;
}

hipError_t launch_fused(int prec, int in_fmt, int mid_fmt, int out_fmt, const FusedArgs &a_in, uint32_t batch, hipStream_t s, uint32_t nTiles)
{
    launch_fresh();
    FusedArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (prec != PREC_FP32 && prec != PREC_FP32_STRICT) return hipErrorInvalidValue;
    const bool strict = prec == PREC_FP32_STRICT;
    if (!strict && easu_fast_pitch(a.cellsW) == 0) return hipErrorInvalidValue;
    const dim3 grid(a.tileList ? nTiles : a.tilesX * a.tilesY, 1, batch);
    // OVRFSR_FUSED_LDS_PAD=<bytes> (diagnostic): extra dynamic LDS nobody touches -> fewer workgroups per CU; measures how the kernel's
    // throughput follows its occupancy (profiles/r04_fused_variants.txt) before anyone rebuilds its planes to gain a workgroup
    static const size_t pad = [] { const char *e = std::getenv("OVRFSR_FUSED_LDS_PAD"); const long v = e ? std::atol(e) : 0; return (size_t)(v > 0 ? v : 0); }();
    const size_t need = fused_lds_bytes(prec, in_fmt, mid_fmt, a.cellsW, a.cellsH);
    if (need > kFusedLdsMax) return hipErrorInvalidValue; // never launch with less LDS than the plane layout assumes (callers pre-check: PrepareResources)
    const size_t lds = need + pad > kFusedLdsMax ? kFusedLdsMax : need + pad; // only the diagnostic pad is clamped
    OVRFSR_DISPATCH_FMT3(fused_go, mid_fmt, strict, a, grid, lds, s)
}

template <int I, int O>
static hipError_t easu_outside_go(int mid_fmt, const EasuArgs &a, dim3 grid, hipStream_t s)
{
    if (mid_fmt < 0) hipLaunchKernelGGL((ovrfsr_fast::easu_outside_kernel<I, O, -1>), grid, dim3(kThreads), 0, s, a);
    else if (mid_fmt == FMT_RGBA32F) hipLaunchKernelGGL((ovrfsr_fast::easu_outside_kernel<I, O, FMT_RGBA32F>), grid, dim3(kThreads), 0, s, a);
    else if (mid_fmt == (int)I && I != FMT_RGB10A2) hipLaunchKernelGGL((ovrfsr_fast::easu_outside_kernel<I, O, (I == FMT_RGB10A2 ? -1 : I)>), grid, dim3(kThreads), 0, s, a);
    else return hipErrorInvalidValue; // not an (input, intermediate) pair a ctx produces (see mid_reachable)
    return hipGetLastError();
}

// nTiles blocks, each resolving tile a.tileList[block]: tiles entirely outside the radius (product build only).
// mid_fmt < 0: EASU pass only; mid_fmt >= 0: write the FINAL pixel of the EASU->RCAS pipeline (RCAS outside the radius
// is a tinted copy of the intermediate texel, so the intermediate's format rounding is applied in registers).
// LDS-staged outside-tile kernel (outside_staged_kernel): upscaling only, RGBA8 sources (any destination format).
// RGBA16F sources stay on the per-pixel kernel: stand-alone the staged form is 11 % faster there too (C5: 1020 -> 904 us),
// but its 20 KB of LDS per workgroup cannot co-reside with three 52 KB fused-kernel workgroups per CU, and the overlapped
// step gets 20 % slower.
// mid_fmt < 0: the EASU pass alone; >= 0: final pixel = tint(value read back from a mid_fmt intermediate), RGBA32F = tint
// of the un-rounded value (also NIS DirectCopy, tileH = 24).
bool outside_staged_ok(const BatchView &v, int in_fmt) { return v.inW <= v.outW && v.inH <= v.outH && in_fmt == FMT_RGBA8; }

template <int TH, int I, int O>
static hipError_t outside_staged_go(int mid_fmt, const OutsideArgs &a, dim3 grid, hipStream_t s)
{
    [[maybe_unused]] const size_t lds = (size_t)a.lds_rows * 3 * kOutsidePitch * sizeof(float); // [row][channel][kOutsidePitch]
    if constexpr (I != FMT_RGBA8) {
        return hipErrorInvalidValue;
    } else if constexpr (TH == 24) { // NIS DirectCopy
        hipLaunchKernelGGL((ovrfsr_fast::outside_staged_kernel<24, I, O, FMT_RGBA32F>), grid, dim3(8 * 24), lds, s, with_lds(a, lds));
    } else {
        switch (mid_fmt) { // RGBA8 sources: a UNORM8 or a float intermediate, or the EASU pass alone
        case FMT_RGBA8: hipLaunchKernelGGL((ovrfsr_fast::outside_staged_kernel<32, I, O, FMT_RGBA8>), grid, dim3(256), lds, s, with_lds(a, lds)); break;
        case FMT_RGBA32F: hipLaunchKernelGGL((ovrfsr_fast::outside_staged_kernel<32, I, O, FMT_RGBA32F>), grid, dim3(256), lds, s, with_lds(a, lds)); break;
        case FMT_RGBA16F: return hipErrorInvalidValue;
        default: hipLaunchKernelGGL((ovrfsr_fast::outside_staged_kernel<32, I, O, -1>), grid, dim3(256), lds, s, with_lds(a, lds)); break;
        }
    }
    return hipGetLastError();
}
template <int I, int O> static hipError_t outside_staged_go32(int mid_fmt, const OutsideArgs &a, dim3 grid, hipStream_t s) { return outside_staged_go<32, I, O>(mid_fmt, a, grid, s); }
template <int I, int O> static hipError_t outside_staged_go24(int mid_fmt, const OutsideArgs &a, dim3 grid, hipStream_t s) { return outside_staged_go<24, I, O>(mid_fmt, a, grid, s); }

hipError_t launch_outside_staged(int tileH, int in_fmt, int mid_fmt, int out_fmt, const OutsideArgs &a_in, uint32_t nTiles, uint32_t batch, hipStream_t s)
{
    launch_fresh();
    OutsideArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (!a.tileList || !a.tileRec || !a.bilX || !a.bilY || nTiles == 0 || !outside_staged_ok(a.v, in_fmt)) return hipErrorInvalidValue;
    if (a.lds_cols < 2 || a.lds_cols > 36 || a.lds_rows < 2 || a.lds_rows > 34) return hipErrorInvalidValue;
    // persistent workgroups: block b walks list entries b, b + G, ... (G a multiple of 8: the list is XCD-banded, entry e
    // belongs to band e % 8, so a workgroup stays in its XCD's band); OVRFSR_OUTSIDE_TPW = tiles per workgroup (tuning)
    static const uint32_t tpw = [] { const char *e = std::getenv("OVRFSR_OUTSIDE_TPW"); const int v = e ? std::atoi(e) : 2; return (uint32_t)(v < 1 ? 1 : v); }();
    a.nTiles = nTiles;
    uint32_t G = ((nTiles + tpw - 1) / tpw + 7u) & ~7u;
    if (G > nTiles) G = nTiles;
    const dim3 grid(G, 1, batch);
    if (tileH == 24) { OVRFSR_DISPATCH_FMT3(outside_staged_go24, mid_fmt, a, grid, s) }
    if (tileH != 32) return hipErrorInvalidValue;
    OVRFSR_DISPATCH_FMT3(outside_staged_go32, mid_fmt, a, grid, s)
}

// nTiles blocks, each resolving tile a.tileList[block]: tiles entirely outside the radius (product build only).
hipError_t launch_easu_outside(int in_fmt, int mid_fmt, int out_fmt, const EasuArgs &a_in, uint32_t nTiles, uint32_t batch, hipStream_t s)
{
    launch_fresh();
    EasuArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (!a.tileList || nTiles == 0) return hipErrorInvalidValue;
    if (a.bilX && a.bilY && a.tileRec && outside_staged_ok(a.v, in_fmt)) { // no records: the per-pixel kernel below needs none
        OutsideArgs o;
        o.v = a.v; o.tilesX = a.tilesX; o.tileList = a.tileList; o.tileRec = a.tileRec; o.bilX = a.bilX; o.bilY = a.bilY; o.debug = a.debug;
        o.lds_cols = a.outsideCols; o.lds_rows = a.outsideRows;
        return launch_outside_staged(kTileH, in_fmt, mid_fmt, out_fmt, o, nTiles, batch, s);
    }
    const dim3 grid(nTiles, 1, batch);
    OVRFSR_DISPATCH_FMT(easu_outside_go, mid_fmt, a, grid, s)
}

// B8G8R8A8 -> R8G8B8A8 into a tightly packed buffer (dst pitch = 4*w, image stride = 4*w*h): a byte shuffle, 16 texels
// per thread row segment; the pipeline behind it is the RGBA8 one.
__global__ __launch_bounds__(256) void bgra_to_rgba_kernel(const uint8_t *__restrict__ src, uint32_t srcPitch, uint64_t srcStride,
                                                            uint8_t *__restrict__ dst, uint32_t w, uint32_t h)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
    if (x >= w) return;
    // (checked builds: image `img` of the submission, and the tight copy of the whole batch as one image of gridDim.z * h rows, through the
    // accessors of fsr_bounds.h; the address expressions are the ones the product build has always had)
    OVRFSR_PTR(const uint8_t) s = OVRFSR_IMAGE(const uint8_t, src + (size_t)img * srcStride, srcPitch, (int)w, (int)h, 4u, K_IMAGE_IN);
    OVRFSR_PTR(uint8_t) d = OVRFSR_IMAGE(uint8_t, dst, w * 4u, (int)w, (int)(h * gridDim.z), 4u, K_IMAGE_OUT);
    const uint32_t v = *OVRFSR_AT(const uint32_t, s + (size_t)y * srcPitch + (size_t)x * 4);
    // byte order in memory B,G,R,A -> R,G,B,A: swap bytes 0 and 2
    const uint32_t o = (v & 0xff00ff00u) | ((v >> 16) & 0xffu) | ((v & 0xffu) << 16);
    *OVRFSR_AT(uint32_t, d + ((size_t)img * h + y) * (size_t)w * 4 + (size_t)x * 4) = o;
}

hipError_t launch_bgra_to_rgba(const uint8_t *src, uint32_t srcPitch, uint64_t srcStride, uint8_t *dst, uint32_t w, uint32_t h,
                               uint32_t batch, hipStream_t s)
{
    launch_fresh();
    hipLaunchKernelGGL(bgra_to_rgba_kernel, dim3((w + 255) / 256, h, batch), dim3(256), 0, s, src, srcPitch, srcStride, dst, w, h);
    return hipGetLastError();
}

#ifdef OVRFSR_BOUNDS
// Drives the checked accessors through every kind of violation exactly once (tests/test_gpu_bounds.py asserts the counts): proof that a
// zero from a campaign means "nothing out of bounds", not "nothing checked".  64 threads, 1024 bytes of dynamic LDS.
__global__ void bounds_selftest_kernel(const uint8_t *img, uint32_t ldsBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using namespace ovrfsr_chk;
    const int lane = threadIdx.x;
    ptr<float> plane = carve<float>(reinterpret_cast<float *>(smem), 64, 16, K_SELFTEST, smem, ldsBytes);
    plane[lane] = (float)lane;                                   // 64 accesses inside the plane
    __syncthreads();
    float sink = 0.0f;
    if (lane == 0) sink += plane[64 + 3];                        // inside the declared pad: 1 pad access
    if (lane == 1) sink += plane[64 + 16];                       // behind the pad: out of bounds
    if (lane == 2) sink += plane[-1];                            // in front of the plane: out of bounds
    // a 12 x 4 image of 4-byte texels with a 64-byte row pitch
    const ptr<const uint8_t> im = image<const uint8_t>(img, 64, 12, 4, 4, K_IMAGE_IN);
    if (lane == 3) sink += (float)*at<const uint32_t>(im + 48);            // row 0, pitch padding: out of bounds
    if (lane == 4) sink += (float)*at<const uint32_t>(im + (3 * 64 + 44)); // the last texel: fine
    if (lane == 5) sink += (float)*at<const uint32_t>(im + (3 * 64 + 48)); // behind the last texel: out of bounds
    if (lane == 6) sink += (float)*at<const uint32_t>(im + 46);            // straddles the end of row 0: out of bounds
    // a plane carved beyond the launch's dynamic LDS: one K_LDS_ALLOC record (thread 0)
    const ptr<float> beyond = carve<float>(reinterpret_cast<float *>(smem) + 250, 16, 0, K_SELFTEST, smem, ldsBytes);
    if (sink == 12345.678f && beyond.p) plane[0] = sink; // keep the reads alive
}
hipError_t bounds_selftest()
{
    uint8_t *img = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&img), 4 * 64);
    if (e != hipSuccess) return e;
    e = hipMemset(img, 1, 4 * 64);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bounds_selftest_kernel, dim3(1), dim3(64), 1024, nullptr, img, 1024u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(img);
    return e;
}
static hipError_t bounds_read_tu(unsigned long long *out, bool reset)
{
    unsigned long long c[ovrfsr_chk::kSlots];
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(c, HIP_SYMBOL(ovrfsr_chk::g_counts), sizeof c);
    if (e != hipSuccess) return e;
    for (int i = 0; i < ovrfsr_chk::kFirstRec; ++i) out[i] += c[i];
    if (out[ovrfsr_chk::kFirstRec] == 0 && c[ovrfsr_chk::kFirstRec] != 0)
        for (int i = ovrfsr_chk::kFirstRec; i < ovrfsr_chk::kSlots; ++i) out[i] = c[i];
    if (reset) {
        for (unsigned long long &v : c) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(ovrfsr_chk::g_counts), c, sizeof c);
    }
    return e;
}
hipError_t bounds_read_fsr(unsigned long long *out, bool reset) { return bounds_read_tu(out, reset); }
#else
hipError_t bounds_read_fsr(unsigned long long *, bool) { return hipErrorNotSupported; }
hipError_t bounds_selftest() { return hipErrorNotSupported; }
#endif

// audit build: read (and optionally clear) the current device's counters; product build: hipErrorNotSupported
hipError_t tie_audit_read(unsigned long long out[6], bool reset)
{
#ifdef OVRFSR_TIE_AUDIT
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ovrfsr_tie_audit), 6 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ovrfsr_tie_audit), z, sizeof z);
    }
    return e;
#else
    (void)out; (void)reset;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_easu(int prec, int in_fmt, int out_fmt, const EasuArgs &a_in, uint32_t batch, hipStream_t s, uint32_t nTiles)
{
    launch_fresh();
    EasuArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (prec != PREC_FP32 && prec != PREC_FP32_STRICT) return hipErrorInvalidValue;
    const bool strict = prec == PREC_FP32_STRICT;
    const dim3 grid(a.tileList ? nTiles : a.tilesX * a.tilesY, 1, batch);
    const size_t lds = easu_lds_bytes(prec, in_fmt, a.cellsW, a.cellsH);
    OVRFSR_DISPATCH_FMT(easu_go, strict, a, grid, lds, s)
}

hipError_t launch_rcas(int prec, int in_fmt, int out_fmt, const RcasArgs &a_in, uint32_t batch, hipStream_t s, uint32_t nTiles)
{
    launch_fresh();
    RcasArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    a.dppTilesXMagic = div_magic((uint32_t)(a.v.outW + kRcasDppTileW - 1) / kRcasDppTileW);
    if (prec != PREC_FP32 && prec != PREC_FP32_STRICT) return hipErrorInvalidValue;
    const bool strict = prec == PREC_FP32_STRICT;
    if (a.tileList && (strict || nTiles == 0)) return hipErrorInvalidValue; // lists are a product-build feature
    const dim3 grid(a.tileList ? nTiles : a.tilesX * a.tilesY, 1, batch);
    OVRFSR_DISPATCH_FMT(rcas_go, strict, a, grid, s)
}

} // namespace ovrfsr
