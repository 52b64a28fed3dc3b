// nis_kernels.hip -- instantiates the gfx950 NIS kernels (product + strict builds) and their launchers.
#include <hip/hip_runtime.h>
#include "fsr_params.h"
#include "fsr_launch.h"
#include "fsr_bounds.h"

namespace ovrfsr_fast {
#define OVRFSR_STRICT 0
#pragma clang fp contract(fast)
#include "fsr_device.inc"
#include "nis_kernels.inc"
#undef OVRFSR_STRICT
} // namespace ovrfsr_fast

namespace ovrfsr_strict {
#define OVRFSR_STRICT 1
#pragma clang fp contract(off)
#include "fsr_device.inc"
#include "nis_kernels.inc"
#undef OVRFSR_STRICT
} // namespace ovrfsr_strict
#pragma clang fp contract(on)

namespace ovrfsr {

int nis_pitch(int cellsW) { return cellsW <= 32 ? 32 : cellsW <= 40 ? 40 : 0; }
size_t nis_scaler_lds_bytes(int cellsW, int cellsH) { return (size_t)nis_pitch(cellsW) * cellsH * (4 + 4 + 16 + 4) + 2 * 512 * 4; } // Yu, Y255, E, raw texel

template <int I, int O>
static hipError_t scaler_go(bool strict, const NisArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    const int pitch = nis_pitch(a.cellsW);
    if (pitch == 32) {
        if (strict) hipLaunchKernelGGL((ovrfsr_strict::nis_scaler_kernel<I, O, 32>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
        else hipLaunchKernelGGL((ovrfsr_fast::nis_scaler_kernel<I, O, 32>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    } else if (pitch == 40) {
        if (strict) hipLaunchKernelGGL((ovrfsr_strict::nis_scaler_kernel<I, O, 40>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
        else hipLaunchKernelGGL((ovrfsr_fast::nis_scaler_kernel<I, O, 40>), grid, dim3(kThreads), lds, s, with_lds(a, lds));
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <int I, int O>
static hipError_t sharpen_go(bool strict, const NisArgs &a, dim3 grid, hipStream_t s)
{
    if (strict) hipLaunchKernelGGL((ovrfsr_strict::nis_sharpen_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((ovrfsr_fast::nis_sharpen_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
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

template <int I, int O>
static hipError_t nis_outside_go(const NisArgs &a, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((ovrfsr_fast::nis_outside_kernel<I, O>), grid, dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_nis_outside(int in_fmt, int out_fmt, const NisArgs &a_in, uint32_t nGroups, uint32_t batch, hipStream_t s)
{
    launch_fresh();
    NisArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (!a.tileList || nGroups == 0) return hipErrorInvalidValue;
    if (a.bilX && a.bilY && a.tileRec && outside_staged_ok(a.v, in_fmt)) { // no records: the per-pixel kernel below needs none
        OutsideArgs o;
        o.v = a.v; o.tilesX = a.tilesX; o.tileList = a.tileList; o.tileRec = a.tileRec; o.bilX = a.bilX; o.bilY = a.bilY; o.debug = a.reserved1 != 0.0f ? 1u : 0u;
        o.lds_cols = a.outsideCols; o.lds_rows = a.outsideRows;
        return launch_outside_staged(24, in_fmt, FMT_RGBA32F, out_fmt, o, nGroups, batch, s);
    }
    const dim3 grid(nGroups, 1, batch);
    OVRFSR_DISPATCH_FMT(nis_outside_go, a, grid, s)
}

hipError_t launch_nis_scaler(int prec, int in_fmt, int out_fmt, const NisArgs &a_in, uint32_t batch, hipStream_t s, uint32_t nGroups)
{
    launch_fresh();
    NisArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (prec != PREC_FP32 && prec != PREC_FP32_STRICT) return hipErrorInvalidValue;
    const dim3 grid(a.tileList ? nGroups : a.tilesX * a.tilesY, 1, batch);
    const size_t lds = nis_scaler_lds_bytes(a.cellsW, a.cellsH);
    OVRFSR_DISPATCH_FMT(scaler_go, prec == PREC_FP32_STRICT, a, grid, lds, s)
}

#ifdef OVRFSR_BOUNDS
hipError_t bounds_read_nis(unsigned long long *out, bool reset)
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
#else
hipError_t bounds_read_nis(unsigned long long *, bool) { return hipErrorNotSupported; }
#endif

hipError_t launch_nis_sharpen(int prec, int in_fmt, int out_fmt, const NisArgs &a_in, uint32_t batch, hipStream_t s)
{
    launch_fresh();
    NisArgs a = a_in;
    a.tilesXMagic = div_magic(a.tilesX);
    if (prec != PREC_FP32 && prec != PREC_FP32_STRICT) return hipErrorInvalidValue;
    const dim3 grid(a.tilesX * a.tilesY, 1, batch);
    OVRFSR_DISPATCH_FMT(sharpen_go, prec == PREC_FP32_STRICT, a, grid, s)
}

} // namespace ovrfsr
