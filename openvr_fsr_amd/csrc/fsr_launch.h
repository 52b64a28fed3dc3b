// fsr_launch.h -- typed launchers implemented in fsr_kernels.hip, used by the host launch manager.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include "fsr_params.h"

namespace ovrfsr {
size_t easu_lds_bytes(int prec, int in_fmt, int cellsW, int cellsH);
// HIP keeps a host thread's last error until somebody reads it, and a kernel launch reports its failure only there.  Every launch_* below
// reads (= clears) it BEFORE launching, so that what it returns afterwards is its own launch's -- not an error the HOST's code left behind:
// a failed hipMalloc of the caller's, handled through its return value, used to fail the next frame as "EASU launch: out of memory"
// (tests/debug/stale_error.c, round 6).
inline void launch_fresh() { (void)hipGetLastError(); }

hipError_t launch_easu(int prec, int in_fmt, int out_fmt, const EasuArgs &a, uint32_t batch, hipStream_t s, uint32_t nTiles = 0);
hipError_t launch_easu_outside(int in_fmt, int mid_fmt, int out_fmt, const EasuArgs &a, uint32_t nTiles, uint32_t batch, hipStream_t s);
size_t fused_lds_bytes(int prec, int in_fmt, int mid_fmt, int cellsW, int cellsH);
hipError_t launch_fused(int prec, int in_fmt, int mid_fmt, int out_fmt, const FusedArgs &a, uint32_t batch, hipStream_t s, uint32_t nTiles = 0);
hipError_t launch_rcas(int prec, int in_fmt, int out_fmt, const RcasArgs &a, uint32_t batch, hipStream_t s, uint32_t nTiles = 0);
int nis_pitch(int cellsW);
size_t nis_scaler_lds_bytes(int cellsW, int cellsH);
hipError_t launch_nis_scaler(int prec, int in_fmt, int out_fmt, const NisArgs &a, uint32_t batch, hipStream_t s, uint32_t nGroups = 0);
hipError_t launch_bgra_to_rgba(const uint8_t *src, uint32_t srcPitch, uint64_t srcStride, uint8_t *dst, uint32_t w, uint32_t h,
                              uint32_t batch, hipStream_t s);
bool outside_staged_ok(const BatchView &v, int in_fmt);
hipError_t launch_outside_staged(int tileH, int in_fmt, int mid_fmt, int out_fmt, const OutsideArgs &a, uint32_t nTiles, uint32_t batch, hipStream_t s);
hipError_t launch_nis_outside(int in_fmt, int out_fmt, const NisArgs &a, uint32_t nGroups, uint32_t batch, hipStream_t s);
hipError_t launch_nis_sharpen(int prec, int in_fmt, int out_fmt, const NisArgs &a, uint32_t batch, hipStream_t s);
hipError_t tie_audit_read(unsigned long long out[6], bool reset); // -DOVRFSR_TIE_AUDIT builds only (fsr_kernels.hip)
// -DOVRFSR_BOUNDS builds only (fsr_bounds.h): the checked accessors' counters of the current device, one array per kernel translation
// unit (fsr_kernels.hip / nis_kernels.hip), ADDED into out[ovrfsr_chk::kSlots] (the first-hit record: copied if out has none yet);
// and a launch that drives the accessors through every kind of violation once (fsr_kernels.hip)
hipError_t bounds_read_fsr(unsigned long long *out, bool reset);
hipError_t bounds_read_nis(unsigned long long *out, bool reset);
hipError_t bounds_selftest();
// the kernel argument block as launched: checked builds add the dynamic LDS size of the launch
#ifdef OVRFSR_BOUNDS
template <typename A> static inline A with_lds(A a, size_t lds) { a.ldsBytes = (uint32_t)lds; return a; }
#else
template <typename A> static inline const A &with_lds(const A &a, size_t) { return a; }
#endif
} // namespace ovrfsr
