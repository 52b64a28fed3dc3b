// fsr_bounds.h -- checked memory accessors of the kernels (round 6; SURVEY.md section 5: "LDS-bounds asserts in debug kernels").
//
// Every LDS plane, every image and every device table the kernels touch is declared through the macros below.
//   * PRODUCT build (OVRFSR_BOUNDS undefined): the macros expand to the plain pointer expressions the kernels were written with
//     -- the machine code of the shipped library is unchanged, instruction for instruction (tools/isa_fingerprint.py --diff).
//   * CHECKED build (make EXTRA=-DOVRFSR_BOUNDS ... -> ab/bounds.so; never shipped, never loaded by the product): the same names
//     are fat pointers (ovrfsr_chk::ptr<T>) that know the extent of what they point into.  Every dereference is tested:
//       - inside [lo, hi): fine, counted as a checked access;
//       - inside [hi, padhi): the plane's DECLARED pad -- allocated bytes the kernel is allowed to read and never relies on (the
//         luma rows behind the analysis sweep, the tap-table quads behind the last column ...): counted per kind, so a test can
//         assert that pad reads happen only where the design says they do, and never beyond the declared extent;
//       - anywhere else: an out-of-bounds access.  It is COUNTED (per kind), its first occurrence is recorded (kind, byte offset,
//         access size, plane size, workgroup), and the access is redirected to the plane's first element -- the kernel runs on,
//         nothing faults, the test reads the counters and fails.
//     Images are additionally checked per ROW: an access must lie inside the `width * texel` valid bytes of its row, not in the
//     pitch padding between rows (legal memory, not image).
// The counters live in one __device__ array per translation unit (fsr_kernels.hip, nis_kernels.hip); capi.cpp sums them in
// ovrfsr_debug_bounds() (exported by checked builds only).  tests/test_gpu_bounds.py drives the campaigns; results:
// profiles/r06_bounds.txt.
//
// This matches what the reference does about failures in this path -- it has no checked build of its shaders at all; D3D11 makes
// out-of-bounds SRV/UAV accesses well-defined (reads return 0, writes are dropped: PostProcessor.cpp binds typed views, :387-392)
// and a failed resource build disables the processor (:145-152).  A HIP kernel has no such net: an out-of-plane LDS read returns
// whatever the neighbouring plane holds and an out-of-image global read may fault, and neither changes a stored pixel often
// enough for a parity test to notice.
#pragma once
#include <stdint.h>

namespace ovrfsr_chk {
// what a checked pointer points into.  tests/test_gpu_bounds.py parses this list (name order = counter index).
enum Kind : uint32_t {
    K_IMAGE_IN = 0,   // an input image (global): extent (H-1)*pitch + W*texel, per-row valid bytes W*texel
    K_IMAGE_OUT,      // an output / intermediate image (global)
    K_TILE_LIST,      // EasuArgs/RcasArgs/FusedArgs/NisArgs::tileList (global): one entry per workgroup of the launch
    K_TILE_REC,       // OutsideArgs::tileRec: 4 dwords per list entry
    K_SPAN_REC,       // RcasArgs::spanRec: 2 dwords per segment
    K_BIL_X,          // column taps [outW] (+ declared pad: up to the next multiple of 32 -- the quads of the last tile behind the last column, copies of the last tap)
    K_BIL_Y,          // row taps [outH]
    K_NIS_COEF,       // coef_scale / coef_usm device banks, 512 floats each
    K_EASU_COL,       // LDS colour plane of EASU / the fused kernel
    K_EASU_ANA,       // LDS analysis plane
    K_EASU_LUM,       // LDS luma plane (+ declared pad: kLumPadRows rows behind it, read by the 4-rows-per-lane analysis sweep)
    K_EASU_ROWINFO,   // LDS per-row table of easu_fast_kernel
    K_TIE_LIST,       // LDS near-tie lists (easu_fast_kernel: static; fused_kernel: inside the luma plane)
    K_TIE_CNT,        // LDS per-wave list lengths
    K_FUSED_MID,      // LDS 34x34 intermediate plane of the fused kernel
    K_RCAS_TILE,      // LDS 34x34 tile of the strict RCAS kernel
    K_OUTSIDE_TEX,    // LDS planar texel plane of outside_staged_kernel
    K_NIS_YU,         // LDS unit-luma plane of NVScaler (later the vertical-sum plane V)
    K_NIS_Y255,       // LDS x255 luma plane
    K_NIS_EDGE,       // LDS edge-map plane
    K_NIS_COEF_LDS,   // LDS coefficient banks
    K_NIS_RAW,        // LDS raw-texel plane
    K_NIS_ROWINFO,    // LDS per-row table of nis_scaler_kernel
    K_NIS_SHARPEN_Y,  // LDS 36x36 luma tile of NVSharpen
    K_LDS_ALLOC,      // a plane was carved beyond the dynamic LDS the launch allocated (checked when the plane is declared)
    K_SELFTEST,       // the self-test kernel's plane (ovrfsr_debug_bounds_selftest)
    K_COUNT
};
// counter layout (unsigned long long each): [k] out-of-bounds accesses of kind k, [K_COUNT + k] accesses inside the declared pad,
// [2*K_COUNT + k] checked accesses (lanes) in total, then the first out-of-bounds record: {kind + 1 (0 = none), byte offset from the
// plane base (two's complement), access bytes, plane bytes (without pad), blockIdx.x | blockIdx.z << 32}
constexpr int kFirstRec = 3 * K_COUNT;
constexpr int kSlots = 3 * K_COUNT + 5;
} // namespace ovrfsr_chk

#ifdef OVRFSR_BOUNDS
#include <hip/hip_runtime.h>
#include <type_traits>

// MUTATION of the checker (-DOVRFSR_BOUNDS -DOVRFSR_BOUNDS_SHRINK=1, never a test dependency): every plane, table and image is DECLARED one
// element / one texel column shorter than it is.  A campaign against that build must light up every kind whose last element a kernel really
// touches -- the proof, in the kernels themselves rather than in the self-test, that each check is live (profiles/r06_bounds.txt, 1g).
#ifndef OVRFSR_BOUNDS_SHRINK
#define OVRFSR_BOUNDS_SHRINK 0
#endif

namespace ovrfsr_chk {
static __device__ unsigned long long g_counts[kSlots]; // one array per translation unit (internal linkage)

__device__ __forceinline__ void count_access(uint32_t kind)
{
    // one atomic per wave-instruction, issued by the first active lane, adds the number of active lanes
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u)
        atomicAdd(&g_counts[2 * K_COUNT + kind], (unsigned long long)__builtin_popcountll(m));
}
__device__ __forceinline__ void record_oob(uint32_t kind, long long off, unsigned long long bytes, unsigned long long plane)
{
    atomicAdd(&g_counts[kind], 1ull);
    if (atomicCAS(&g_counts[kFirstRec], 0ull, (unsigned long long)kind + 1ull) == 0ull) {
        g_counts[kFirstRec + 1] = (unsigned long long)off;
        g_counts[kFirstRec + 2] = bytes;
        g_counts[kFirstRec + 3] = plane;
        g_counts[kFirstRec + 4] = (unsigned long long)blockIdx.x | ((unsigned long long)blockIdx.z << 32);
    }
}

// The fat pointer.  T may be const-qualified.  Arithmetic moves `p` only; lo / hi / padhi stay with the plane.
template <typename T> struct ptr {
    T *p;
    const char *lo, *hi, *padhi;
    uint32_t kind;
    uint32_t rowPitch, rowBytes; // images: rowBytes valid bytes every rowPitch bytes; 0 = a flat plane

    __device__ __forceinline__ ptr() : p(nullptr), lo(nullptr), hi(nullptr), padhi(nullptr), kind(K_COUNT), rowPitch(0), rowBytes(0) {}
    __device__ __forceinline__ ptr(decltype(nullptr)) : ptr() {}
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
    __device__ __forceinline__ ptr(T *base, unsigned long long n, unsigned long long padN, uint32_t k)
        : p(base), lo(reinterpret_cast<const char *>(base)), hi(reinterpret_cast<const char *>(base) + (n > OVRFSR_BOUNDS_SHRINK ? n - OVRFSR_BOUNDS_SHRINK : n) * sizeof(T)),
          padhi(reinterpret_cast<const char *>(base) + (padN ? n + padN : (n > OVRFSR_BOUNDS_SHRINK ? n - OVRFSR_BOUNDS_SHRINK : n)) * sizeof(T)), kind(k), rowPitch(0), rowBytes(0) {}

    // the address of an access of `bytes` bytes at q, or the plane base when q is out of bounds (counted)
    __device__ __forceinline__ const char *check(const char *q, unsigned long long bytes) const
    {
        count_access(kind);
        bool ok = q >= lo && q + bytes <= hi;
        if (ok && rowPitch != 0u) {
            const unsigned long long off = (unsigned long long)(q - lo);
            ok = off % rowPitch + bytes <= rowBytes; // not in the pitch padding between rows
        }
        if (ok) return q;
        if (rowPitch == 0u && q >= lo && q + bytes <= padhi) { // reaches into the declared pad: counted, allowed
            atomicAdd(&g_counts[K_COUNT + kind], 1ull);
            return q;
        }
        record_oob(kind, (long long)(q - lo), bytes, (unsigned long long)(hi - lo));
        return lo;
    }
    template <typename I> __device__ __forceinline__ T &operator[](I i) const
    {
        return *const_cast<T *>(reinterpret_cast<const T *>(check(reinterpret_cast<const char *>(p + i), sizeof(T))));
    }
    __device__ __forceinline__ T &operator*() const { return (*this)[0]; }
    template <typename I> __device__ __forceinline__ ptr operator+(I i) const { ptr r = *this; r.p = p + i; return r; }
    template <typename I> __device__ __forceinline__ ptr operator-(I i) const { ptr r = *this; r.p = p - i; return r; }
    __device__ __forceinline__ operator ptr<const T>() const
    {
        ptr<const T> r; r.p = p; r.lo = lo; r.hi = hi; r.padhi = padhi; r.kind = kind; r.rowPitch = rowPitch; r.rowBytes = rowBytes; return r;
    }
};

// reinterpret the element type, keep the plane (OVRFSR_AS)
template <typename U, typename T> __device__ __forceinline__ ptr<U> cast(const ptr<T> &s)
{
    static_assert(std::is_const<U>::value || !std::is_const<T>::value, "cast drops const");
    ptr<U> r; r.p = reinterpret_cast<U *>(const_cast<typename std::remove_const<T>::type *>(s.p));
    r.lo = s.lo; r.hi = s.hi; r.padhi = s.padhi; r.kind = s.kind; r.rowPitch = s.rowPitch; r.rowBytes = s.rowBytes;
    return r;
}
// validated raw address of ONE access of sizeof(U) at s (OVRFSR_AT): the caller dereferences with its own (possibly under-aligned) type
template <typename U, typename T> __device__ __forceinline__ U *at(const ptr<T> &s)
{
    static_assert(std::is_const<U>::value || !std::is_const<T>::value, "access drops const");
    return const_cast<U *>(reinterpret_cast<const U *>(s.check(reinterpret_cast<const char *>(s.p), sizeof(U))));
}
// the same for an access whose size is not sizeof(U): a 3-element ext_vector is 16 bytes to sizeof and 12 bytes to the load that fetches it
// (global_load_dwordx3) -- checked as 16 it "reaches" 4 bytes past a row it ends flush with, and the redirect that follows corrupts pixels of the
// CHECKED build only (found by running the fuzz seeds against it: 12 of 9 750 instances)
template <typename U, unsigned BYTES, typename T> __device__ __forceinline__ U *at_n(const ptr<T> &s)
{
    static_assert(std::is_const<U>::value || !std::is_const<T>::value, "access drops const");
    return const_cast<U *>(reinterpret_cast<const U *>(s.check(reinterpret_cast<const char *>(s.p), BYTES)));
}
// raw pointers pass through (host-side tables handed to helpers that are also used unchecked) -- not used by the kernels
template <typename U, typename T> __device__ __forceinline__ U *at(T *s) { return reinterpret_cast<U *>(s); }

// an image of a batch: base + i*stride, H rows of W*texel valid bytes every `pitch` bytes
template <typename T> __device__ __forceinline__ ptr<T> image(T *base, uint32_t pitch, int w, int h, uint32_t texel, uint32_t kind)
{
    ptr<T> r;
    r.p = base; r.lo = reinterpret_cast<const char *>(base);
    r.hi = r.padhi = r.lo + ((unsigned long long)(h - 1) * pitch + (unsigned long long)w * texel);
    r.kind = kind; r.rowPitch = pitch; r.rowBytes = (uint32_t)(w > OVRFSR_BOUNDS_SHRINK ? w - OVRFSR_BOUNDS_SHRINK : w) * texel;
    return r;
}
// a plane carved from the dynamic LDS of the launch: [base, base + n + pad) must lie inside [smem, smem + ldsBytes)
template <typename T> __device__ __forceinline__ ptr<T> carve(T *base, unsigned long long n, unsigned long long padN, uint32_t kind,
                                                            const unsigned char *smem, uint32_t ldsBytes)
{
    ptr<T> r(base, n, padN, kind);
    if (reinterpret_cast<const char *>(base) < reinterpret_cast<const char *>(smem) || r.padhi > reinterpret_cast<const char *>(smem) + ldsBytes) {
        if (threadIdx.x == 0) record_oob(K_LDS_ALLOC, (long long)(r.padhi - reinterpret_cast<const char *>(smem)), (unsigned long long)kind, ldsBytes);
    }
    return r;
}
template <typename T> __device__ __forceinline__ T *raw(const ptr<T> &s) { return s.p; }

// LDS POISON.  A workgroup's LDS holds whatever the previous workgroup on that CU left there; several kernels read cells they never stage
// (the declared pads above; border columns of the analysis sweep; planes a branch did not fill) on the argument that the value never reaches a
// stored pixel.  Checked builds make that argument testable: every byte of the launch's dynamic LDS and of every static LDS array is overwritten
// with 0xff (as float: a NaN; as a list index: 65535, far outside every plane) by the whole workgroup before the kernel's first own write.  If a
// stored pixel depended on an unstaged cell it would now be NaN / garbage DETERMINISTICALLY, and the parity tests, which also run against the
// checked build (tests/test_gpu_bounds.py, tests/debug/fuzz_campaign.py), would fail.  Must be called in workgroup-uniform control flow.
__device__ __forceinline__ void poison(void *base, unsigned long long bytes)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(base);
    for (unsigned long long i = threadIdx.x; i < bytes / 4; i += blockDim.x) w[i] = 0xffffffffu;
    unsigned char *b = reinterpret_cast<unsigned char *>(base);
    for (unsigned long long i = (bytes & ~3ull) + threadIdx.x; i < bytes; i += blockDim.x) b[i] = 0xffu;
    __syncthreads();
}
} // namespace ovrfsr_chk

#define OVRFSR_PTR(T) ovrfsr_chk::ptr<T>                                   /* a pointer variable / parameter */
#define OVRFSR_PTR_R(T) ovrfsr_chk::ptr<T>                                 /* ... that is __restrict__ in the product build */
#define OVRFSR_PLANE(T, base, n, pad, kind) ovrfsr_chk::ptr<T>((base), (n), (pad), ovrfsr_chk::kind)
#define OVRFSR_PLANE_C(T, base, n, pad, kind) ovrfsr_chk::ptr<T>(const_cast<T *>(base), (n), (pad), ovrfsr_chk::kind) /* a read-only table handed to an OVRFSR_PTR_RC parameter */
#define OVRFSR_PTR_C(T) ovrfsr_chk::ptr<T>                                 /* `const T *` parameter of the product build */
#define OVRFSR_PTR_RC(T) ovrfsr_chk::ptr<T>                                /* `const T *__restrict__` parameter of the product build */
#define OVRFSR_CARVE(T, base, n, pad, kind, smem, ldsBytes) ovrfsr_chk::carve<T>((base), (n), (pad), ovrfsr_chk::kind, (smem), (ldsBytes))
#define OVRFSR_IMAGE(T, base, pitch, w, h, texel, kind) ovrfsr_chk::image<T>((base), (pitch), (w), (h), (texel), ovrfsr_chk::kind)
#define OVRFSR_AS(T, e) ovrfsr_chk::cast<T>(e)                             /* reinterpret the element type; index / dereference through the result */
#define OVRFSR_AT(T, e) ovrfsr_chk::at<T>(e)                               /* raw T* of one validated access (under-aligned vector typedefs) */
#define OVRFSR_AT_N(T, bytes, e) ovrfsr_chk::at_n<T, bytes>(e)             /* ... of `bytes` bytes (3-element vectors: 12, not sizeof = 16) */
#define OVRFSR_RAW(e) ovrfsr_chk::raw(e)                                   /* the unchecked pointer (to carve the next plane from) */
#define OVRFSR_LDS_ARRAY(T, name, n, kind) __shared__ T name##_lds[n]; ovrfsr_chk::poison(name##_lds, sizeof(T) * (n)); const ovrfsr_chk::ptr<T> name(name##_lds, (n), 0, ovrfsr_chk::kind)
#define OVRFSR_LDS_POISON(smem, ldsBytes) ovrfsr_chk::poison((smem), (ldsBytes))   /* the launch's dynamic LDS, at kernel start */
#else
#define OVRFSR_PTR(T) T *
#define OVRFSR_PTR_R(T) T *__restrict__
#define OVRFSR_PLANE(T, base, n, pad, kind) (base)
#define OVRFSR_PLANE_C(T, base, n, pad, kind) (base)
#define OVRFSR_PTR_C(T) const T *
#define OVRFSR_PTR_RC(T) const T *__restrict__
#define OVRFSR_CARVE(T, base, n, pad, kind, smem, ldsBytes) (base)
#define OVRFSR_IMAGE(T, base, pitch, w, h, texel, kind) (base)
#define OVRFSR_AS(T, e) reinterpret_cast<T *>(e)
#define OVRFSR_AT(T, e) reinterpret_cast<T *>(e)
#define OVRFSR_AT_N(T, bytes, e) reinterpret_cast<T *>(e)
#define OVRFSR_RAW(e) (e)
#define OVRFSR_LDS_ARRAY(T, name, n, kind) __shared__ T name[n]
#define OVRFSR_LDS_POISON(smem, ldsBytes)
#endif
