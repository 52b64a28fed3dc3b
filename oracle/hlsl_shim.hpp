// oracle/hlsl_shim.hpp -- a small HLSL-on-C++ emulation layer.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: let g++ compile the reference's *own* HLSL bodies (FsrEasuF / FsrRcasF in
// /root/reference/src/fsr/ffx_fsr1.h, the shader entry points in fsr/fsr_easu.hlsl and
// fsr/fsr_rcas.hlsl, and nis/NIS_Scaler.h) where they lie, so that the C restatement in
// oracle/fsr_oracle.c / oracle/nis_oracle.c can be pinned bit-for-bit against real reference code.
// oracle/build_ref.py does the (mechanical, committed) source transformation; nothing from the
// reference is stored in this repository.
//
// Everything lives in namespace hlsl so that min/max/abs/... resolve to the HLSL-semantics
// overloads below and never to <cstdlib>'s integer abs or std::min.
//
// Semantics chosen (documented in DESIGN.md "oracle"): IEEE fp32 evaluated as written (no FMA
// contraction: build with -ffp-contract=off), rcp(x)=1.0f/x, rsqrt(x)=1.0f/sqrtf(x), min/max
// ignore NaN (D3D11 functional spec), saturate(NaN)=0, uint arithmetic wraps mod 2^32.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <ucontext.h>
#include <vector>

namespace hlsl {

typedef uint32_t uint;

struct float2; struct float3; struct float4;
struct uint2; struct uint3; struct uint4;
struct int2; struct int3;

template <class V, class T, int A, int B> struct Swz2 {
    T v[4];
    operator V() const { return V(v[A], v[B]); }
};
template <class V, class T, int A, int B, int C> struct Swz3 {
    T v[4];
    operator V() const { return V(v[A], v[B], v[C]); }
};

struct float2 {
    float x, y;
    float2() : x(0), y(0) {}
    float2(float a, float b) : x(a), y(b) {}
    explicit float2(float a) : x(a), y(a) {}
    explicit float2(const uint2 &u);
    explicit float2(const int2 &u);
};
struct float3 {
    union { struct { float x, y, z; }; struct { float r, g, b; }; };
    float3() : x(0), y(0), z(0) {}
    float3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit float3(float a) : x(a), y(a), z(a) {}
};
struct float4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        Swz2<float2, float, 0, 1> xy;
        Swz2<float2, float, 2, 3> zw;
        Swz3<float3, float, 0, 1, 2> rgb;
        Swz3<float3, float, 0, 1, 2> xyz;
    };
    float4() : x(0), y(0), z(0), w(0) {}
    float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    float4(const float3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    explicit float4(float a) : x(a), y(a), z(a), w(a) {}
};
struct uint2 {
    uint x, y;
    uint2() : x(0), y(0) {}
    uint2(uint a, uint b) : x(a), y(b) {}
    uint2(const int2 &i); // HLSL converts int<->uint vectors implicitly
};
struct int2 {
    int x, y;
    int2() : x(0), y(0) {}
    int2(int a, int b) : x(a), y(b) {}
    int2(const uint2 &u) : x((int)u.x), y((int)u.y) {}
};
inline uint2::uint2(const int2 &i) : x((uint)i.x), y((uint)i.y) {}
inline float2::float2(const uint2 &u) : x((float)u.x), y((float)u.y) {}
inline float2::float2(const int2 &u) : x((float)u.x), y((float)u.y) {}
struct uint3 {
    union { struct { uint x, y, z; }; Swz2<uint2, uint, 0, 1> xy; };
    uint3() : x(0), y(0), z(0) {}
    uint3(uint a, uint b, uint c) : x(a), y(b), z(c) {}
};
struct int3 {
    int x, y, z;
    int3(int a, int b, int c) : x(a), y(b), z(c) {}
    int3(const int2 &v, int c) : x(v.x), y(v.y), z(c) {}
};
struct uint4 {
    union {
        struct { uint x, y, z, w; };
        Swz2<uint2, uint, 0, 1> xy;
        Swz2<uint2, uint, 2, 3> zw;
    };
    uint4() : x(0), y(0), z(0), w(0) {}
    uint4(uint a, uint b, uint c, uint d) : x(a), y(b), z(c), w(d) {}
    uint &operator[](int i) { return (&x)[i]; }
    const uint &operator[](int i) const { return (&x)[i]; }
};

// ---- component-wise operators ---------------------------------------------------------------
#define HLSL_BINOP2(V, T, op)                                                                      \
    inline V operator op(const V &a, const V &b) { return V(a.x op b.x, a.y op b.y); }             \
    inline V operator op(const V &a, T b) { return V(a.x op b, a.y op b); }                        \
    inline V operator op(T a, const V &b) { return V(a op b.x, a op b.y); }                        \
    inline V &operator op##=(V &a, const V &b) { a.x op## = b.x; a.y op## = b.y; return a; }       \
    inline V &operator op##=(V &a, T b) { a.x op## = b; a.y op## = b; return a; }
#define HLSL_BINOP3(V, T, op)                                                                      \
    inline V operator op(const V &a, const V &b) { return V(a.x op b.x, a.y op b.y, a.z op b.z); } \
    inline V operator op(const V &a, T b) { return V(a.x op b, a.y op b, a.z op b); }              \
    inline V operator op(T a, const V &b) { return V(a op b.x, a op b.y, a op b.z); }              \
    inline V &operator op##=(V &a, const V &b) { a.x op## = b.x; a.y op## = b.y; a.z op## = b.z; return a; } \
    inline V &operator op##=(V &a, T b) { a.x op## = b; a.y op## = b; a.z op## = b; return a; }
#define HLSL_BINOP4(V, T, op)                                                                      \
    inline V operator op(const V &a, const V &b) { return V(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); } \
    inline V operator op(const V &a, T b) { return V(a.x op b, a.y op b, a.z op b, a.w op b); }    \
    inline V operator op(T a, const V &b) { return V(a op b.x, a op b.y, a op b.z, a op b.w); }    \
    inline V &operator op##=(V &a, const V &b) { a.x op## = b.x; a.y op## = b.y; a.z op## = b.z; a.w op## = b.w; return a; } \
    inline V &operator op##=(V &a, T b) { a.x op## = b; a.y op## = b; a.z op## = b; a.w op## = b; return a; }
#define HLSL_ARITH(M, V, T) M(V, T, +) M(V, T, -) M(V, T, *) M(V, T, /)
HLSL_ARITH(HLSL_BINOP2, float2, float)
HLSL_ARITH(HLSL_BINOP3, float3, float)
HLSL_ARITH(HLSL_BINOP4, float4, float)
HLSL_ARITH(HLSL_BINOP2, uint2, uint)
HLSL_ARITH(HLSL_BINOP2, int2, int)
inline float2 operator-(const float2 &a) { return float2(-a.x, -a.y); }
// float2 / uint2 (e.g. float2(pos) / Radius.zw): uint -> float conversion, then true division
inline float2 operator/(const float2 &a, const uint2 &b) { return float2(a.x / (float)b.x, a.y / (float)b.y); }

// ---- intrinsics -------------------------------------------------------------------------------
inline float asfloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float asfloat(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float asfloat(float f) { return f; }
inline uint asuint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
inline uint asuint(uint u) { return u; }
inline float2 asfloat(const uint2 &u) { return float2(asfloat(u.x), asfloat(u.y)); }
inline uint2 asuint(const float2 &f) { return uint2(asuint(f.x), asuint(f.y)); }

inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline uint min(uint a, uint b) { return a < b ? a : b; }
inline uint max(uint a, uint b) { return a > b ? a : b; }
inline float abs(float a) { return fabsf(a); }
inline int abs(int a) { return a < 0 ? -a : a; }
inline float floor(float a) { return floorf(a); }
inline float ceil(float a) { return ceilf(a); }
inline float saturate(float a) { return fminf(fmaxf(a, 0.0f), 1.0f); }
inline float rcp(float a) { return 1.0f / a; }
inline float rsqrt(float a) { return 1.0f / sqrtf(a); }
inline float sqrt(float a) { return sqrtf(a); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); } // D3D: x + s(y - x)
#define HLSL_MAP2(V, fn) inline V fn(const V &a, const V &b) { return V(fn(a.x, b.x), fn(a.y, b.y)); }
#define HLSL_MAP3(V, fn) inline V fn(const V &a, const V &b) { return V(fn(a.x, b.x), fn(a.y, b.y), fn(a.z, b.z)); }
#define HLSL_MAP4(V, fn) inline V fn(const V &a, const V &b) { return V(fn(a.x, b.x), fn(a.y, b.y), fn(a.z, b.z), fn(a.w, b.w)); }
HLSL_MAP2(float2, min) HLSL_MAP2(float2, max) HLSL_MAP3(float3, min) HLSL_MAP3(float3, max)
HLSL_MAP4(float4, min) HLSL_MAP4(float4, max)
inline float2 floor(const float2 &a) { return float2(floorf(a.x), floorf(a.y)); }
inline float2 abs(const float2 &a) { return float2(fabsf(a.x), fabsf(a.y)); }
inline float3 saturate(const float3 &a) { return float3(saturate(a.x), saturate(a.y), saturate(a.z)); }
inline uint dot(const uint2 &a, const uint2 &b) { return a.x * b.x + a.y * b.y; }
inline float dot(const float2 &a, const float2 &b) { return a.x * b.x + a.y * b.y; }

// ---- resources --------------------------------------------------------------------------------
struct SamplerState {};

// A float RGBA image (4 floats per texel) bound as Texture2D<float4>; linear/clamp sampler only
// (that is what a NULL sampler slot gives on D3D11 -- SURVEY.md appendix A).
//
// The four fixed-function behaviours restated here, where they are specified, and where each is pinned in isolation
// (tests/test_d3d_semantics.py through oracle/shim_probe.cpp).  The D3D11.3 Functional Specification is public
// (microsoft.github.io/DirectX-Specs); chapter/section titles below are quoted from memory -- this container has no
// network -- so the d3d11.h constants and the reference's own comments are given as the checkable anchors:
//  (1) Gather4 footprint and component order: spec "gather4" instruction (Shader Model 4.1+/5 instruction reference,
//      ch. 22): the 2x2 bilinear footprint of the unnormalised coordinate u*W-0.5, returned as
//      (-,+),(+,+),(+,-),(-,-) = .x bottom-left, .y bottom-right, .z top-right, .w top-left (same order in the HLSL docs
//      of Texture2D::Gather).  The reference relies on exactly this: "Gather 4 ordering: a b / r g" and the names
//      bczz / ijfe / klhg / zzon, src/fsr/ffx_fsr1.h:333-360.  Test: test_gather_component_order,
//      test_easu_gather_positions_match_the_tap_comments.
//  (2) Clamp addressing (D3D11_TEXTURE_ADDRESS_CLAMP, the default sampler state's mode): spec "Texture Addressing"
//      (sampler state, ch. 7.18): every texel index of the footprint is clamped to [0, size-1] individually.
//      Test: test_clamp_addressing.
//  (3) ld / Texture2D::Load / operator[] out of range: spec "ld" instruction: "out of bounds addresses ... return 0 in
//      all components" (the FSR sources depend on it: RCAS's border taps).  Test: test_load_out_of_bounds_is_zero,
//      test_oracle_rcas_border_equals_explicit_zero_ring.
//  (4) Bilinear weights: spec "Texture Filtering / fixed point texel addressing": texel-space coordinates are converted to
//      fixed point with D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8 fractional bits (d3d11.h; the same value is reported as
//      sub-texel precision by every D3D11 hardware tier), round to nearest, before the footprint and the weights are
//      derived.  Test: test_bilinear_has_8_subtexel_bits, test_oracle_bilinear_fallback_equals_shim.
//  UNORM stores (RWTexture2D<unorm float4>): spec "Floating point to UNORM conversion": saturate (NaN -> 0), scale,
//  +0.5, truncate; FSR documents the same at src/fsr/ffx_fsr1.h:1075-1080.  Test: test_unorm_store.
struct Texture2D {
    const float *px = nullptr;
    int w = 0, h = 0;
    static int cl(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
    float4 at_clamp(int x, int y) const {
        const float *p = px + 4 * ((size_t)cl(y, h - 1) * w + cl(x, w - 1));
        return float4(p[0], p[1], p[2], p[3]);
    }
    float4 at_zero(int x, int y) const { // Load / operator[]: out of bounds reads return 0
        if (x < 0 || y < 0 || x >= w || y >= h) return float4();
        const float *p = px + 4 * ((size_t)y * w + x);
        return float4(p[0], p[1], p[2], p[3]);
    }
    // Gather4: texel-space coordinate u*W-0.5, footprint (floor, floor)+{0,1}^2, result order
    // x=(0,1) y=(1,1) z=(1,0) w=(0,0)  [BL, BR, TR, TL]
    float4 gather(const float2 &uv, int ch) const {
        float tx = uv.x * (float)w - 0.5f, ty = uv.y * (float)h - 0.5f;
        int x0 = (int)floorf(tx), y0 = (int)floorf(ty);
        float4 tl = at_clamp(x0, y0), tr = at_clamp(x0 + 1, y0);
        float4 bl = at_clamp(x0, y0 + 1), br = at_clamp(x0 + 1, y0 + 1);
        return float4((&bl.x)[ch], (&br.x)[ch], (&tr.x)[ch], (&tl.x)[ch]);
    }
    float4 GatherRed(const SamplerState &, const float2 &uv, const int2 &) const { return gather(uv, 0); }
    float4 GatherGreen(const SamplerState &, const float2 &uv, const int2 &) const { return gather(uv, 1); }
    float4 GatherBlue(const SamplerState &, const float2 &uv, const int2 &) const { return gather(uv, 2); }
    // bilinear with D3D11's fixed-point texel addressing: the texel-space coordinate u*W-0.5 is
    // snapped to D3D11_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8 fractional bits (round to nearest) before the
    // footprint and the weights are derived -- a sample at a texel centre returns that texel exactly.
    static void fixed8(float t, int &i0, float &frac) {
        float s = floorf(t * 256.0f + 0.5f);
        float f = floorf(s * (1.0f / 256.0f));
        i0 = (int)f;
        frac = (s - f * 256.0f) * (1.0f / 256.0f);
    }
    float4 SampleLevel(const SamplerState &, const float2 &uv, float) const {
        float tx = uv.x * (float)w - 0.5f, ty = uv.y * (float)h - 0.5f;
        int x0, y0; float fx, fy;
        fixed8(tx, x0, fx); fixed8(ty, y0, fy);
        float4 c00 = at_clamp(x0, y0), c10 = at_clamp(x0 + 1, y0);
        float4 c01 = at_clamp(x0, y0 + 1), c11 = at_clamp(x0 + 1, y0 + 1);
        float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy);
        float w01 = (1.0f - fx) * fy, w11 = fx * fy;
        return ((c00 * w00 + c10 * w10) + c01 * w01) + c11 * w11;
    }
    float4 Load(const int3 &p) const { return at_zero(p.x, p.y); }
    float4 operator[](const uint2 &p) const { return at_zero((int)p.x, (int)p.y); }
    float4 operator[](const int2 &p) const { return at_zero(p.x, p.y); }
};

// RWTexture2D<float4> (and the "unorm float4" flavour, which clamps to [0,1] on store)
struct RWTexture2D {
    float *px = nullptr;
    int w = 0, h = 0;
    bool unorm_clamp = false;
    struct Ref {
        RWTexture2D *t; int x, y;
        void operator=(const float4 &v) const {
            if (x < 0 || y < 0 || x >= t->w || y >= t->h) return; // OOB UAV writes are dropped
            float *p = t->px + 4 * ((size_t)y * t->w + x);
            if (t->unorm_clamp) { p[0] = saturate(v.x); p[1] = saturate(v.y); p[2] = saturate(v.z); p[3] = saturate(v.w); }
            else { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
        }
    };
    Ref operator[](const uint2 &p) { return Ref{this, (int)p.x, (int)p.y}; }
    Ref operator[](const int2 &p) { return Ref{this, p.x, p.y}; }
};

// ---- thread-group emulation with a real barrier (fibers) ----------------------------------------
// GroupMemoryBarrierWithGroupSync() yields to the group scheduler, which resumes every thread of
// the group round-robin until all have finished.  All threads reach the same barriers in uniform
// control flow, which is what the shaders do.
struct GroupRunner {
    static GroupRunner *&cur() { static GroupRunner *g = nullptr; return g; }
    ucontext_t sched;
    std::vector<ucontext_t> ctx;
    std::vector<std::vector<char>> stacks;
    std::vector<char> done;
    int running = -1;
    void (*body)(void *, int) = nullptr;
    void *arg = nullptr;
    static void tramp() {
        GroupRunner *g = cur();
        int me = g->running;
        g->body(g->arg, me);
        g->done[me] = 1;
        swapcontext(&g->ctx[me], &g->sched);
    }
    void run(int nthreads, void (*b)(void *, int), void *a, size_t stack = 256 * 1024) {
        body = b; arg = a;
        ctx.resize(nthreads); stacks.resize(nthreads); done.assign(nthreads, 0);
        for (int i = 0; i < nthreads; ++i) {
            stacks[i].resize(stack);
            getcontext(&ctx[i]);
            ctx[i].uc_stack.ss_sp = stacks[i].data();
            ctx[i].uc_stack.ss_size = stack;
            ctx[i].uc_link = nullptr;
            makecontext(&ctx[i], (void (*)())tramp, 0);
        }
        GroupRunner *prev = cur();
        cur() = this;
        for (;;) {
            bool any = false;
            for (int i = 0; i < nthreads; ++i) {
                if (done[i]) continue;
                any = true;
                running = i;
                swapcontext(&sched, &ctx[i]);
            }
            if (!any) break;
        }
        cur() = prev;
    }
    void yield() { int me = running; swapcontext(&ctx[me], &sched); }
};
inline void GroupMemoryBarrierWithGroupSync() { GroupRunner::cur()->yield(); }

} // namespace hlsl
