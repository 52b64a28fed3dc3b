// oracle/shim_probe.cpp -- TEST INFRASTRUCTURE ONLY.  Exposes the fixed-function D3D11 behaviours that
// oracle/hlsl_shim.hpp restates (and that oracle/fsr_oracle.c / nis_oracle.c restate again in C) one at a
// time, so tests/test_d3d_semantics.py can pin each of them in isolation: Gather4 component order and
// footprint, clamp addressing, Load out-of-bounds = 0, 8-bit sub-texel bilinear, UNORM stores.
// Built from the shim alone (no reference sources needed): make -C oracle libshimprobe.so
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include "hlsl_shim.hpp"
using namespace hlsl;

extern "C" {
#define PROBE __attribute__((visibility("default")))

// Texture2D::Gather{Red,Green,Blue}(sampler, uv): out[4] = .x .y .z .w
PROBE void probe_gather(const float *rgba, int w, int h, float u, float v, int channel, float out[4])
{
    Texture2D t; t.px = rgba; t.w = w; t.h = h;
    SamplerState s;
    float4 r = channel == 0 ? t.GatherRed(s, float2(u, v), int2(0, 0)) : channel == 1 ? t.GatherGreen(s, float2(u, v), int2(0, 0))
                                                                                      : t.GatherBlue(s, float2(u, v), int2(0, 0));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// Texture2D::SampleLevel(linearClamp, uv, 0)
PROBE void probe_sample(const float *rgba, int w, int h, float u, float v, float out[4])
{
    Texture2D t; t.px = rgba; t.w = w; t.h = h;
    SamplerState s;
    float4 r = t.SampleLevel(s, float2(u, v), 0.0f);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// Texture2D::Load(int3(x, y, 0)) and operator[]
PROBE void probe_load(const float *rgba, int w, int h, int x, int y, float out[4])
{
    Texture2D t; t.px = rgba; t.w = w; t.h = h;
    float4 r = t.Load(int3(x, y, 0));
    float4 q = t[int2(x, y)];
    if (r.x != q.x || r.y != q.y || r.z != q.z || r.w != q.w) std::abort(); // the two spellings are one operation
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// RWTexture2D<unorm float4> store: what lands in the resource for a written value
PROBE void probe_unorm_store(const float in[4], float out[4])
{
    float buf[4] = {-1, -1, -1, -1};
    RWTexture2D t; t.px = buf; t.w = 1; t.h = 1; t.unorm_clamp = true;
    t[uint2(0, 0)] = float4(in[0], in[1], in[2], in[3]);
    for (int i = 0; i < 4; ++i) out[i] = buf[i];
}
}
