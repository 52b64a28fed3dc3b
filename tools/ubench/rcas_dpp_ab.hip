// tools/ubench/rcas_dpp_ab.hip -- A/B of the "wave64 shuffles for neighbour reuse" idea of BASELINE.json's north_star on the
// one kernel where lanes hold each other's taps: RCAS.  FsrRcasF's left/right taps d, f (ffx_fsr1.h:698-707) are the centre
// taps e of the neighbouring pixels, so a lane can take them from its neighbour lanes with DPP moves instead of loading and
// unpacking them itself.
//   A  the product kernel's form (rcas_direct_kernel): a lane resolves 4 vertically adjacent pixels of one column and loads
//      its 6 centre, 4 left and 4 right taps itself: 14 loads, 42 v_cvt_f32_ubyteN per 4 pixels.
//   B  DPP form: a wave covers 64 consecutive columns of 4 rows; a lane loads only its 6 centre taps (18 v_cvt) and takes
//      the 24 side values from lanes +-1 with v_mov_b32_dpp wave_shr:1 / wave_shl:1.  Lanes 0 and 63 of a wave are halo
//      lanes (they have no neighbour on one side): they load and convert but store nothing, so a wave yields 62 columns.
// Both run the same arithmetic (byte domain, v_rcp_f32, v_cvt_pk_u8_f32) on the interior of a 2244x2492 RGBA8 image
// (BASELINE C2's output size) x 8 images; outputs are compared, times printed.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize rcas_dpp_ab.hip -o rcas_dpp_ab && ./rcas_dpp_ab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define RCAS_LIMIT ((float)(0.25 - (1.0 / 16.0)))

__device__ __forceinline__ float raw_min3(float a, float b, float c) { float r; __asm__("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float raw_max3(float a, float b, float c) { float r; __asm__("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float raw_min(float a, float b) { float r; __asm__("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max(float a, float b) { float r; __asm__("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float prx_med_rcp(float a)
{
    float b = __uint_as_float(0x7ef19fffu - __float_as_uint(a));
    return b * (-b * a + 2.0f);
}
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 unpack(uint32_t v) { return {(float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu)}; }

__device__ __forceinline__ uint32_t rcas_px(const f3 &b, const f3 &d, const f3 &e, const f3 &f, const f3 &h, float sharp)
{
    const float PEAK = 255.0f;
    const float mnR = raw_min(raw_min3(b.x, d.x, f.x), h.x), mxR = raw_max(raw_max3(b.x, d.x, f.x), h.x);
    const float mnG = raw_min(raw_min3(b.y, d.y, f.y), h.y), mxG = raw_max(raw_max3(b.y, d.y, f.y), h.y);
    const float mnB = raw_min(raw_min3(b.z, d.z, f.z), h.z), mxB = raw_max(raw_max3(b.z, d.z, f.z), h.z);
    const float hitMinR = mnR * __builtin_amdgcn_rcpf(4.0f * mxR), hitMinG = mnG * __builtin_amdgcn_rcpf(4.0f * mxG), hitMinB = mnB * __builtin_amdgcn_rcpf(4.0f * mxB);
    const float hitMaxR = (PEAK - mxR) * __builtin_amdgcn_rcpf(4.0f * mnR + -4.0f * PEAK);
    const float hitMaxG = (PEAK - mxG) * __builtin_amdgcn_rcpf(4.0f * mnG + -4.0f * PEAK);
    const float hitMaxB = (PEAK - mxB) * __builtin_amdgcn_rcpf(4.0f * mnB + -4.0f * PEAK);
    const float lobeR = fmaxf(-hitMinR, hitMaxR), lobeG = fmaxf(-hitMinG, hitMaxG), lobeB = fmaxf(-hitMinB, hitMaxB);
    const float lobe = __builtin_amdgcn_fmed3f(fmaxf(lobeR, fmaxf(lobeG, lobeB)), -RCAS_LIMIT, 0.0f) * sharp;
    const float rcpL = prx_med_rcp(4.0f * lobe + 1.0f);
    const float pr = (lobe * ((b.x + d.x) + (h.x + f.x)) + e.x) * rcpL;
    const float pg = (lobe * ((b.y + d.y) + (h.y + f.y)) + e.y) * rcpL;
    const float pb = (lobe * ((b.z + d.z) + (h.z + f.z)) + e.z) * rcpL;
    uint32_t v = 0xff000000u;
    v = __builtin_amdgcn_cvt_pk_u8_f32(pr, 0, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(pg, 1, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(pb, 2, v);
    return v;
}

// interior region: x in [X0, X0 + cols), y in [Y0, Y0 + rows); every tap is inside the image
constexpr int X0 = 64, Y0 = 4;

// A: 32 columns x 32 rows per 256-thread workgroup, 4 rows per lane (the product kernel's tile)
__global__ __launch_bounds__(256) void rcas_direct(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t pitch, uint64_t stride,
                                                   int cols, int rows, float sharp)
{
    const uint8_t *src = in + blockIdx.z * stride;
    uint8_t *dst = out + blockIdx.z * stride;
    const int lx = threadIdx.x & 31, ly = (threadIdx.x >> 5) * 4;
    const int x = blockIdx.x * 32 + lx, y = blockIdx.y * 32 + ly;
    if (x >= cols || y >= rows) return;
    const int ox = X0 + x, oy = Y0 + y;
    f3 c[6], l[4], r[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = unpack(*reinterpret_cast<const uint32_t *>(src + (uint32_t)(oy - 1 + i) * pitch + (uint32_t)ox * 4u));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        l[i] = unpack(*reinterpret_cast<const uint32_t *>(src + (uint32_t)(oy + i) * pitch + (uint32_t)(ox - 1) * 4u));
        r[i] = unpack(*reinterpret_cast<const uint32_t *>(src + (uint32_t)(oy + i) * pitch + (uint32_t)(ox + 1) * 4u));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (y + i < rows) *reinterpret_cast<uint32_t *>(dst + (uint32_t)(oy + i) * pitch + (uint32_t)ox * 4u) = rcas_px(c[i], l[i], c[i + 1], r[i], c[i + 2], sharp);
}

// gfx9 DPP controls: wave_shr:1 (lane i reads lane i-1) = 0x138, wave_shl:1 (lane i reads lane i+1) = 0x130
__device__ __forceinline__ float from_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float from_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false)); }

// B: a wave = 64 consecutive columns (62 stored) x 4 rows; 4 waves stacked = 16 rows per workgroup
__global__ __launch_bounds__(256) void rcas_dpp(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t pitch, uint64_t stride,
                                                int cols, int rows, float sharp)
{
    const uint8_t *src = in + blockIdx.z * stride;
    uint8_t *dst = out + blockIdx.z * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 62 + lane - 1, y = blockIdx.y * 16 + wave * 4;   // lane 0 / 63: halo columns
    if (y >= rows) return;                                                         // wave-uniform
    const int ox = X0 + min(x, cols), oy = Y0 + y;                                 // halo / tail lanes read a valid texel
    f3 c[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) c[i] = unpack(*reinterpret_cast<const uint32_t *>(src + (uint32_t)(oy - 1 + i) * pitch + (uint32_t)ox * 4u));
    f3 l[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        l[i] = {from_left(c[i + 1].x), from_left(c[i + 1].y), from_left(c[i + 1].z)};
        r[i] = {from_right(c[i + 1].x), from_right(c[i + 1].y), from_right(c[i + 1].z)};
    }
    const bool store = lane >= 1 && lane <= 62 && x < cols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t v = rcas_px(c[i], l[i], c[i + 1], r[i], c[i + 2], sharp);
        if (store && y + i < rows) *reinterpret_cast<uint32_t *>(dst + (uint32_t)(oy + i) * pitch + (uint32_t)ox * 4u) = v;
    }
}

// C: like B, but a lane owns TWO adjacent columns (one 8-byte load per row): 128 columns per wave (124 stored), 3 loads and
// 9 v_cvt per 4 pixels... and only the outer side of each column pair comes from a neighbour lane (12 DPP moves per 8 px)
__global__ __launch_bounds__(256) void rcas_dpp2(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t pitch, uint64_t stride,
                                                 int cols, int rows, float sharp)
{
    const uint8_t *src = in + blockIdx.z * stride;
    uint8_t *dst = out + blockIdx.z * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 124 + 2 * lane - 2, y = blockIdx.y * 16 + wave * 4;   // lane 0 / 63: halo column pairs
    if (y >= rows) return;
    const int ox = X0 + min(x, cols), oy = Y0 + y;
    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    f3 c0[6], c1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const u32x2_a4 v = *reinterpret_cast<const u32x2_a4 *>(src + (uint32_t)(oy - 1 + i) * pitch + (uint32_t)ox * 4u);
        c0[i] = unpack(v.x); c1[i] = unpack(v.y);
    }
    const bool store = lane >= 1 && lane <= 62 && x + 1 < cols;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f3 l = {from_left(c1[i + 1].x), from_left(c1[i + 1].y), from_left(c1[i + 1].z)};
        const f3 r = {from_right(c0[i + 1].x), from_right(c0[i + 1].y), from_right(c0[i + 1].z)};
        u32x2_a4 o;
        o.x = rcas_px(c0[i], l, c0[i + 1], c1[i + 1], c0[i + 2], sharp);
        o.y = rcas_px(c1[i], c0[i + 1], c1[i + 1], r, c1[i + 2], sharp);
        if (store && y + i < rows) *reinterpret_cast<u32x2_a4 *>(dst + (uint32_t)(oy + i) * pitch + (uint32_t)ox * 4u) = o;
    }
}

__global__ void fill(uint32_t *p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // smooth-ish content with noise: keeps RCAS off the degenerate all-noise lobe
        p[i] = (h & 0x0f0f0fu) + (((uint32_t)(i % 2244) >> 3) & 0xffu) * 0x010101u % 0xf0f0f0u | 0xff000000u;
    }
}

template <typename K>
static float time_kernel(K launch, int iters)
{
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms / iters;
}

int main()
{
    const int W = 2244, H = 2492, N = 32;
    const uint32_t pitch = W * 4;
    const uint64_t stride = (uint64_t)pitch * H;
    const int cols = W - 2 * X0, rows = H - 2 * Y0;
    uint8_t *in, *outA, *outB;
    hipMalloc(&in, stride * N); hipMalloc(&outA, stride * N); hipMalloc(&outB, stride * N);
    hipMemset(outA, 0, stride * N); hipMemset(outB, 0, stride * N);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(in), (size_t)stride * N / 4, 12345u);
    const float sharp = 0.87055f;
    auto A = [&] { hipLaunchKernelGGL(rcas_direct, dim3((cols + 31) / 32, (rows + 31) / 32, N), dim3(256), 0, 0, in, outA, pitch, stride, cols, rows, sharp); };
    auto B = [&] { hipLaunchKernelGGL(rcas_dpp, dim3((cols + 61) / 62, (rows + 15) / 16, N), dim3(256), 0, 0, in, outB, pitch, stride, cols, rows, sharp); };
    // interleaved timing: the chip clocks to its power budget, so alternate the two forms
    uint8_t *outC;
    hipMalloc(&outC, stride * N); hipMemset(outC, 0, stride * N);
    auto Cc = [&] { hipLaunchKernelGGL(rcas_dpp2, dim3((cols + 123) / 124, (rows + 15) / 16, N), dim3(256), 0, 0, in, outC, pitch, stride, cols, rows, sharp); };
    float tA = 0, tB = 0, tC = 0;
    for (int r = 0; r < 3; ++r) { tA += time_kernel(A, 20); tB += time_kernel(B, 20); tC += time_kernel(Cc, 20); }
    tA /= 3; tB /= 3; tC /= 3;
    hipDeviceSynchronize();
    std::vector<uint32_t> a((size_t)stride / 4), b((size_t)stride / 4);
    size_t diff = 0, total = 0;
    for (int img : {0, N - 1}) {
        hipMemcpy(a.data(), outA + img * stride, stride, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), outB + img * stride, stride, hipMemcpyDeviceToHost);
        for (int y = Y0; y < Y0 + rows; ++y)
            for (int x = X0; x < X0 + cols; ++x) { diff += a[(size_t)y * W + x] != b[(size_t)y * W + x]; ++total; }
    }
    const double px = (double)cols * rows * N, bytes = px * 8.0;
    printf("RCAS interior %dx%d x %d images (RGBA8 -> RGBA8), sharpness con %.5f\n", cols, rows, N, sharp);
    printf("A direct loads  (14 loads, 42 cvt per 4 px)        : %.4f ms/launch = %.2f us/eye, %.0f GB/s\n", tA, tA * 1e3 / N, bytes / tA / 1e6);
    printf("B DPP neighbours (6 loads, 18 cvt + 24 DPP mov / 4 px, 62 of 64 lanes store): %.4f ms/launch = %.2f us/eye, %.0f GB/s\n", tB, tB * 1e3 / N, bytes / tB / 1e6);
    printf("B/A time ratio %.3f; outputs differ in %zu of %zu pixels\n", tB / tA, diff, total);
    size_t diffC = 0;
    const int colsC = (cols / 124) * 124;   // C stores whole column pairs only: compare the columns every form wrote
    for (int img : {0, N - 1}) {
        hipMemcpy(a.data(), outA + img * stride, stride, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), outC + img * stride, stride, hipMemcpyDeviceToHost);
        for (int y = Y0; y < Y0 + rows; ++y)
            for (int x = X0; x < X0 + colsC; ++x) diffC += a[(size_t)y * W + x] != b[(size_t)y * W + x];
    }
    printf("C DPP, 2 columns per lane (3 loads, 18 cvt + 12 DPP mov / 8 px, 124 of 128 columns stored): %.4f ms/launch = %.2f us/eye, %.0f GB/s; C/A %.3f; differs in %zu pixels\n",
           tC, tC * 1e3 / N, bytes / tC / 1e6, tC / tA, diffC);
    return diff != 0;
}
