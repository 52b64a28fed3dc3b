// fsr_params.h -- kernel argument blocks shared by the host launch manager and the gfx950 kernels.
// These play the role of the reference's constant buffers: UpscaleConstants
// (src/postprocess/PostProcessor.cpp:276-283) and SharpenConstants (:403-407).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ovrfsr {

enum : int { FMT_RGBA8 = 0, FMT_RGBA16F = 1, FMT_RGBA32F = 2, FMT_RGB10A2 = 3 };
enum : int { PREC_FP32 = 0, PREC_FP32_STRICT = 2 }; // ovrfsr_precision (1 is not a mode)

// mask_mode: every 16x16 group inside the radius / every group outside / mixed (test per group)
enum : uint32_t { MASK_ALL_INSIDE = 0, MASK_ALL_OUTSIDE = 1, MASK_MIXED = 2 };

constexpr int kTileW = 32;  // output pixels per workgroup tile (two 16-px mask groups wide)
constexpr int kTileH = 32;
constexpr int kThreads = 256;
constexpr int kLumPadRows = 5;    // rows past the EASU luma plane that the 4-rows-per-lane analysis sweep may read (allocated, never written)
// dynamic LDS the fused kernel may ask for: the 160 KiB of a CU minus its static LDS (row / column tables of stage 1, list counters: < 2 KiB)
constexpr size_t kFusedLdsMax = 158 * 1024;
#ifndef OVRFSR_FUSED_NT
#define OVRFSR_FUSED_NT 256
#endif
constexpr int kFusedThreads = OVRFSR_FUSED_NT; // threads per workgroup of the product build's fused kernel (one 32x32 tile either way)
// fused kernel: waves x sweeps x 64 near-tie entries (uint16), kept in the luma plane
constexpr size_t kFusedTieListBytes = (size_t)(kFusedThreads / 64) * (((kTileW + 2) * (kTileH + 2) + kFusedThreads - 1) / kFusedThreads) * 64 * 2;
constexpr int kOutsidePitch = 40; // outside_staged_kernel: floats per channel row of its planar LDS texel plane (>= 36 columns)
constexpr int kRcasDppTileW = 62; // rcas_dpp_kernel: a wave = 64 consecutive columns, 62 stored (2 halo lanes)
constexpr int kRcasDppTileH = 32; //                  4 waves x 8 rows per lane
constexpr uint64_t kRcasResident = 8ull * 256; // workgroups of rcas_dpp_kernel an MI355X holds at once (8 per CU: 48 VGPRs, no LDS)

// q = n / d for n*d < 2^32 as one scalar multiply-high: magic = floor(2^32/d) + 1 (0 = "d is 1").  Tile indices are
// workgroup-uniform, but the hardware has no scalar divide: `tile / tilesX` costs ~20 VALU instructions per thread.
static inline uint32_t div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }

// -DOVRFSR_BOUNDS builds (fsr_bounds.h) tell every kernel how much dynamic LDS its launch allocated, so that the planes it carves
// can be checked against it; the product build's argument blocks do not carry the field.
#ifdef OVRFSR_BOUNDS
#define OVRFSR_ARGS_LDS_BYTES uint32_t ldsBytes = 0;
#else
#define OVRFSR_ARGS_LDS_BYTES
#endif

struct BatchView {          // image i of a batch lives at base + i*stride
    const uint8_t *in;
    uint8_t *out;
    uint64_t in_stride;     // bytes
    uint64_t out_stride;    // bytes
    uint32_t in_pitch;      // bytes per row
    uint32_t out_pitch;
    int32_t inW, inH, outW, outH;
};

// Bilinear fallback mapping of one output column / row (fsr_easu.hlsl:33-36 with D3D11 8-bit sub-texel
// addressing), built on the host with the same IEEE operations the kernels and the oracle use.
struct BilinTap { int32_t i0; float frac; };

struct MaskArgs {           // imageCentre / radius of the reference's cbuffers, one set per eye
    uint32_t centre[2][4];  // [eye][c1x, c1y, c2x, c2y]
    uint32_t r2;            // radius[1]
    uint32_t mode[2];       // per eye: MASK_*
    uint32_t first_eye;     // eye of image 0
    uint32_t alternate;     // eye of image i = first_eye ^ (i & alternate)
};

struct EasuArgs {
    BatchView v;
    float sx, sy, cx, cy;   // FsrEasuCon con0 (the only row integer addressing needs)
    MaskArgs m;
    int32_t cellsW, cellsH; // LDS input tile extent (max over tiles) incl. the 1+2 apron
    uint32_t tilesX, tilesY;
    uint32_t tilesXMagic;     // div_magic(tilesX), filled by the launchers
    const BilinTap *bilX;   // [outW], [outH] device tables for the bilinear fallback (product build)
    const BilinTap *bilY;
    const uint32_t *tileList; // optional: tile index of each block (mask-sorted launch); null = all tiles in XCD order
    const uint32_t *tileRec;  // records parallel to tileList (OutsideArgs::tileRec), for the lists the staged outside kernel walks
    uint32_t debug;           // RCAS const0[3]; only read by the "final" outside kernel (tinted copy of the fused path)
    uint32_t outsideCols, outsideRows; // largest bilinear footprint of a 32x32 tile (outside_staged_kernel's LDS plane)
    float rcpOutW, rcpOutH;   // RN(1/outW), RN(1/outH): o/out as mul + 2 fma (Markstein), see div_exact
    uint32_t rcpExact;        // host verified that form against IEEE division for every o < outW (outH); else 0
    float tieHalfMin;         // near-tie guard of RGBA16F stores: values below it are not guarded (see half_tie_code); +inf = off
    uint32_t ringStrips;      // mask-sorted form: a listed tile with no group inside the radius is a RING tile whose output only
                              // feeds the RCAS taps of inside neighbours -- write just the edge pixels those taps read
    OVRFSR_ARGS_LDS_BYTES
};

// LDS-staged bilinear fallback / DirectCopy of mask-sorted tiles entirely outside the radius (product build,
// outside_staged_kernel)
struct OutsideArgs {
    BatchView v;
    uint32_t tilesX;          // tiles (32 x TH output pixels) per row
    uint32_t tilesXMagic;     // div_magic(tilesX), filled by the launcher
    const uint32_t *tileList;
    const BilinTap *bilX;     // host-built column / row taps (see BilinTap); column taps padded to a multiple of 32 with copies of the last
    const BilinTap *bilY;
    uint32_t debug;
    uint32_t lds_cols, lds_rows; // LDS texel plane extent (set by launch_outside_staged from the tap tables' host copy)
    const uint32_t *tileRec;     // 4 dwords per list entry (host-built, parallel to tileList): ox0 | oy0 << 16, (X0+1) | (Y0+1) << 16,
                                 // colsN | rowsN << 8, 0 -- tile origin, footprint origin and extent (see outside_staged_kernel)
    uint32_t nTiles;             // list length; the kernel is persistent: block b walks entries b, b + gridDim.x, ...
    OVRFSR_ARGS_LDS_BYTES
};

struct RcasArgs {
    BatchView v;
    float sharp;            // FsrRcasCon con[0] as float
    uint32_t debug;         // const0[3]
    MaskArgs m;
    uint32_t tilesX, tilesY;
    uint32_t tilesXMagic;     // div_magic(tilesX), filled by the launchers
    const uint32_t *tileList; // optional mask-sorted tile list (see EasuArgs); product build only
    uint32_t dppTilesXMagic;  // div_magic of rcas_dpp_kernel's own tile count per row (62-pixel tiles)
    // masked RGBA8 pipelines (mask-sorted form): 62-column segments of the runs of tiles touching the radius, two dwords each
    // (x0 | tileY << 16, xEnd), built by the host next to the tile lists; nSpans = 0: walk tileList with rcas_direct_kernel
    const uint32_t *spanRec;
    uint32_t nSpans;
};

struct FusedArgs {
    BatchView v;
    float sx, sy, cx, cy;
    float sharp;
    uint32_t debug;
    MaskArgs m;
    int32_t cellsW, cellsH;
    uint32_t tilesX, tilesY;
    uint32_t tilesXMagic;     // div_magic(tilesX), filled by the launchers
    const uint32_t *tileList; // optional mask-sorted tile list (see EasuArgs)
    float tieHalfMin;         // near-tie guard of a half intermediate (see EasuArgs)
    const BilinTap *bilX;     // [outW], [outH] column / row taps of the bilinear fallback (product build: groups outside the
    const BilinTap *bilY;     // radius inside a tile that touches it take their texels from the LDS colour plane)
    OVRFSR_ARGS_LDS_BYTES
};

struct NisArgs {            // the NISConfig cbuffer (NIS_Upscale.hlsl:28-68) minus the unused viewport fields
    BatchView v;
    float kDetectRatio, kDetectThres, kMinContrastRatio, kRatioNorm;
    float kContrastBoost, kEps, kSharpStartY, kSharpScaleY;
    float kSharpStrengthMin, kSharpStrengthScale, kSharpLimitMin, kSharpLimitScale;
    float kScaleX, kScaleY, kDstNormX, kDstNormY;
    float kSrcNormX, kSrcNormY;
    float reserved1;        // debugMode as float (PostProcessor.cpp:309)
    MaskArgs m;
    const float *coefScale; // device copies of coef_scale / coef_usm, [64][8]
    const float *coefUsm;
    int32_t cellsW, cellsH; // LDS luma/edge tile extent of the scaler (incl. 3-texel ring)
    uint32_t tilesX, tilesY;
    uint32_t tilesXMagic;     // div_magic(tilesX), filled by the launchers
    const uint32_t *tileList; // optional mask-sorted group list (see EasuArgs)
    const uint32_t *tileRec;  // records parallel to tileList (OutsideArgs::tileRec)
    const BilinTap *bilX;     // DirectCopy taps of the mask-sorted outside kernel (see OutsideArgs)
    const BilinTap *bilY;
    uint32_t outsideCols, outsideRows; // largest bilinear footprint of a 32x24 group
    OVRFSR_ARGS_LDS_BYTES
};

} // namespace ovrfsr
