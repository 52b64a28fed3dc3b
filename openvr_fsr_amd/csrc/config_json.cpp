// config_json.cpp -- reads the reference's openvr_mod.cfg (JSON with // comments) into ovrfsr_config,
// with the defaulting rules of Config::Load (src/postprocess/Config.h:30-63): every key is optional,
// missing "sharpness" means 1.0 (not the struct default 0.75), negative sharpness becomes 0, and a
// file that cannot be parsed leaves the struct defaults in place ("Could not read config file").
// Hotkey settings are parsed past and ignored (Win32 virtual-key codes have no meaning here).
// Also: a binary PPM dump of a device image, standing in for the F7 DDS capture
// (PostProcessor.cpp:640-657 via ScreenGrab11) as the "image out" format for visual diffs.
#include <hip/hip_runtime.h>
#include <clocale>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../../include/openvr_fsr_amd.h"

namespace {

// The text is parsed in the "C" locale whatever the HOST process has set: a game that calls setlocale(LC_ALL, "") on a machine whose decimal
// separator is a comma makes atof("0.77") return 0 -- renderScale, sharpness and radius all read 0 (round 6; jsoncpp, which the reference
// uses, is locale-independent too).  Numbers go through strtod_l with an explicit C locale, character classes are ASCII by construction.
double c_atof(const char *s)
{
    static const locale_t c = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    return c ? strtod_l(s, nullptr, c) : std::strtod(s, nullptr);
}
inline bool is_space(char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\f' || ch == '\v'; }
inline bool is_alnum(char ch) { return (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'); }

// Minimal JSON reader: objects, arrays, strings, numbers, true/false/null, // and /* */ comments.
// Only the scalar leaves of the top-level "fsr" object are kept, addressed as "key" or "sub.key".
struct Parser {
    const char *p, *end;
    bool ok = true;
    std::map<std::string, std::string> leaves; // path -> literal text ("true", "0.77", ...)

    void ws()
    {
        for (;;) {
            while (p < end && is_space(*p)) ++p;
            if (p + 1 < end && p[0] == '/' && p[1] == '/') { while (p < end && *p != '\n') ++p; continue; }
            if (p + 1 < end && p[0] == '/' && p[1] == '*') {
                p += 2;
                while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p;
                p = p + 2 <= end ? p + 2 : end;
                continue;
            }
            return;
        }
    }
    bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    std::string str()
    {
        std::string s;
        ws();
        if (p >= end || *p != '"') { ok = false; return s; }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) ++p;
            s.push_back(*p++);
        }
        if (p >= end) ok = false; else ++p;
        return s;
    }
    void value(const std::string &path, int depth)
    {
        ws();
        if (p >= end || depth > 16) { ok = false; return; }
        if (*p == '{') {
            ++p;
            if (lit('}')) return;
            do {
                std::string k = str();
                if (!ok || !lit(':')) { ok = false; return; }
                value(path.empty() ? k : path + "." + k, depth + 1);
                if (!ok) return;
            } while (lit(','));
            if (!lit('}')) ok = false;
        } else if (*p == '[') {
            ++p;
            if (lit(']')) return;
            int i = 0;
            do { value(path + "[" + std::to_string(i++) + "]", depth + 1); if (!ok) return; } while (lit(','));
            if (!lit(']')) ok = false;
        } else if (*p == '"') {
            leaves[path] = str();
        } else {
            const char *s = p;
            while (p < end && (is_alnum(*p) || *p == '+' || *p == '-' || *p == '.')) ++p;
            if (p == s) { ok = false; return; }
            leaves[path] = std::string(s, p);
        }
    }
};

bool as_bool(const std::map<std::string, std::string> &m, const char *k, bool def)
{
    auto it = m.find(k);
    if (it == m.end()) return def;
    if (it->second == "true") return true;
    if (it->second == "false" || it->second == "null") return false;
    return c_atof(it->second.c_str()) != 0.0; // jsoncpp asBool(): non-zero numbers are true
}
float as_float(const std::map<std::string, std::string> &m, const char *k, double def)
{
    auto it = m.find(k);
    if (it == m.end()) return (float)def;
    if (it->second == "true") return 1.0f;
    if (it->second == "false" || it->second == "null") return 0.0f;
    return (float)c_atof(it->second.c_str()); // asFloat(): double -> float
}

} // namespace

extern "C" {

static int config_from_json_impl(const char *text, size_t len, ovrfsr_config *cfg)
{
    Parser ps{text, text + len};
    ps.value("", 0);
    ps.ws();
    if (!ps.ok || ps.p != ps.end) return OVRFSR_ERR_INVALID_ARGUMENT; // struct defaults stay, like the reference's catch(...)
    const auto &m = ps.leaves;
    cfg->fsr_enabled = as_bool(m, "fsr.enabled", false);
    cfg->sharpness = as_float(m, "fsr.sharpness", 1.0);
    if (cfg->sharpness < 0) cfg->sharpness = 0;
    cfg->render_scale = as_float(m, "fsr.renderScale", 1.0);
    cfg->radius = as_float(m, "fsr.radius", 0.5);
    if (cfg->radius < 0) cfg->radius = 0; // the rule the reference applies to sharpness (Config.h:40), applied to the other field a negative
                                          // value is undefined for (float -> uint32, PostProcessor.cpp:303): a file never yields a cfg create refuses
    cfg->debug_mode = as_bool(m, "fsr.debugMode", false);
    cfg->use_nis = as_bool(m, "fsr.useNIS", false);
    // a literal that overflows float (1e999) or is not a number: "Could not read config file" -- struct defaults, like a parse error
    if (!std::isfinite(cfg->sharpness) || !std::isfinite(cfg->radius) || !std::isfinite(cfg->render_scale)) {
        ovrfsr_config_default(cfg);
        return OVRFSR_ERR_INVALID_ARGUMENT;
    }
    return OVRFSR_OK;
}

OVRFSR_API int ovrfsr_config_from_json(const char *text, size_t len, ovrfsr_config *cfg)
{
    if (!cfg || (!text && len)) return OVRFSR_ERR_INVALID_ARGUMENT;
    ovrfsr_config_default(cfg);
    try { // std::map / std::string may throw; nothing unwinds through the C boundary
        return config_from_json_impl(text, len, cfg);
    } catch (...) {
        ovrfsr_config_default(cfg);
        return OVRFSR_ERR_OUT_OF_MEMORY;
    }
}

// IEEE binary16 -> float, exact (plain integer code: this translation unit also builds with compilers that have no _Float16 in C++)
static float half_bits_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int s = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++s; } u = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3ffu) << 13); } // subnormal
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &u, 4);
    return f;
}

// What the capture writers accept: the rules ovrfsr_apply puts on caller images (PostProcessor::CheckImage: size, pitch, alignment).
// A pitch below width * texel would make the row loops read past the host copy (round 6).
static bool capture_image_ok(const ovrfsr_image *img, uint32_t tb)
{
    return img->width != 0 && img->height != 0 && img->width <= 16384u && img->height <= 16384u && img->pitch_bytes >= img->width * tb &&
           img->pitch_bytes % tb == 0 && (uint64_t)img->pitch_bytes * img->height <= 0x100000000ull && reinterpret_cast<uintptr_t>(img->data) % tb == 0;
}
// bytes of the image as it sits in memory: the last row ends with its last texel, not with its pitch (a view into a larger
// allocation owns nothing behind that)
static size_t capture_span(const ovrfsr_image *img, uint32_t tb) { return (size_t)(img->height - 1) * img->pitch_bytes + (size_t)img->width * tb; }

// Binary PPM (P6) of a device image; RGBA16F/32F/RGB10A2 are converted like a UNORM8 store.  Synchronises `stream`.
static int save_ppm_impl(const ovrfsr_image *img, const char *path, void *stream);
OVRFSR_API int ovrfsr_save_ppm(const ovrfsr_image *img, const char *path, void *stream)
{
    try {
        return save_ppm_impl(img, path, stream);
    } catch (...) {
        return OVRFSR_ERR_OUT_OF_MEMORY;
    }
}
static int save_ppm_impl(const ovrfsr_image *img, const char *path, void *stream)
{
    if (!img || !img->data || !path || img->format > OVRFSR_FORMAT_BGRA8_UNORM) return OVRFSR_ERR_INVALID_ARGUMENT;
    const bool ten = img->format == OVRFSR_FORMAT_RGB10A2_UNORM;
    const bool bgra = img->format == OVRFSR_FORMAT_BGRA8_UNORM;
    const size_t tb = img->format == OVRFSR_FORMAT_RGBA8_UNORM || ten || bgra ? 4 : img->format == OVRFSR_FORMAT_RGBA16F ? 8 : 16;
    if (!capture_image_ok(img, (uint32_t)tb)) return OVRFSR_ERR_INVALID_ARGUMENT;
    std::vector<unsigned char> host(capture_span(img, (uint32_t)tb));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(host.data(), img->data, host.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return OVRFSR_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return OVRFSR_ERR_HIP;
    FILE *f = std::fopen(path, "wb");
    if (!f) return OVRFSR_ERR_INVALID_ARGUMENT;
    std::fprintf(f, "P6\n%u %u\n255\n", img->width, img->height);
    std::vector<unsigned char> row((size_t)img->width * 3);
    auto q = [](float x) { x = x < 0 ? 0 : (x > 1 ? 1 : x); return (unsigned char)std::floor(x * 255.0f + 0.5f); };
    for (uint32_t y = 0; y < img->height; ++y) {
        const unsigned char *src = host.data() + (size_t)y * img->pitch_bytes;
        for (uint32_t x = 0; x < img->width; ++x)
            for (int c = 0; c < 3; ++c) {
                if (ten) { uint32_t v; std::memcpy(&v, src + x * 4, 4); row[x * 3 + c] = q((float)((v >> (10 * c)) & 1023u) / 1023.0f); }
                else if (tb == 4) row[x * 3 + c] = src[x * 4 + (bgra ? 2 - c : c)];
                else if (tb == 16) { float v; std::memcpy(&v, src + x * 16 + c * 4, 4); row[x * 3 + c] = q(v); }
                else { uint16_t h; std::memcpy(&h, src + x * 8 + c * 2, 2); row[x * 3 + c] = q(half_bits_to_float(h)); }
            }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return OVRFSR_OK;
}

// The F7 capture itself writes a DDS (SaveDDSTextureToFile, PostProcessor.cpp:640-657 / ScreenGrab11.h): a DDS file with the DX10 header
// extension, one mip, the texels exactly as they sit in the image (no conversion: what a capture is for) -- "DDS " + DDS_HEADER (124 bytes:
// caps, height, width, pitch, pixel format FourCC 'DX10') + DDS_HEADER_DXT10 (DXGI format, TEXTURE2D, array size 1) + tightly packed rows.
static int save_dds_impl(const ovrfsr_image *img, const char *path, void *stream)
{
    if (!img || !img->data || !path || img->format > OVRFSR_FORMAT_BGRA8_UNORM || img->width == 0 || img->height == 0) return OVRFSR_ERR_INVALID_ARGUMENT;
    // DXGI_FORMAT_R8G8B8A8_UNORM = 28, R16G16B16A16_FLOAT = 10, R32G32B32A32_FLOAT = 2, R10G10B10A2_UNORM = 24, B8G8R8A8_UNORM = 87
    static const uint32_t dxgi[5] = {28u, 10u, 2u, 24u, 87u}, bytes[5] = {4u, 8u, 16u, 4u, 4u};
    const uint32_t tb = bytes[img->format];
    if (!capture_image_ok(img, tb)) return OVRFSR_ERR_INVALID_ARGUMENT;
    const uint32_t rowBytes = img->width * tb;
    std::vector<unsigned char> host(capture_span(img, tb));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(host.data(), img->data, host.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return OVRFSR_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return OVRFSR_ERR_HIP;
    uint32_t h[32] = {0};                       // "DDS " + 31 dwords of DDS_HEADER
    h[0] = 0x20534444u;                         // magic
    h[1] = 124u;                                // dwSize
    h[2] = 0x1u | 0x2u | 0x4u | 0x8u | 0x1000u; // DDSD_CAPS | HEIGHT | WIDTH | PITCH | PIXELFORMAT
    h[3] = img->height; h[4] = img->width; h[5] = rowBytes;
    h[7] = 1u;                                  // dwMipMapCount
    h[19] = 32u;                                // DDS_PIXELFORMAT.dwSize
    h[20] = 0x4u;                               // DDPF_FOURCC
    h[21] = 0x30315844u;                        // 'DX10'
    h[27] = 0x1000u;                            // DDSCAPS_TEXTURE
    const uint32_t dx10[5] = {dxgi[img->format], 3u /* D3D10_RESOURCE_DIMENSION_TEXTURE2D */, 0u, 1u /* arraySize */, 0u};
    FILE *f = std::fopen(path, "wb");
    if (!f) return OVRFSR_ERR_INVALID_ARGUMENT;
    bool ok = std::fwrite(h, 4, 32, f) == 32 && std::fwrite(dx10, 4, 5, f) == 5;
    for (uint32_t y = 0; ok && y < img->height; ++y) ok = std::fwrite(host.data() + (size_t)y * img->pitch_bytes, 1, rowBytes, f) == rowBytes;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? OVRFSR_OK : OVRFSR_ERR_INVALID_ARGUMENT;
}
OVRFSR_API int ovrfsr_save_dds(const ovrfsr_image *img, const char *path, void *stream)
{
    try {
        return save_dds_impl(img, path, stream);
    } catch (...) {
        return OVRFSR_ERR_OUT_OF_MEMORY;
    }
}

} // extern "C"
