/* examples/headless.c -- the C ABI used from plain C, no Python anywhere: what a headless caller (or the reference-side
 * stub of INTEGRATION.md) does.  Generates one synthetic RGBA8 eye image on the host, uploads it, runs the configured
 * pipeline for both eyes through ovrfsr_apply, prints the GPU time of the last call and dumps the result as a PPM.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ examples/headless.c -Iinclude -I/opt/rocm/include -Lopenvr_fsr_amd -lopenvr_fsr_amd \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../openvr_fsr_amd' -Wl,-rpath,/opt/rocm/lib -o examples/headless
 *   (plain gcc: the only HIP the caller needs is hipMalloc/hipMemcpy for its own buffers; __graft_entry__.build() does this)
 *   examples/headless [openvr_mod.cfg | -] [out.ppm | out.dds] [--pair | --pair-rl]
 * A capture path ending in .dds is written with ovrfsr_save_dds (the reference's F7 container), anything else as a PPM.  --pair runs the same
 * frames through cfg.pair_submit: the apply of a frame's FIRST eye only records (ovrfsr_pair_pending() == 1: a Submit detour would hold that
 * eye's forwarded Submit back), the apply of the other eye launches both as one batch of two -- the first eye's ctx-owned image
 * is complete once the second call has returned, and the right eye must produce the same checksum as without pairing.  --pair-rl submits the
 * right eye first, as some games do (ABI 5: pairing is by arrival order).  Each eye has its own input texture (one texture submitted for both
 * eyes is never paired).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "openvr_fsr_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    const uint32_t inW = 1683, inH = 1869; /* BASELINE C2: one eye at renderScale 0.75 of 2244x2492 */
    ovrfsr_config cfg;
    ovrfsr_config_default(&cfg);
    cfg.fsr_enabled = 1; cfg.render_scale = 0.75f; cfg.sharpness = 0.9f; cfg.radius = 0.5f; cfg.debug_mode = 1;
    if (argc > 1 && strcmp(argv[1], "-") != 0) { /* the reference's own config file format (openvr_mod.cfg) */
        FILE *f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 1; }
        static char text[1 << 16];
        const size_t n = fread(text, 1, sizeof text, f);
        fclose(f);
        if (ovrfsr_config_from_json(text, n, &cfg) != OVRFSR_OK) { fprintf(stderr, "cannot parse %s\n", argv[1]); return 1; }
        cfg.debug_mode = 1; /* for ovrfsr_last_gpu_time_ms */
    }
    cfg.out_width = 2244; cfg.out_height = 2492;
    int pair = 0, right_first = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--pair")) pair = 1;
        if (!strcmp(argv[i], "--pair-rl")) pair = right_first = 1;
    }
    cfg.pair_submit = pair;

    uint8_t *h = (uint8_t *)malloc((size_t)inW * inH * 4);
    for (uint32_t y = 0; y < inH; ++y)
        for (uint32_t x = 0; x < inW; ++x) {
            uint8_t *p = h + ((size_t)y * inW + x) * 4;
            p[0] = (uint8_t)(127.5 + 127.5 * sin(x * 0.021) * cos(y * 0.017));
            p[1] = (uint8_t)(((x / 24 + y / 24) & 1) ? 220 : 40);
            p[2] = (uint8_t)((x + y) & 255);
            p[3] = 255;
        }
    void *d_in = NULL, *d_in_right = NULL; /* one texture per eye (the same content: the checksum below does not depend on the mode) */
    CHECK_HIP(hipMalloc(&d_in, (size_t)inW * inH * 4));
    CHECK_HIP(hipMemcpy(d_in, h, (size_t)inW * inH * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&d_in_right, (size_t)inW * inH * 4));
    CHECK_HIP(hipMemcpy(d_in_right, h, (size_t)inW * inH * 4, hipMemcpyHostToDevice));

    ovrfsr_ctx *ctx = NULL;
    int rc = ovrfsr_create(0, &cfg, &ctx);
    if (rc != OVRFSR_OK) { fprintf(stderr, "ovrfsr_create failed: %d\n", rc); return 1; }
    ovrfsr_image right_out = { NULL, 0, 0, 0, 0 };
    for (int rep = 0; rep < 3; ++rep)
        for (int k = 0; k < 2; ++k) {
            const int eye = right_first ? 1 - k : k;
            ovrfsr_image in = { eye ? d_in_right : d_in, inW, inH, inW * 4, OVRFSR_FORMAT_RGBA8_UNORM };
            ovrfsr_image out = { NULL, 0, 0, 0, 0 }; /* ctx-owned output, as the reference swaps Texture_t::handle */
            rc = ovrfsr_apply(ctx, eye, &in, NULL, &out, NULL);
            if (rc != OVRFSR_OK) { fprintf(stderr, "ovrfsr_apply: %d (%s)\n", rc, ovrfsr_last_error(ctx)); return 1; }
            if (eye == 1) right_out = out;
            if (ovrfsr_pair_pending(ctx)) { /* recorded only: nothing launched yet; a detour holds this eye's Submit until the next call returns */
                printf("eye %d: recorded (pair_submit), result will be at %ux%u\n", eye, out.width, out.height);
                continue;
            }
            float ms = 0.f;
            if (ovrfsr_last_gpu_time_ms(ctx, &ms) == OVRFSR_OK)
                printf("eye %d: %ux%u -> %ux%u  %s  %.3f ms on the GPU%s\n", eye, inW, inH, out.width, out.height,
                       cfg.use_nis ? "NIS" : "EASU+RCAS", ms, pair ? " (both eyes, one batch of two)" : "");
            if (rep == 2 && k == 1) { /* a checksum of the right eye's result: the same in every mode */
                out = right_out;
                const size_t nb = (size_t)out.pitch_bytes * out.height;
                uint8_t *o = (uint8_t *)malloc(nb);
                CHECK_HIP(hipMemcpy(o, out.data, nb, hipMemcpyDeviceToHost));
                uint64_t hsh = 0xcbf29ce484222325ull;
                for (size_t k = 0; k < nb; ++k) { hsh ^= o[k]; hsh *= 0x100000001b3ull; }
                free(o);
                printf("right eye checksum %016llx%s\n", (unsigned long long)hsh, pair ? " (pair_submit)" : "");
            }
            if (rep == 2 && k == 1 && argc > 2 && strncmp(argv[2], "--pair", 6) != 0) {
                const size_t L = strlen(argv[2]);
                const int dds = L > 4 && !strcmp(argv[2] + L - 4, ".dds");
                rc = dds ? ovrfsr_save_dds(&out, argv[2], NULL) : ovrfsr_save_ppm(&out, argv[2], NULL);
                printf("%s %s\n", rc == OVRFSR_OK ? "wrote" : "could not write", argv[2]);
            }
        }
    ovrfsr_destroy(ctx);
    CHECK_HIP(hipFree(d_in));
    CHECK_HIP(hipFree(d_in_right));
    free(h);
    return 0;
}
