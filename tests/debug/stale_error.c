/* tests/debug/stale_error.c -- a HIP error the CALLER left behind on its thread must not fail (or be blamed on) the next ovrfsr_apply.
 * HIP keeps the last error of a host thread until somebody reads it (hipGetLastError); the launch wrappers of the library read it after
 * every launch, so a host whose own failed call -- here an allocation no device can satisfy -- was handled without being cleared would see
 * its next frame refused with that stale code.  The library therefore clears the thread's error state on entry to a launch and reports only
 * what its own launch raised.  Exit status 0 when the apply after the stale error succeeds and produces the pixels of the apply before it.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ tests/debug/stale_error.c -Iinclude -I/opt/rocm/include -Lopenvr_fsr_amd -lopenvr_fsr_amd \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../openvr_fsr_amd' -Wl,-rpath,/opt/rocm/lib -o tests/debug/stale_error */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "openvr_fsr_amd.h"

enum { IN_W = 160, IN_H = 120, OUT_W = 213, OUT_H = 160 };

int main(void)
{
    const size_t in_bytes = (size_t)IN_W * IN_H * 4, out_bytes = (size_t)OUT_W * OUT_H * 4;
    void *d_in = NULL, *d_out = NULL, *huge = NULL;
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "stale_error: no device\n"); return 2; }
    uint8_t *h = (uint8_t *)malloc(in_bytes), *a = (uint8_t *)malloc(out_bytes), *b = (uint8_t *)malloc(out_bytes);
    for (size_t i = 0; i < in_bytes; ++i) h[i] = (uint8_t)((i * 2654435761u) >> 24);
    if (hipMalloc(&d_in, in_bytes) != hipSuccess || hipMalloc(&d_out, out_bytes) != hipSuccess ||
        hipMemcpy(d_in, h, in_bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "stale_error: set-up failed\n"); return 1; }
    ovrfsr_config cfg;
    ovrfsr_config_default(&cfg);
    cfg.fsr_enabled = 1; cfg.radius = 0.6f; cfg.sharpness = 0.9f; cfg.out_width = OUT_W; cfg.out_height = OUT_H;
    ovrfsr_ctx *ctx = NULL;
    if (ovrfsr_create(0, &cfg, &ctx) != OVRFSR_OK) { fprintf(stderr, "stale_error: ovrfsr_create failed\n"); return 1; }
    const ovrfsr_image in = { d_in, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
    ovrfsr_image out = { d_out, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
    int st = ovrfsr_apply(ctx, OVRFSR_EYE_LEFT, &in, NULL, &out, NULL);
    if (st != OVRFSR_OK || hipDeviceSynchronize() != hipSuccess || hipMemcpy(a, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        fprintf(stderr, "stale_error: first apply: status %d: %s\n", st, ovrfsr_last_error(ctx)); return 1;
    }
    (void)hipMemset(d_out, 0, out_bytes);
    const hipError_t e = hipMalloc(&huge, (size_t)1 << 60); /* the caller's own failure, handled by looking at the return value only */
    if (e == hipSuccess) { fprintf(stderr, "stale_error: a 2^60-byte allocation succeeded?\n"); return 2; }
    st = ovrfsr_apply(ctx, OVRFSR_EYE_LEFT, &in, NULL, &out, NULL);
    if (st != OVRFSR_OK) { fprintf(stderr, "stale_error: apply after the caller's failed hipMalloc (%s): status %d: %s\n", hipGetErrorName(e), st, ovrfsr_last_error(ctx)); return 1; }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(b, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "stale_error: download\n"); return 1; }
    if (memcmp(a, b, out_bytes) != 0) { fprintf(stderr, "stale_error: pixels differ after the stale error\n"); return 1; }
    /* and the other direction: a reset + rebuild (allocations, table uploads, launches) with a stale error pending */
    (void)hipMalloc(&huge, (size_t)1 << 60);
    if (ovrfsr_reset(ctx) != OVRFSR_OK || (st = ovrfsr_apply(ctx, OVRFSR_EYE_LEFT, &in, NULL, &out, NULL)) != OVRFSR_OK) {
        fprintf(stderr, "stale_error: rebuild after a stale error: status %d: %s\n", st, ovrfsr_last_error(ctx)); return 1;
    }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(b, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess || memcmp(a, b, out_bytes) != 0) {
        fprintf(stderr, "stale_error: pixels differ after the rebuild\n"); return 1;
    }
    ovrfsr_destroy(ctx);
    printf("stale_error: a pending %s of the caller's neither failed nor changed the next apply / rebuild\n", hipGetErrorName(e));
    return 0;
}
