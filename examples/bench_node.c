/* examples/bench_node.c -- the multi-GPU throughput loop of bench.py with no Python in it: ONE process, one pthread per
 * device, each with its own ctx (ovrfsr_create(dev_i)), stream and resident sub-batch of stereo pairs; nothing is exchanged
 * between devices (the path shards embarrassingly, no collective).  The timed region is EXACTLY --steps calls
 * of ovrfsr_apply_batch per device, bracketed by a stream sync + a barrier over all threads on both sides; per-device time
 * from HIP events on the launch stream, whole-job value from the wall clock of the slowest thread (host max-join).
 *
 *   gcc -std=c11 -O2 -pthread -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ examples/bench_node.c -Iinclude -I/opt/rocm/include \
 *       -Lopenvr_fsr_amd -lopenvr_fsr_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../openvr_fsr_amd' \
 *       -Wl,-rpath,/opt/rocm/lib -o examples/bench_node          (__graft_entry__.build() does this)
 *   examples/bench_node [--gpus N] [--pairs P] [--steps K] [--warmup W] [--radius R] [--fused] [--oversubscribe]
 *
 * Workload: BASELINE C2's shape (1683x1869 -> 2244x2492 RGBA8, EASU -> UNORM8 -> RCAS, sharpness 0.9); the eye images are a
 * cheap deterministic pattern generated on the host once per device (gradients, a checker, a diagonal ramp, hashed noise),
 * uploaded before the timed region -- the kernels' cost does not depend on content.  --oversubscribe maps thread i to
 * device i % device_count (testing the N > 1 path on a box with fewer GPUs; not a scaling measurement).
 * Self-verifying per device: after the timed region every shard downloads image 0 of its own output batch and checksums it; the main
 * thread then has DEVICE 0 process every shard's image 0 (regenerated from the shard's seed) and compares the checksums.  A device
 * whose kernels, per-device LDS attribute or ctx state went wrong fails the run by name (exit status 1, "parity_check" in the line);
 * --corrupt-shard K (testing) flips one byte of shard K's downloaded image to show that it does.
 * Prints one JSON line.  bench.py remains the round driver's entry point; this is the C caller a node deployment would use.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "openvr_fsr_amd.h"

enum { IN_W = 1683, IN_H = 1869, OUT_W = 2244, OUT_H = 2492 };

typedef struct {
    int index, device, pairs, steps, warmup, fused, corrupt;
    float radius;
    uint64_t out0_sum;     /* FNV-1a of image 0 of the output batch the timed calls wrote */
    pthread_barrier_t *gate;
    double t_start, t_end; /* host seconds around the timed region */
    float device_ms;       /* HIP events on the launch stream */
    int status;
    char error[256];
} shard_t;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

static void synth_eye(uint8_t *p, uint32_t seed)
{
    const float ph = (float)(seed & 1023u) * 0.006f;
    for (uint32_t y = 0; y < IN_H; ++y)
        for (uint32_t x = 0; x < IN_W; ++x, p += 4) {
            const uint32_t n = hash32(seed * 0x9e3779b9u + y * IN_W + x);
            const int noise = (int)(n % 9u) - 4;
            int r = (int)(127.5f + 89.0f * sinf(x * 0.011f + ph) * cosf(y * 0.008f - ph)) + noise;
            int g = (((x + y) / 311u) & 1u) ? 200 : 60;
            int b = (int)((x * 3u + y * 5u) >> 3) & 255;
            g += (int)((n >> 8) % 9u) - 4;
            p[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            p[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
            p[2] = (uint8_t)b;
            p[3] = 255;
        }
}

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

static uint32_t shard_seed0(int pairs, int index) { return 0x5EED0000u + 2u * (uint32_t)pairs * (uint32_t)index; }

#define FAIL(s, ...) do { snprintf((s)->error, sizeof (s)->error, __VA_ARGS__); (s)->status = 1; } while (0)

static void *run_shard(void *arg)
{
    shard_t *s = (shard_t *)arg;
    const size_t in_bytes = (size_t)IN_W * IN_H * 4, out_bytes = (size_t)OUT_W * OUT_H * 4;
    const uint32_t n_img = 2u * (uint32_t)s->pairs;
    void *d_in = NULL, *d_out = NULL;
    hipStream_t stream = NULL;
    hipEvent_t ev0 = NULL, ev1 = NULL;
    ovrfsr_ctx *ctx = NULL;
    uint8_t *h = NULL;
    s->status = 0;
    if (hipSetDevice(s->device) != hipSuccess) FAIL(s, "hipSetDevice(%d)", s->device);
    if (!s->status && (hipStreamCreate(&stream) != hipSuccess || hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess))
        FAIL(s, "stream / event creation on device %d", s->device);
    if (!s->status && (hipMalloc(&d_in, in_bytes * n_img) != hipSuccess || hipMalloc(&d_out, out_bytes * n_img) != hipSuccess))
        FAIL(s, "hipMalloc of %u eye images on device %d", n_img, s->device);
    if (!s->status) {
        /* the sub-batch of this device: global pairs [index * pairs, (index + 1) * pairs), seed 0x5EED0000 + 2 * pair + eye */
        h = (uint8_t *)malloc(in_bytes);
        for (uint32_t i = 0; i < n_img && h && !s->status; ++i) {
            if (i < 4) synth_eye(h, shard_seed0(s->pairs, s->index) + i); /* four distinct images, then copies */
            const hipError_t e = i < 4 ? hipMemcpy((uint8_t *)d_in + in_bytes * i, h, in_bytes, hipMemcpyHostToDevice)
                                       : hipMemcpy((uint8_t *)d_in + in_bytes * i, (uint8_t *)d_in + in_bytes * (i & 3u), in_bytes, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) FAIL(s, "upload of image %u", i);
        }
        if (!h) FAIL(s, "host allocation");
    }
    if (!s->status) {
        ovrfsr_config cfg;
        ovrfsr_config_default(&cfg);
        cfg.fsr_enabled = 1; cfg.sharpness = 0.9f; cfg.radius = s->radius; cfg.out_width = OUT_W; cfg.out_height = OUT_H;
        if (s->fused) cfg.fused = 1; /* one launch, intermediate in LDS: its raised dynamic-LDS attribute is per device */
        const int rc = ovrfsr_create(s->device, &cfg, &ctx);
        if (rc != OVRFSR_OK) FAIL(s, "ovrfsr_create(%d): status %d", s->device, rc);
    }
    const ovrfsr_image in0 = { d_in, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
    const ovrfsr_image out0 = { d_out, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
#define STEP() ovrfsr_apply_batch(ctx, n_img, OVRFSR_EYE_LEFT, 1, &in0, in_bytes, &out0, out_bytes, stream)
    for (int i = 0; i < s->warmup && !s->status; ++i)
        if (STEP() != OVRFSR_OK) FAIL(s, "apply_batch (warm-up): %s", ovrfsr_last_error(ctx));
    if (stream) (void)hipStreamSynchronize(stream);
    pthread_barrier_wait(s->gate); /* every thread reaches both barriers, failed or not */
    s->t_start = now_s();
    if (!s->status) {
        (void)hipEventRecord(ev0, stream);
        for (int i = 0; i < s->steps && !s->status; ++i)
            if (STEP() != OVRFSR_OK) FAIL(s, "apply_batch: %s", ovrfsr_last_error(ctx));
        (void)hipEventRecord(ev1, stream);
        (void)hipStreamSynchronize(stream);
    }
    pthread_barrier_wait(s->gate);
    s->t_end = now_s();
    if (!s->status && hipEventElapsedTime(&s->device_ms, ev0, ev1) != hipSuccess) FAIL(s, "hipEventElapsedTime");
    if (!s->status) { /* image 0 of what the timed calls wrote on THIS device */
        uint8_t *o = (uint8_t *)malloc(out_bytes);
        if (!o || hipMemcpy(o, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) FAIL(s, "download of output image 0");
        else {
            if (s->corrupt) o[out_bytes / 2] ^= 0x10; /* --corrupt-shard: prove the comparison below notices */
            s->out0_sum = fnv1a(o, out_bytes);
        }
        free(o);
    }
    if (ctx) ovrfsr_destroy(ctx);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
    free(h);
    return NULL;
}

int main(int argc, char **argv)
{
    int gpus = 1, pairs = 64, steps = 20, warmup = 5, oversubscribe = 0, fused = 0, corrupt_shard = -1;
    float radius = 2.0f;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--oversubscribe")) oversubscribe = 1;
        else if (!strcmp(argv[i], "--fused")) fused = 1;
        else if (i + 1 < argc && !strcmp(argv[i], "--gpus")) gpus = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--pairs")) pairs = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--steps")) steps = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--warmup")) warmup = atoi(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--radius")) radius = (float)atof(argv[++i]);
        else if (i + 1 < argc && !strcmp(argv[i], "--corrupt-shard")) corrupt_shard = atoi(argv[++i]);
        else { fprintf(stderr, "usage: %s [--gpus N] [--pairs P] [--steps K] [--warmup W] [--radius R] [--fused] [--oversubscribe] [--corrupt-shard K]\n", argv[0]); return 2; }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
    if (gpus < 1 || pairs < 1 || steps < 1 || warmup < 0 || gpus > 64) { fprintf(stderr, "bad arguments\n"); return 2; }
    if (gpus > count && !oversubscribe) { fprintf(stderr, "--gpus %d but %d device(s) visible (use --oversubscribe to test)\n", gpus, count); return 1; }

    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)gpus);
    shard_t *sh = (shard_t *)calloc((size_t)gpus, sizeof *sh);
    pthread_t *th = (pthread_t *)calloc((size_t)gpus, sizeof *th);
    for (int i = 0; i < gpus; ++i) {
        sh[i].index = i; sh[i].device = i % count; sh[i].pairs = pairs; sh[i].steps = steps; sh[i].warmup = warmup; sh[i].radius = radius; sh[i].fused = fused;
        sh[i].corrupt = i == corrupt_shard;
        sh[i].gate = &gate;
        pthread_create(&th[i], NULL, run_shard, &sh[i]);
    }
    double t0 = 1e300, t1 = 0.0;
    int failed = 0;
    for (int i = 0; i < gpus; ++i) {
        pthread_join(th[i], NULL);
        if (sh[i].status) { fprintf(stderr, "shard %d: %s\n", i, sh[i].error); failed = 1; }
        if (sh[i].t_start < t0) t0 = sh[i].t_start;
        if (sh[i].t_end > t1) t1 = sh[i].t_end;
    }
    if (failed) return 1;
    /* per-device parity: device 0 processes image 0 of every shard (same seed, same configuration) and must reproduce the bytes
     * that shard's own device wrote in the timed region */
    int *match = (int *)calloc((size_t)gpus, sizeof *match);
    int parity_ok = 1;
    {
        const size_t in_bytes = (size_t)IN_W * IN_H * 4, out_bytes = (size_t)OUT_W * OUT_H * 4;
        void *d_in = NULL, *d_out = NULL;
        uint8_t *h = (uint8_t *)malloc(in_bytes), *o = (uint8_t *)malloc(out_bytes);
        ovrfsr_ctx *ctx = NULL;
        ovrfsr_config cfg;
        ovrfsr_config_default(&cfg);
        cfg.fsr_enabled = 1; cfg.sharpness = 0.9f; cfg.radius = radius; cfg.out_width = OUT_W; cfg.out_height = OUT_H;
        if (fused) cfg.fused = 1;
        int bad = !h || !o || hipSetDevice(sh[0].device) != hipSuccess || hipMalloc(&d_in, in_bytes) != hipSuccess ||
                  hipMalloc(&d_out, out_bytes) != hipSuccess || ovrfsr_create(sh[0].device, &cfg, &ctx) != OVRFSR_OK;
        const ovrfsr_image in0 = { d_in, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        const ovrfsr_image out0 = { d_out, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        for (int i = 0; i < gpus && !bad; ++i) {
            synth_eye(h, shard_seed0(pairs, i));
            bad = hipMemcpy(d_in, h, in_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                  ovrfsr_apply_batch(ctx, 1, OVRFSR_EYE_LEFT, 1, &in0, in_bytes, &out0, out_bytes, NULL) != OVRFSR_OK ||
                  hipDeviceSynchronize() != hipSuccess || hipMemcpy(o, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess;
            if (!bad) match[i] = fnv1a(o, out_bytes) == sh[i].out0_sum;
            if (!bad && !match[i]) { fprintf(stderr, "shard %d (device %d): image 0 differs from device %d's result for the same seed\n", i, sh[i].device, sh[0].device); parity_ok = 0; }
        }
        if (bad) { fprintf(stderr, "parity check could not run (%s)\n", ctx ? ovrfsr_last_error(ctx) : "setup"); parity_ok = 0; }
        if (ctx) ovrfsr_destroy(ctx);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        free(h); free(o);
    }
    const double wall = t1 - t0;
    printf("{\"metric\": \"stereo eye-pairs/sec at 1683x1869->2244x2492 (EASU+RCAS)\", \"value\": %.2f, \"unit\": \"eye-pairs/s\", "
           "\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", "
           "\"data\": \"synthetic (host-generated pattern)\", \"config\": {\"workload\": \"C2 shape, radius %.1f\", \"pairs_per_gpu_per_step\": %d, "
           "\"launcher\": \"examples/bench_node: one process, one pthread + ctx + stream per device, no Python\"%s, \"per_device_ms_per_step\": [",
           (double)pairs * gpus * steps / wall, gpus, steps, warmup, wall / steps * 1e3, (double)radius, pairs,
           oversubscribe ? ", \"oversubscribed\": \"TEST RUN: shards share devices, not a scaling measurement\"" : "");
    for (int i = 0; i < gpus; ++i) printf("%s%.4f", i ? ", " : "", sh[i].device_ms / steps);
    printf("]}, \"parity_check\": {\"ok\": %s, \"method\": \"image 0 of every shard's timed output batch (FNV-1a of its 22.4 MB) against device %d processing the "
           "same seed\", \"per_shard\": [", parity_ok ? "true" : "false", sh[0].device);
    for (int i = 0; i < gpus; ++i)
        printf("%s{\"shard\": %d, \"device\": %d, \"checksum\": \"%016llx\", \"matches_device0\": %s}", i ? ", " : "", i, sh[i].device,
               (unsigned long long)sh[i].out0_sum, match[i] ? "true" : "false");
    printf("]}}\n");
    pthread_barrier_destroy(&gate);
    free(sh); free(th); free(match);
    return parity_ok ? 0 : 1;
}
