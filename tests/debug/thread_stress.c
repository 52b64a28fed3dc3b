/* tests/debug/thread_stress.c -- "distinct ctxs are independent" (include/openvr_fsr_amd.h, ovrfsr_ctx) as a test: T host threads on ONE device,
 * each creating, driving and destroying its own ctxs through every launch form of the library at the same time, every result compared with the
 * checksum the same job produced when the main thread ran it alone.  The reference has a single render thread and no race detection
 * ("Race detection / sanitizers: none"); this driver is what the ThreadSanitizer build of the host translation units
 * (tools/build_tsan.sh -> ab/tsan.so, ab/thread_stress_tsan) runs, and the product library runs it too (tests/test_gpu_threads.py).
 *
 *   gcc -std=c11 -O2 -pthread -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ tests/debug/thread_stress.c -Iinclude -I/opt/rocm/include \
 *       -Lopenvr_fsr_amd -lopenvr_fsr_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../../openvr_fsr_amd' -Wl,-rpath,/opt/rocm/lib \
 *       -o tests/debug/thread_stress                                                             (__graft_entry__.build() does this)
 *   tests/debug/thread_stress [--threads T] [--rounds R]
 *   tests/debug/thread_stress --misuse      the POSITIVE CONTROL of the race detector: two threads drive ONE ctx at the same time, which the
 *                                           header forbids ("not thread-safe").  Nothing is checked; under ThreadSanitizer this run must
 *                                           produce reports with frames in ovrfsr::PostProcessor, or the clean run above proves nothing.
 *
 * Jobs (each: its own ctx, stream, device buffers; 320x270 -> 427x360 unless stated):
 *   0 two kernels, no mask            4 pair_submit, eyes R then L, ctx-owned outputs      8 strict build of job 1
 *   1 radius 0.5 (sorted tiles + the concurrent outside kernel on the ctx's auxiliary stream)
 *   2 NVScaler, radius 0.45           5 debug_mode (timing ring, average read back)
 *   3 fused kernel, radius 0.5        6 input size changes twice (rebuild between launches)
 *                                     7 set_config between launches (hotkey path) + reset
 * Every thread first drives a ctx that the MAIN thread created for it (job 1's configuration) and the main thread destroys after the join.
 * Host-only entry points (config_from_json, mask_constants, nis_scaler_config, easu_con) are called from every thread between jobs.
 * Exit status 0 and one line "thread_stress: T threads x R rounds x 9 jobs: all checksums equal the serial run" on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "openvr_fsr_amd.h"

enum { IN_W = 320, IN_H = 270, OUT_W = 427, OUT_H = 360, IN2_W = 256, IN2_H = 200, N_JOBS = 9, ITERS = 6 };

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

static void synth(uint8_t *p, uint32_t w, uint32_t h, uint32_t seed)
{
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x, p += 4) {
            const uint32_t n = hash32(seed * 0x9e3779b9u + y * w + x);
            int r = (int)(127.5f + 100.0f * sinf(x * 0.05f + (float)seed) * cosf(y * 0.04f)) + (int)(n % 7u) - 3;
            p[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            p[1] = (uint8_t)((((x + 2u * y) / 23u) & 1u) ? 210 : 40);
            p[2] = (uint8_t)(n >> 24);
            p[3] = 255;
        }
}

static uint64_t fnv1a(uint64_t h, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

typedef struct { char text[256]; } err_t;
#define CHECK(cond, ...) do { if (!(cond)) { snprintf(err->text, sizeof err->text, __VA_ARGS__); goto done; } } while (0)

static ovrfsr_config job_config(int job)
{
    ovrfsr_config cfg;
    ovrfsr_config_default(&cfg);
    cfg.fsr_enabled = 1; cfg.sharpness = 0.9f; cfg.radius = 0.5f; cfg.out_width = OUT_W; cfg.out_height = OUT_H;
    cfg.proj_centre[0] = 0.55f; cfg.proj_centre[2] = 0.45f;
    switch (job) {
    case 0: cfg.radius = 2.0f; break;
    case 2: cfg.use_nis = 1; cfg.radius = 0.45f; break;
    case 3: cfg.fused = 1; break;
    case 4: cfg.pair_submit = 1; break;
    case 5: cfg.debug_mode = 1; break;
    case 8: cfg.precision = OVRFSR_PRECISION_FP32_STRICT; break;
    default: break;
    }
    return cfg;
}

/* one job, start to finish; returns 0 and the checksum of everything it downloaded, or -1 and err->text */
static int run_job(int job, uint32_t seed, uint64_t *sum, err_t *err, ovrfsr_ctx *given)
{
    int rc = -1;
    const size_t in_bytes = (size_t)IN_W * IN_H * 4, in2_bytes = (size_t)IN2_W * IN2_H * 4, out_bytes = (size_t)OUT_W * OUT_H * 4;
    void *d_in = NULL, *d_in2 = NULL, *d_out = NULL;
    hipStream_t stream = NULL;
    ovrfsr_ctx *ctx = given; /* a ctx made by ANOTHER thread (how a VR host does it: created at start-up, driven from the render thread) */
    uint8_t *h = (uint8_t *)malloc(2 * in_bytes > 2 * out_bytes ? 2 * in_bytes : 2 * out_bytes);
    uint64_t acc = 0xcbf29ce484222325ull;
    CHECK(h, "host allocation");
    CHECK(hipStreamCreate(&stream) == hipSuccess, "hipStreamCreate");
    CHECK(hipMalloc(&d_in, 2 * in_bytes) == hipSuccess && hipMalloc(&d_in2, 2 * in2_bytes) == hipSuccess && hipMalloc(&d_out, 2 * out_bytes) == hipSuccess, "hipMalloc");
    synth(h, IN_W, IN_H, seed); synth(h + in_bytes, IN_W, IN_H, seed + 1);
    CHECK(hipMemcpy(d_in, h, 2 * in_bytes, hipMemcpyHostToDevice) == hipSuccess, "upload");
    synth(h, IN2_W, IN2_H, seed + 2); synth(h + in2_bytes, IN2_W, IN2_H, seed + 3);
    CHECK(hipMemcpy(d_in2, h, 2 * in2_bytes, hipMemcpyHostToDevice) == hipSuccess, "upload");
    CHECK(hipMemset(d_out, 0, 2 * out_bytes) == hipSuccess, "memset");

    ovrfsr_config cfg = job_config(job);
    if (!given) CHECK(ovrfsr_create(0, &cfg, &ctx) == OVRFSR_OK, "ovrfsr_create (job %d)", job);

    const ovrfsr_bounds one_eye = { 0.f, 0.f, 1.f, 1.f };
    for (int it = 0; it < ITERS; ++it) {
        const int small = job == 6 && (it == 2 || it == 3); /* job 6: A A B B A A -- two rebuilds */
        if (job == 7 && it == 2) { cfg.sharpness = 0.4f; CHECK(ovrfsr_set_config(ctx, &cfg) == OVRFSR_OK, "set_config"); }
        if (job == 7 && it == 4) CHECK(ovrfsr_reset(ctx) == OVRFSR_OK, "reset");
        ovrfsr_image outs[2];
        for (int e = 0; e < 2; ++e) {
            const int eye = job == 4 ? 1 - e : e; /* pair_submit pairs by arrival order: this host submits R,L */
            const ovrfsr_image in = small ? (ovrfsr_image){ (uint8_t *)d_in2 + in2_bytes * eye, IN2_W, IN2_H, IN2_W * 4, OVRFSR_FORMAT_RGBA8_UNORM }
                                          : (ovrfsr_image){ (uint8_t *)d_in + in_bytes * eye, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
            ovrfsr_image out = { job == 4 ? NULL : (uint8_t *)d_out + out_bytes * eye, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
            const int st = ovrfsr_apply(ctx, eye, &in, &one_eye, &out, stream);
            CHECK(st == OVRFSR_OK, "ovrfsr_apply (job %d, iteration %d, eye %d): status %d: %s", job, it, eye, st, ovrfsr_last_error(ctx));
            if (job == 4) CHECK(ovrfsr_pair_pending(ctx) == (e == 0), "pair_pending after eye %d of the frame", e);
            outs[eye] = out;
        }
        CHECK(hipStreamSynchronize(stream) == hipSuccess, "hipStreamSynchronize");
        for (int eye = 0; eye < 2; ++eye) {
            CHECK(outs[eye].data && outs[eye].width == OUT_W && outs[eye].height == OUT_H, "output descriptor");
            CHECK(hipMemcpy2D(h, OUT_W * 4, outs[eye].data, outs[eye].pitch_bytes, OUT_W * 4, OUT_H, hipMemcpyDeviceToHost) == hipSuccess, "download");
            acc = fnv1a(acc, h, out_bytes);
        }
    }
    if (job == 5) {
        float ms = -1.f;
        CHECK(ovrfsr_last_gpu_time_ms(ctx, &ms) == OVRFSR_OK && ms > 0.f, "last_gpu_time_ms");
    }
    *sum = acc;
    rc = 0;
done:
    if (ctx && !given) ovrfsr_destroy(ctx);
    if (d_in) (void)hipFree(d_in);
    if (d_in2) (void)hipFree(d_in2);
    if (d_out) (void)hipFree(d_out);
    if (stream) (void)hipStreamDestroy(stream);
    free(h);
    return rc;
}

/* the entry points that need no device: same answers from every thread */
static uint64_t host_only(void)
{
    uint64_t acc = 0xcbf29ce484222325ull;
    static const char text[] = "{ \"fsr\": { \"enabled\": true, \"sharpness\": 0.8, \"radius\": 0.6, \"renderScale\": 0.75, \"useNIS\": false } }";
    ovrfsr_config c;
    (void)ovrfsr_config_from_json(text, sizeof text - 1, &c);
    acc = fnv1a(acc, (const uint8_t *)&c.sharpness, 3 * sizeof(float));
    uint32_t centre[4], radius[4], con[16];
    const float proj[4] = { 0.55f, 0.5f, 0.45f, 0.5f };
    ovrfsr_mask_constants(centre, radius, OUT_W, OUT_H, proj, 0.5f, 1, 1);
    acc = fnv1a(acc, (const uint8_t *)centre, sizeof centre);
    acc = fnv1a(acc, (const uint8_t *)radius, sizeof radius);
    ovrfsr_easu_con(con, IN_W, IN_H, IN_W, IN_H, OUT_W, OUT_H);
    acc = fnv1a(acc, (const uint8_t *)con, sizeof con);
    uint8_t nis[256];
    memset(nis, 0, sizeof nis);
    (void)ovrfsr_nis_scaler_config(nis, 0.9f, IN_W, IN_H, OUT_W, OUT_H);
    acc = fnv1a(acc, nis, sizeof nis);
    return acc;
}

typedef struct { int index, rounds, status; const uint64_t *ref; uint64_t host_ref; pthread_barrier_t *gate; err_t err; ovrfsr_ctx *handed; } worker_t;

static void *worker(void *arg)
{
    worker_t *w = (worker_t *)arg;
    w->status = 0;
    if (hipSetDevice(0) != hipSuccess) { snprintf(w->err.text, sizeof w->err.text, "hipSetDevice"); w->status = 1; }
    pthread_barrier_wait(w->gate); /* everybody starts creating ctxs at once */
    if (!w->status && w->handed) { /* first the ctx the main thread created and handed over (job 1's configuration) */
        uint64_t sum = 0;
        if (run_job(1, 110u, &sum, &w->err, w->handed) != 0) w->status = 1;
        else if (sum != w->ref[1]) { snprintf(w->err.text, sizeof w->err.text, "the ctx created by the main thread gave checksum %016llx, the serial run %016llx",
                                               (unsigned long long)sum, (unsigned long long)w->ref[1]); w->status = 1; }
    }
    for (int r = 0; r < w->rounds && !w->status; ++r)
        for (int k = 0; k < N_JOBS && !w->status; ++k) {
            const int job = (k + 2 * w->index + r) % N_JOBS; /* neighbours run different jobs at any moment */
            uint64_t sum = 0;
            if (run_job(job, 100u + 10u * (uint32_t)job, &sum, &w->err, NULL) != 0) w->status = 1;
            else if (sum != w->ref[job]) {
                snprintf(w->err.text, sizeof w->err.text, "job %d, round %d: checksum %016llx, the serial run gave %016llx", job, r,
                         (unsigned long long)sum, (unsigned long long)w->ref[job]);
                w->status = 1;
            }
            if (!w->status && host_only() != w->host_ref) { snprintf(w->err.text, sizeof w->err.text, "host-only entry points disagree with the serial run"); w->status = 1; }
        }
    return NULL;
}

/* --misuse: one ctx, two threads (see the header of this file).  The ctx is warmed up by the main thread first so that the concurrent calls
 * allocate nothing: the races left are on the ctx's scalar state, enough for the detector and harmless to the process. */
typedef struct { ovrfsr_ctx *ctx; int eye; void *d_in, *d_out; pthread_barrier_t *gate; } misuse_t;

static void *misuse_worker(void *arg)
{
    misuse_t *m = (misuse_t *)arg;
    hipStream_t stream = NULL;
    if (hipSetDevice(0) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) return NULL;
    const ovrfsr_bounds one_eye = { 0.f, 0.f, 1.f, 1.f };
    pthread_barrier_wait(m->gate);
    for (int i = 0; i < 50; ++i) {
        const ovrfsr_image in = { m->d_in, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        ovrfsr_image out = { m->d_out, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        (void)ovrfsr_apply(m->ctx, m->eye, &in, &one_eye, &out, stream);
    }
    (void)hipStreamSynchronize(stream);
    (void)hipStreamDestroy(stream);
    return NULL;
}

static int misuse(void)
{
    const size_t in_bytes = (size_t)IN_W * IN_H * 4, out_bytes = (size_t)OUT_W * OUT_H * 4;
    void *d_in = NULL, *d_out = NULL;
    ovrfsr_ctx *ctx = NULL;
    ovrfsr_config cfg;
    ovrfsr_config_default(&cfg);
    cfg.fsr_enabled = 1; cfg.radius = 2.0f; cfg.out_width = OUT_W; cfg.out_height = OUT_H;
    if (hipMalloc(&d_in, 2 * in_bytes) != hipSuccess || hipMalloc(&d_out, 2 * out_bytes) != hipSuccess || hipMemset(d_in, 0x55, 2 * in_bytes) != hipSuccess ||
        ovrfsr_create(0, &cfg, &ctx) != OVRFSR_OK) { fprintf(stderr, "thread_stress: --misuse set-up failed\n"); return 1; }
    const ovrfsr_bounds one_eye = { 0.f, 0.f, 1.f, 1.f };
    for (int eye = 0; eye < 2; ++eye) { /* warm-up, serial */
        const ovrfsr_image in = { (uint8_t *)d_in + in_bytes * eye, IN_W, IN_H, IN_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        ovrfsr_image out = { (uint8_t *)d_out + out_bytes * eye, OUT_W, OUT_H, OUT_W * 4, OVRFSR_FORMAT_RGBA8_UNORM };
        if (ovrfsr_apply(ctx, eye, &in, &one_eye, &out, NULL) != OVRFSR_OK) { fprintf(stderr, "thread_stress: --misuse warm-up failed\n"); return 1; }
    }
    (void)hipDeviceSynchronize();
    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, 2);
    misuse_t m[2];
    pthread_t th[2];
    for (int t = 0; t < 2; ++t) {
        m[t] = (misuse_t){ ctx, t, (uint8_t *)d_in + in_bytes * t, (uint8_t *)d_out + out_bytes * t, &gate };
        pthread_create(&th[t], NULL, misuse_worker, &m[t]);
    }
    for (int t = 0; t < 2; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&gate);
    (void)hipDeviceSynchronize();
    ovrfsr_destroy(ctx);
    (void)hipFree(d_in); (void)hipFree(d_out);
    printf("thread_stress: misuse run finished (one ctx driven by two threads: undefined by contract, nothing checked)\n");
    return 0;
}

int main(int argc, char **argv)
{
    int threads = 4, rounds = 2, do_misuse = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--misuse")) { do_misuse = 1; continue; }
        if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
        else { fprintf(stderr, "usage: thread_stress [--threads T] [--rounds R] | --misuse\n"); return 2; }
    }
    if (threads < 1 || threads > 64 || rounds < 1) { fprintf(stderr, "thread_stress: bad --threads / --rounds\n"); return 2; }
    if (ovrfsr_abi_version() != OVRFSR_ABI_VERSION) { fprintf(stderr, "thread_stress: library ABI %u, header %u\n", ovrfsr_abi_version(), OVRFSR_ABI_VERSION); return 2; }
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "thread_stress: no device\n"); return 2; }
    if (do_misuse) return misuse();

    uint64_t ref[N_JOBS];
    err_t err;
    for (int j = 0; j < N_JOBS; ++j) /* the serial run */
        if (run_job(j, 100u + 10u * (uint32_t)j, &ref[j], &err, NULL) != 0) { fprintf(stderr, "thread_stress: serial job %d: %s\n", j, err.text); return 1; }
    { /* a checksum that could not see a wrong pipeline would prove nothing: the jobs that must differ do */
        uint64_t again = 0;
        if (run_job(1, 110u, &again, &err, NULL) != 0 || again != ref[1]) { fprintf(stderr, "thread_stress: job 1 is not reproducible when run alone\n"); return 1; }
        if (ref[0] == ref[1] || ref[1] == ref[2] || ref[6] == ref[1] || ref[7] == ref[1]) { fprintf(stderr, "thread_stress: jobs that must differ share a checksum\n"); return 1; }
    }
    const uint64_t host_ref = host_only();

    pthread_barrier_t gate;
    pthread_barrier_init(&gate, NULL, (unsigned)threads);
    worker_t *w = (worker_t *)calloc((size_t)threads, sizeof *w);
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    for (int i = 0; i < threads; ++i) {
        w[i].index = i; w[i].rounds = rounds; w[i].ref = ref; w[i].host_ref = host_ref; w[i].gate = &gate;
        const ovrfsr_config c1 = job_config(1);
        if (ovrfsr_create(0, &c1, &w[i].handed) != OVRFSR_OK) { fprintf(stderr, "thread_stress: ovrfsr_create for thread %d\n", i); return 1; }
        pthread_create(&th[i], NULL, worker, &w[i]);
    }
    int bad = 0;
    for (int i = 0; i < threads; ++i) {
        pthread_join(th[i], NULL);
        if (w[i].status) { fprintf(stderr, "thread_stress: thread %d: %s\n", i, w[i].err.text); bad = 1; }
        ovrfsr_destroy(w[i].handed); /* made here, driven there, destroyed here */
    }
    pthread_barrier_destroy(&gate);
    free(w); free(th);
    if (bad) return 1;
    printf("thread_stress: %d threads x %d rounds x %d jobs: all checksums equal the serial run\n", threads, rounds, N_JOBS);
    return 0;
}
