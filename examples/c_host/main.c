/*
 * c_host - a plain C host of libfpv_hip.so: no Python, no torch, only the C ABI of include/fpv_abi.h
 * and the HIP runtime for device memory.  It is what a non-Python integration of the reference's loop
 *     drone.reset(...); for i in range(k): drone.step(action[i], wind, [])     (src/core/simulator.py:59,:83-156)
 * looks like for N drones.
 *
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/c_host/main.c \
 *       -L fpyv_amd -lfpv_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/fpyv_amd -Wl,-rpath,/opt/rocm/lib -o c_host
 *   ./c_host params.bin actions.bin n k state_out.bin [fused | split]
 *
 * `split` is the split-phase layout of FpvVecEnv(partitions=2) written against the bare C ABI: TWO handles over the two
 * column halves of the SAME buffers (n_p drones each, global ids continuing: drone_id_offset + lo, every pointer moved
 * by lo elements, the same row stride), each stepped on its own stream without ever joining the other - two independent
 * kernel chains that overlap on the GPU.  The result is bit for bit that of the single batch.
 *
 * params.bin  = one fpv_params_t (written by the caller, e.g. fpyv_amd._lib.pack_params)
 * actions.bin = k * n * 4 float32 sticks;  state_out.bin receives rows * ld float32 + n reward float32 + n done bytes
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fpv_abi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define FPV_CHECK(x) do { int rc_ = (x); if (rc_ != FPV_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, fpv_error_name(rc_), fpv_last_error()); return 3; } } while (0)

static void* slurp(const char* path, size_t bytes)
{
    FILE* f = fopen(path, "rb");
    void* p = malloc(bytes);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, path); exit(1); }
    fclose(f);
    return p;
}

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s params.bin actions.bin n k state_out.bin [fused]\n", argv[0]); return 1; }
    const int64_t n = atoll(argv[3]);
    const int k = atoi(argv[4]);
    const int fused = argc > 6 && !strcmp(argv[6], "fused");
    const int split = argc > 6 && !strcmp(argv[6], "split");
    if (fpv_abi_version() != FPV_ABI_VERSION || fpv_sizeof(0) != (int)sizeof(fpv_params_t) || fpv_sizeof(1) != (int)sizeof(fpv_buffers_t)) {
        fprintf(stderr, "header and library disagree\n");
        return 1;
    }
    fpv_params_t* P = (fpv_params_t*)slurp(argv[1], sizeof(fpv_params_t));
    float* actions = (float*)slurp(argv[2], (size_t)k * n * 4 * sizeof(float));

    fpv_handle_t h = NULL;
    FPV_CHECK(fpv_create(P, n, 0, &h));
    const int rows = fpv_state_rows((int)P->mode);
    const int64_t ld = fpv_recommended_ld_device(n, 0);          /* the row stride for THIS device (negative: an error code) */
    if (ld < n) { fprintf(stderr, "fpv_recommended_ld_device: %s\n", fpv_last_error()); return 1; }

    fpv_buffers_t b;
    memset(&b, 0, sizeof(b));
    float* d_actions = NULL;
    HIP_OK(hipMalloc((void**)&b.state, (size_t)rows * ld * sizeof(float)));
    HIP_OK(hipMemset(b.state, 0, (size_t)rows * ld * sizeof(float)));
    HIP_OK(hipMalloc((void**)&d_actions, (size_t)k * n * 4 * sizeof(float)));
    HIP_OK(hipMalloc((void**)&b.reward, (size_t)n * sizeof(float)));
    HIP_OK(hipMalloc((void**)&b.done, (size_t)n));
    HIP_OK(hipMemcpy(d_actions, actions, (size_t)k * n * 4 * sizeof(float), hipMemcpyHostToDevice));
    b.ld = ld;
    b.action = d_actions;

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    FPV_CHECK(fpv_reset(h, &b, NULL, NULL, NULL, NULL, stream));                 /* Drone.reset with the params' defaults */
    if (fused) {
        FPV_CHECK(fpv_step_n(h, &b, k, n * 4, 0, stream));                       /* the whole loop in one launch */
    } else if (split) {
        /* two partitions of the same buffers: [0, lo) and [lo, n), lo a multiple of 128 (whole workgroups, aligned rows) */
        const int64_t lo = (n / 2) / 128 * 128;
        if (lo <= 0 || lo >= n) { fprintf(stderr, "split needs n >= 256\n"); return 1; }
        HIP_OK(hipStreamSynchronize(stream));                                     /* the reset is done before the chains start */
        fpv_handle_t hp[2] = {NULL, NULL};
        fpv_buffers_t bp[2];
        hipStream_t sp[2];
        for (int p = 0; p < 2; ++p) {
            const int64_t off = p ? lo : 0, np = p ? n - lo : lo;
            fpv_params_t Pp = *P;
            Pp.drone_id_offset = P->drone_id_offset + (uint64_t)off;              /* global ids: streams keyed by them do not move */
            FPV_CHECK(fpv_create(&Pp, np, 0, &hp[p]));
            bp[p] = b;
            bp[p].state = b.state + off;                                          /* column `off` of every row; ld unchanged */
            bp[p].reward = b.reward + off;
            bp[p].done = b.done + off;
            HIP_OK(hipStreamCreateWithFlags(&sp[p], hipStreamNonBlocking));
        }
        for (int t = 0; t < k; ++t)
            for (int p = 0; p < 2; ++p) {                                         /* nothing joins the two chains between steps */
                bp[p].action = d_actions + ((size_t)t * n + (p ? lo : 0)) * 4;
                FPV_CHECK(fpv_step(hp[p], &bp[p], sp[p]));
            }
        for (int p = 0; p < 2; ++p) { HIP_OK(hipStreamSynchronize(sp[p])); fpv_destroy(hp[p]); (void)hipStreamDestroy(sp[p]); }
    } else {
        for (int t = 0; t < k; ++t) {                                             /* one Drone.step per launch */
            b.action = d_actions + (size_t)t * n * 4;
            FPV_CHECK(fpv_step(h, &b, stream));
        }
    }
    HIP_OK(hipStreamSynchronize(stream));

    float* state = (float*)malloc((size_t)rows * ld * sizeof(float));
    float* reward = (float*)malloc((size_t)n * sizeof(float));
    unsigned char* done = (unsigned char*)malloc((size_t)n);
    HIP_OK(hipMemcpy(state, b.state, (size_t)rows * ld * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(reward, b.reward, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done, b.done, (size_t)n, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[5], "wb");
    if (!f) return 1;
    fwrite(state, sizeof(float), (size_t)rows * ld, f);
    fwrite(reward, sizeof(float), (size_t)n, f);
    fwrite(done, 1, (size_t)n, f);
    fclose(f);
    printf("c_host: %lld drones x %d steps (%s), ld = %lld, drone 0 at (%.6f, %.6f, %.6f)\n", (long long)n, k,
           fused ? "fpv_step_n" : split ? "two partitions, two streams" : "fpv_step", (long long)ld, state[0], state[ld], state[2 * ld]);
    fpv_destroy(h);
    (void)hipFree(b.state); (void)hipFree(d_actions); (void)hipFree(b.reward); (void)hipFree(b.done);
    (void)hipStreamDestroy(stream);
    free(state); free(reward); free(done); free(actions); free(P);
    return 0;
}
