// How many concurrent row streams does HBM like?  A physics-free model of the step kernel's memory shape beyond the Infinity
// Cache: every lane reads R dwords (one per state row) and writes R dwords back, in place, for 2^23 lanes.
//   soa        rows [R][ld]: R read streams and R write streams 32 MiB apart (what fpv_drone_step_kernel does)
//   tiled<T>   rows [n/T][R][T]: a workgroup's R rows sit within R*T*4 contiguous bytes (one stream, like a copy)
//   copy4      dst[i] = src[i], 16 bytes per lane (the streaming ceiling, fpv_diag_stream_copy_wide's shape), same byte count
// Several allocations per process ("re-rolls": a spacer allocation moves the next buffer elsewhere) show how much of the
// spread between processes is where the buffer landed.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_variants/stream_layout_probe tools/stream_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int R = 14;

template <bool INPLACE>
__global__ __launch_bounds__(128) void rows_soa(float* __restrict__ dst, const float* __restrict__ src, int64_t ld, uint32_t n)
{
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(int64_t)r * ld + i];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
#pragma unroll
    for (int r = 0; r < R; ++r) (INPLACE ? const_cast<float*>(src) : dst)[(int64_t)r * ld + i] = v[r] + s * 1e-9f;
}

template <bool INPLACE>
__global__ __launch_bounds__(128) void rows_tiled(float* __restrict__ dst, const float* __restrict__ src, uint32_t T, uint32_t n)
{
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    const int64_t base = (int64_t)(i / T) * R * T + (i % T);
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[base + (int64_t)r * T];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
#pragma unroll
    for (int r = 0; r < R; ++r) (INPLACE ? const_cast<float*>(src) : dst)[base + (int64_t)r * T] = v[r] + s * 1e-9f;
}

__global__ __launch_bounds__(256) void copy4(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

template <class F>
static double time_us(F&& launch, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) launch();
    CK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; ++k) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3 / reps);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[1];
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : (1u << 23);
    const int rolls = argc > 2 ? atoi(argv[2]) : 4;
    const int64_t ld = (int64_t)n + 256;
    const size_t bytes = (size_t)R * ld * 4;
    const double moved = 2.0 * R * n * 4;          // read + write
    printf("n = %u, R = %d rows, %.0f MB per buffer, %.0f MB moved per launch\n", n, R, bytes / 1e6, moved / 1e6);
    std::vector<void*> spacers;
    for (int roll = 0; roll < rolls; ++roll) {
        void* sp = nullptr;
        CK(hipMalloc(&sp, (size_t)(3 + 37 * roll) << 20));       // odd-sized spacer: the next buffers land elsewhere
        spacers.push_back(sp);
        float *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
        const unsigned grid = (n + 127) / 128;
        printf("roll %d: a = %p  b = %p\n", roll, (void*)a, (void*)b);
        const int64_t n4 = (int64_t)R * n / 4;
        double t = time_us([&] { copy4<<<(unsigned)((n4 + 255) / 256), 256>>>((float4*)b, (const float4*)a, n4); });
        printf("  copy4 (2 streams, 16 B/lane)      : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
        t = time_us([&] { rows_soa<false><<<grid, 128>>>(b, a, ld, n); });
        printf("  soa  a->b  (14 + 14 streams)      : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
        t = time_us([&] { rows_soa<true><<<grid, 128>>>(a, a, ld, n); });
        printf("  soa  in place (14 streams r+w)    : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
        for (uint32_t T : {128u, 256u, 512u, 1024u, 4096u, 16384u, 65536u, 1048576u}) {
            t = time_us([&] { rows_tiled<true><<<grid, 128>>>(a, a, T, n); });
            printf("  tiled in place T = %-8u       : %8.2f us  %7.0f GB/s\n", T, t, moved / t / 1e3);
        }
        t = time_us([&] { rows_tiled<false><<<grid, 128>>>(b, a, 256u, n); });
        printf("  tiled a->b T = 256                : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
        CK(hipFree(a)); CK(hipFree(b));
        fflush(stdout);
    }
    for (void* sp : spacers) CK(hipFree(sp));
    return 0;
}
