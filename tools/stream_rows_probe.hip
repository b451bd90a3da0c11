// Follow-up to stream_layout_probe.hip: does the RELATIVE position of the 14 row streams decide what HBM delivers?
// Every lane reads 14 dwords (one per row) and writes them back in place; the rows are given as 14 pointers:
//   stride     row r at base + r * ld               (the shipped layout; ld = n + 256 floats)
//   separate   14 separate hipMalloc's              (each row wherever the driver put it)
//   skew X     row r at base + r * ld + skew_r      (skew_r = pseudo-random multiples of a granule inside a window)
//   hipcc -O3 --offload-arch=gfx950 -o tools/_variants/stream_rows_probe tools/stream_rows_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int R = 14;
struct Rows { float* p[R]; };

__global__ __launch_bounds__(128) void rows_inplace(Rows rows, uint32_t n)
{
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = rows.p[r][i];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += v[r];
#pragma unroll
    for (int r = 0; r < R; ++r) rows.p[r][i] = v[r] + s * 1e-9f;
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

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : (1u << 23);
    const int64_t ld = (int64_t)n + 256;
    const size_t window = 4u << 20;                       // room for the skews
    const size_t bytes = (size_t)R * ld * 4 + window * R;
    const double moved = 2.0 * R * n * 4;
    const unsigned grid = (n + 127) / 128;
    float *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    const int64_t n4 = (int64_t)R * n / 4;
    double t = time_us([&] { copy4<<<(unsigned)((n4 + 255) / 256), 256>>>((float4*)b, (const float4*)a, n4); });
    printf("copy4                                  : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
    Rows rows;
    for (int r = 0; r < R; ++r) rows.p[r] = a + (int64_t)r * ld;
    t = time_us([&] { rows_inplace<<<grid, 128>>>(rows, n); });
    printf("stride ld = n + 256                    : %8.2f us  %7.0f GB/s\n", t, moved / t / 1e3);
    for (int trial = 0; trial < 3; ++trial) {
        std::vector<void*> keep;
        for (int r = 0; r < R; ++r) {
            void* sp; CK(hipMalloc(&sp, (size_t)(1 + (rnd() % 61)) << 20)); keep.push_back(sp);
            void* p; CK(hipMalloc(&p, (size_t)n * 4)); CK(hipMemset(p, 0, (size_t)n * 4)); rows.p[r] = (float*)p; keep.push_back(p);
        }
        t = time_us([&] { rows_inplace<<<grid, 128>>>(rows, n); });
        printf("separate allocations, trial %d          : %8.2f us  %7.0f GB/s   (row 0 %p row 1 %p)\n", trial, t, moved / t / 1e3, (void*)rows.p[0], (void*)rows.p[1]);
        for (void* p : keep) CK(hipFree(p));
    }
    struct { const char* name; size_t granule, span; } pats[] = {
        {"skew 256 B x rnd in 4 KiB", 256, 4096}, {"skew 256 B x rnd in 64 KiB", 256, 65536}, {"skew 4 KiB x rnd in 2 MiB", 4096, 2u << 20},
        {"skew 64 KiB x rnd in 4 MiB", 65536, 4u << 20}, {"skew 1 KiB x rnd in 1 MiB", 1024, 1u << 20}, {"skew 2 MiB x rnd? (0 or 2 MiB)", 2u << 20, 4u << 20}};
    for (auto& pt : pats)
        for (int trial = 0; trial < 3; ++trial) {
            for (int r = 0; r < R; ++r) rows.p[r] = a + (int64_t)r * (ld + (int64_t)window / 4) + (rnd() % (pt.span / pt.granule)) * (pt.granule / 4);
            t = time_us([&] { rows_inplace<<<grid, 128>>>(rows, n); });
            printf("%-32s trial %d: %8.2f us  %7.0f GB/s\n", pt.name, trial, t, moved / t / 1e3);
        }
    // constant strides again, wider range (floats past n)
    for (int64_t pad : {256, 4096 + 256, 65536 + 256, (1 << 20) + 256, (1 << 20) + 65536 + 4096 + 256, 3 * (1 << 18) + 1024 + 256}) {
        for (int r = 0; r < R; ++r) rows.p[r] = a + (int64_t)r * (n + pad);
        t = time_us([&] { rows_inplace<<<grid, 128>>>(rows, n); });
        printf("stride n + %-10lld                    : %8.2f us  %7.0f GB/s\n", (long long)pad, t, moved / t / 1e3);
    }
    return 0;
}
