// Random search over where the 14 row streams sit inside ONE large allocation: which relative placements make HBM slow?
// Each configuration: row r at base + slot_r * 32 MiB + sub_r (slot_r distinct, sub_r a multiple of 256 B below 1 MiB; rows are
// 32 MiB long, the slots 33 MiB apart so that rows never overlap).  In-place read + write of 14 dwords per lane, 2^23 lanes.
// Output: one line per configuration: time, then (slot, sub/256) per row - for offline regression.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_variants/stream_rows_search tools/stream_rows_search.hip
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
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
int main(int argc, char** argv)
{
    const uint32_t n = 1u << 23;
    const int configs = argc > 1 ? atoi(argv[1]) : 200;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;            // 0 random slots + subs, 1 consecutive slots + random subs, 2 random slots, sub = 0
    const size_t slot = 33u << 20;
    const int nslots = 60;
    char* base; CK(hipMalloc((void**)&base, slot * nslots)); CK(hipMemset(base, 0, slot * nslots));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = n / 128;
    printf("# base %p mode %d\n", (void*)base, mode);
    for (int c = 0; c < configs; ++c) {
        int slots[R]; uint32_t subs[R];
        std::vector<int> perm(nslots); for (int k = 0; k < nslots; ++k) perm[k] = k;
        for (int k = 0; k < R; ++k) { int j = k + rnd() % (nslots - k); std::swap(perm[k], perm[j]); }
        const int first = rnd() % (nslots - R);
        Rows rows;
        for (int r = 0; r < R; ++r) {
            slots[r] = mode == 1 ? first + r : perm[r];
            subs[r] = mode == 2 ? 0 : rnd() % 4096;
            rows.p[r] = (float*)(base + slots[r] * slot + (size_t)subs[r] * 256);
        }
        for (int k = 0; k < 2; ++k) rows_inplace<<<grid, 128>>>(rows, n);
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < 10; ++k) rows_inplace<<<grid, 128>>>(rows, n);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%8.2f", ms * 100.0);
        for (int r = 0; r < R; ++r) printf(" %d:%u", slots[r], subs[r]);
        printf("\n");
    }
    return 0;
}
