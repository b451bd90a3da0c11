#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// wait until at most ONE vector-memory operation is outstanding, then read 16 bytes of LDS
__device__ __forceinline__ v4 lds_read_after_vmcnt1(unsigned addr)
{
    v4 r;
    asm volatile("s_waitcnt vmcnt(1)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__global__ __launch_bounds__(128) void k(const v4* __restrict__ src, v4* __restrict__ dst, int steps, long stride)
{
    __shared__ v4 sa[128];
    __shared__ v4 sb[128];
    const unsigned tid = threadIdx.x, wave0 = tid & ~63u;
    const v4* p = src + blockIdx.x * 128 + tid;      // per-lane global address; the LDS side is M0 (wave-uniform) + lane * 16
    const unsigned la = (unsigned)(uintptr_t)LDS_PTR(&sa[tid]), lb = (unsigned)(uintptr_t)LDS_PTR(&sb[tid]);
    __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(&sa[wave0]), 16, 0, 2);
    __builtin_amdgcn_global_load_lds(GLB_PTR(p + stride), LDS_PTR(&sb[wave0]), 16, 0, 2);
    v4 acc = {0, 0, 0, 0};
    int t = 0;
    for (; t + 2 <= steps; t += 2) {
        v4 a = lds_read_after_vmcnt1(la);
        __builtin_amdgcn_global_load_lds(GLB_PTR(p + (long)(t + 2 < steps ? t + 2 : steps - 1) * stride), LDS_PTR(&sa[wave0]), 16, 0, 2);
        acc = acc * 0.5f + a;
        for (int j = 0; j < 40; ++j) acc = acc * 1.0001f + a * 0.001f;
        v4 b = lds_read_after_vmcnt1(lb);
        __builtin_amdgcn_global_load_lds(GLB_PTR(p + (long)(t + 3 < steps ? t + 3 : steps - 1) * stride), LDS_PTR(&sb[wave0]), 16, 0, 2);
        acc = acc * 0.5f + b;
        for (int j = 0; j < 40; ++j) acc = acc * 1.0001f + b * 0.001f;
    }
    dst[blockIdx.x * 128 + tid] = acc;
}
int main()
{
    const int n = 128 * 1000, steps = 16; const long stride = n;
    std::vector<v4> h((size_t)n * steps), out(n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (v4){(float)(i % 97) * 0.01f, (float)(i % 89) * 0.02f, (float)(i % 83), 1.0f};
    v4 *d, *o; hipMalloc(&d, h.size() * 16); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    k<<<n / 128, 128>>>(d, o, steps, stride);
    hipMemcpy(out.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        v4 acc = {0, 0, 0, 0};
        for (int t = 0; t < steps; ++t) { v4 a = h[(size_t)t * stride + i]; acc = acc * 0.5f + a; for (int j = 0; j < 40; ++j) acc = acc * 1.0001f + a * 0.001f; }
        for (int c = 0; c < 4; ++c) if (fabsf(acc[c] - out[i][c]) > 1e-3f * fabsf(acc[c]) + 1e-5f) { if (bad < 5) printf("mismatch i=%d c=%d %g vs %g\n", i, c, acc[c], out[i][c]); ++bad; }
    }
    printf("bad = %d of %d\n", bad, n * 4);
    return bad != 0;
}
