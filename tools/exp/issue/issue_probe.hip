// issue_probe.hip - how many shader cycles one SIMD of gfx950 needs per wave64 VALU instruction, by instruction form and
// by the number of waves resident on the SIMD.  EXPERIMENT (tools/exp): not part of the product.
//
// Why: the k-step kernel executes ~187 VALU instructions per env-step and runs at 3.4 cycles per instruction and SIMD
// with 7-8 waves resident, while the guide's table gives 2 cycles for v_fma_f32 (4 for one wave alone).  Removing 5 % of
// the instructions, or running 8 spill-free waves instead of 7, did not change the time (profiles/r03_exp_kstep_spillfree.log).
// This probe measures the issue cost of the instruction FORMS that loop is made of.
//
//   hipcc -O2 --offload-arch=gfx950 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int kUnroll = 64;       // instructions per loop iteration (8 accumulators x 8)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, long long* cycles, int iters, float sb, float sc)
{
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = threadIdx.x * 1e-3f + u;
    float b = sb, c = sc;
    asm volatile("" : "+v"(b), "+v"(c));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4], pb = {sb, sb}, pc = {sc, sc};
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = f2{a[2 * u], a[2 * u + 1]};
    asm volatile("" : "+v"(pb), "+v"(pc));
    unsigned ia = threadIdx.x | 1u;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < kUnroll / 8; ++r) {
#define ONE(u)                                                                                                              \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));                                   \
    else if (KIND == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));                              \
    else if (KIND == 2) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b));                                       \
    else if (KIND == 3) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[u]) : "s"(sb));                                      \
    else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "s"(sb), "v"(c));                              \
    else if (KIND == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7ff000" : "+v"(a[u]) : "v"(b));                             \
    else if (KIND == 6) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b));                                       \
    else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u & 3]) : "v"(pb), "v"(pc));                      \
    else if (KIND == 8) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : "vcc");                      \
    else if (KIND == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));                              \
    else if (KIND == 10) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[u]));                                                 \
    else if (KIND == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia) : "v"(ia));                                       \
    else if (KIND == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));                              \
    else if (KIND == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u & 1]) : "v"(b), "v"(c));                          \
    else if (KIND == 14) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[u]) : "v"(b));                                          \
    else if (KIND == 15) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : "vcc"); \
    else if (KIND == 16) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "s"(sc));                             \
    else if (KIND == 17) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(a[u]) : "v"(b));                                     \
    else if (KIND == 18) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a[u]) : "v"(b));                                     \
    else if (KIND == 19) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s20, s20, 1" : "+v"(a[u]) : "v"(b), "v"(c) : "s20"); \
    else if (KIND == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(*(unsigned long long*)&p[u & 3]) : "v"(ia) : "vcc");
            REP8(ONE)
#undef ONE
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += a[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) sum += p[u].x + p[u].y;
    sum += (float)ia;
    if (sum == 123.456f) out[0] = sum;                       // keeps the chains alive
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*Kern)(float*, long long*, int, float, float);
struct Case { const char* name; Kern k; int per_asm; };

int main()
{
    const Case cases[] = {
        {"v_fma_f32 v,v,v,v        (VOP3, 8 chains)", probe<0>, 1}, {"v_fmac_f32 v,v,v         (VOP2)", probe<1>, 1},
        {"v_mul_f32 v,v,v          (VOP2)", probe<2>, 1},           {"v_mul_f32 v,S,v          (VOP2, SGPR)", probe<3>, 1},
        {"v_fma_f32 v,v,S,v        (VOP3, SGPR)", probe<4>, 1},     {"v_fma_f32 v,v,v,S        (VOP3, SGPR addend)", probe<16>, 1},
        {"v_fmaak_f32 v,v,v,LIT    (8-byte literal)", probe<5>, 1}, {"v_fma_f32 v,v,v,1.0      (inline constant)", probe<18>, 1},
        {"v_mul_f32_e64 v,v,-v     (VOP3, neg)", probe<17>, 1},     {"v_add_f32 v,v,v          (VOP2)", probe<6>, 1},
        {"v_pk_fma_f32             (packed, 2 flop pairs)", probe<7>, 1}, {"v_cndmask_b32 ..,vcc", probe<8>, 1},
        {"v_med3_f32", probe<9>, 1},                                {"v_sqrt_f32               (transcendental)", probe<10>, 1},
        {"v_mul_lo_u32             (1 chain)", probe<11>, 1},       {"v_mad_u64_u32", probe<20>, 1},
        {"v_fma_f32  1 dependent chain", probe<12>, 1},             {"v_fma_f32  2 chains", probe<13>, 1},
        {"v_mov_b32", probe<14>, 1},                                {"v_cmp_gt_f32 vcc + v_cndmask (pair)", probe<15>, 2},
        {"v_fma_f32 + s_add_u32 (VALU+SALU pair)", probe<19>, 1},
    };
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    long long* cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, sizeof(long long) * cus * 8 * 4 * 2);
    const int iters = 2000;
    printf("%-52s %s\n", "instruction form", "cycles per instruction PER SIMD at 1 / 2 / 4 / 8 waves per SIMD   [wall-clock GHz implied]");
    for (const Case& c : cases) {
        printf("%-52s", c.name);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;                       // 256 threads = 4 waves = one wave per SIMD of a CU
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 50, 1.0001f, 1e-7f);      // warm-up
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 1e-7f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * 4);
            hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double n_inst = (double)iters * kUnroll * c.per_asm;
            const double per_simd = (double)h[h.size() / 2] / n_inst / w;           // median wave's cycles / instructions / waves sharing the SIMD
            const double ghz = (double)h[h.size() / 2] / (ms * 1e6);                // shader cycles per ns while the kernel ran (tail excluded: ~)
            printf("  %5.2f", per_simd);
            if (w == 8) printf("   [%4.2f GHz, %.0f us]", ghz, ms * 1e3);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        printf("\n");
    }
    return 0;
}
