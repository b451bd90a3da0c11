// Feasibility probe for a 4-lanes-per-drone AoS design: one float4 load + one float4 store per lane
// (the only structure measured at 7.3 TB/s), one shared action float4 per quad, plus X dependent
// FMAs and a quad DPP exchange every 8 FMAs per lane.  How much per-lane VALU work hides?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int X>
__global__ __launch_bounds__(256) void k_quad(v4* __restrict__ st, const v4* __restrict__ act, size_t n4, float c)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    v4 v = st[i];
    const v4 a = __builtin_nontemporal_load(&act[i >> 2]);
    float x = v.x + a.x, y = v.y + a.y, z = v.z + a.z, w = v.w + a.w;
#pragma unroll
    for (int k = 0; k < X; k += 8) {
        x = fmaf(x, c, y); y = fmaf(y, c, z); z = fmaf(z, c, w); w = fmaf(w, c, x);
        x = fmaf(x, c, z); y = fmaf(y, c, w); z = fmaf(z, c, x); w = fmaf(w, c, y);
        // quad exchange: read lane^1 and lane^2 of the quad (DPP quad_perm)
        x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        z += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    }
    v.x = x; v.y = y; v.z = z; v.w = w;
    st[i] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F> double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); std::vector<float> ts;
  for (int r = 0; r < 7; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(r * reps + i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms / reps); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2] * 1e-3; }
template <int X> void run(v4* st, v4* act, size_t n, int ring) {
  const size_t n4 = n * 4; const int bs = 256;
  double t = timeit([&](int it) { k_quad<X><<<(unsigned)((n4 + bs - 1) / bs), bs>>>(st, act + (size_t)(it % ring) * n, n4, 0.999f); }, 64);
  printf("X=%4d FMAs/lane (+%3d DPP): %7.2f us per 2^20-drone step   %7.1f GB/s on 144 B/drone  (133 B-equivalent %7.1f GB/s)\n", X, X / 4, t * 1e6, 144.0 * n / t / 1e9, 133.0 * n / t / 1e9); }
int main() {
  const size_t n = 1u << 20; const int ring = 32;
  v4 *st, *act; CK(hipMalloc(&st, n * 64)); CK(hipMalloc(&act, (size_t)ring * n * 16));
  CK(hipMemset(st, 0, n * 64)); CK(hipMemset(act, 0, (size_t)ring * n * 16));
  run<0>(st, act, n, ring); run<64>(st, act, n, ring); run<128>(st, act, n, ring); run<192>(st, act, n, ring);
  run<256>(st, act, n, ring); run<320>(st, act, n, ring); run<400>(st, act, n, ring); run<512>(st, act, n, ring);
  return 0; }
