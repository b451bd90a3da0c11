// Streaming ceilings on this chip for working sets inside / outside the 256 MiB Infinity Cache.
// Build: hipcc -O3 --offload-arch=gfx950 -o stream_probe stream_probe.hip ; run: ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void k_rmw4(v4* x, size_t n4) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n4) { v4 v = x[i]; v *= 1.0001f; x[i] = v; } }
__global__ void k_rmw1(float* x, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f; }
__global__ void k_copy4(v4* y, const v4* x, size_t n4) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n4) y[i] = x[i]; }
__global__ void k_read4(const v4* x, size_t n4, float* sink) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n4) { v4 v = x[i]; if (v.x == 123.456f) sink[0] = v.y; } }
__global__ void k_write4(v4* x, size_t n4) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n4) x[i] = (v4){1.f, 2.f, 3.f, 4.f}; }
// 14 row streams RMW like the step kernel (dword per lane), padded stride
__global__ void k_rows(float* x, size_t n, size_t ld) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { float v[14];
#pragma unroll
  for (int r = 0; r < 14; ++r) v[r] = x[r * ld + i];
#pragma unroll
  for (int r = 0; r < 14; ++r) x[r * ld + i] = v[r] * 1.0001f; } }
// the same 14 streams, but written to a SECOND set of rows (ping-pong state): no read->write turnaround on one address
__global__ void k_rows_copy(float* y, const float* x, size_t n, size_t ld) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { float v[14];
#pragma unroll
  for (int r = 0; r < 14; ++r) v[r] = x[r * ld + i];
#pragma unroll
  for (int r = 0; r < 14; ++r) y[r * ld + i] = v[r] * 1.0001f; } }
typedef float v2 __attribute__((ext_vector_type(2)));
// AoSoA planes: 3 float4 planes + 1 float2 plane per drone (56 B), one drone per lane
__global__ void k_planes(v4* p0, v4* p1, v4* p2, v2* p3, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) {
  v4 a = p0[i], b = p1[i], c = p2[i]; v2 d = p3[i]; a *= 1.0001f; b *= 1.0001f; c *= 1.0001f; d *= 1.0001f; p0[i] = a; p1[i] = b; p2[i] = c; p3[i] = d; } }
// AoS rows of 16 floats (64 B), one row per lane as 4 float4
__global__ void k_aos16(v4* x, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) {
  v4 a = x[4 * i], b = x[4 * i + 1], c = x[4 * i + 2], d = x[4 * i + 3]; a *= 1.0001f; b *= 1.0001f; c *= 1.0001f; d *= 1.0001f;
  x[4 * i] = a; x[4 * i + 1] = b; x[4 * i + 2] = c; x[4 * i + 3] = d; } }
// 4 float4 planes (64 B per drone, SoA of float4)
__global__ void k_planes4(v4* p0, v4* p1, v4* p2, v4* p3, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) {
  v4 a = p0[i], b = p1[i], c = p2[i], d = p3[i]; a *= 1.0001f; b *= 1.0001f; c *= 1.0001f; d *= 1.0001f; p0[i] = a; p1[i] = b; p2[i] = c; p3[i] = d; } }
// tiled AoSoA: [n/W][14][W] - the 14 rows of W consecutive drones are contiguous, so the kernel walks memory linearly
template <int W> __global__ void k_rows_tiled(float* x, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { float v[14];
  float* b = x + (i / W) * (14 * W) + (i % W);
#pragma unroll
  for (int r = 0; r < 14; ++r) v[r] = b[r * W];
#pragma unroll
  for (int r = 0; r < 14; ++r) b[r * W] = v[r] * 1.0001f; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F> double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); std::vector<float> ts;
  for (int r = 0; r < 7; ++r) { CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms / reps); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2] * 1e-3; }
int main(int argc, char** argv) {
  std::vector<size_t> sizes = {59u << 20, 118u << 20};                 // or: ./stream_probe <MiB> <MiB> ... (<= 1024)
  if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; ++i) sizes.push_back((size_t)atoi(argv[i]) << 20); }
  float* buf; float* buf2; float* sink; CK(hipMalloc(&buf, 1024u << 20)); CK(hipMalloc(&buf2, 1024u << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, 1024u << 20)); CK(hipMemset(buf2, 0, 1024u << 20));
  for (size_t bytes : sizes) {
    size_t n = bytes / 4, n4 = n / 4; int bs = 256; int reps = (int)std::max<size_t>(20, (4096u << 20) / bytes);
    double t;
    t = timeit([&] { k_rmw4<<<(n4 + bs - 1) / bs, bs>>>((v4*)buf, n4); }, reps); printf("%5zu MiB rmw float4 in-place : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * bytes / t / 1e9);
    t = timeit([&] { k_rmw1<<<(n + bs - 1) / bs, bs>>>(buf, n); }, reps); printf("%5zu MiB rmw dword  in-place : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * bytes / t / 1e9);
    t = timeit([&] { k_copy4<<<(n4 + bs - 1) / bs, bs>>>((v4*)buf2, (const v4*)buf, n4); }, reps); printf("%5zu MiB copy float4         : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * bytes / t / 1e9);
    t = timeit([&] { k_read4<<<(n4 + bs - 1) / bs, bs>>>((const v4*)buf, n4, sink); }, reps); printf("%5zu MiB read float4         : %7.2f us  %7.1f GB/s (R)\n", bytes >> 20, t * 1e6, 1.0 * bytes / t / 1e9);
    t = timeit([&] { k_write4<<<(n4 + bs - 1) / bs, bs>>>((v4*)buf, n4); }, reps); printf("%5zu MiB write float4        : %7.2f us  %7.1f GB/s (W)\n", bytes >> 20, t * 1e6, 1.0 * bytes / t / 1e9);
    { size_t nd2 = bytes / 56; size_t pl = ((nd2 + 63) / 64) * 64 + 64; v4* q = (v4*)buf;
      for (int b2 : {128, 256}) { t = timeit([&] { k_planes<<<(nd2 + b2 - 1) / b2, b2>>>(q, q + pl, q + 2 * pl, (v2*)(q + 3 * pl), nd2); }, reps); printf("%5zu MiB planes 3xf4+f2 bs=%3d  : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, b2, t * 1e6, 2.0 * nd2 * 56 / t / 1e9); }
      size_t na = bytes / 64; t = timeit([&] { k_aos16<<<(na + 127) / 128, 128>>>((v4*)buf, na); }, reps); printf("%5zu MiB aos 64B rows bs=128     : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * na * 64 / t / 1e9);
      size_t pl4 = ((na + 63) / 64) * 64 + 64; t = timeit([&] { k_planes4<<<(na + 127) / 128, 128>>>(q, q + pl4, q + 2 * pl4, q + 3 * pl4, na); }, reps); printf("%5zu MiB planes 4xf4 bs=128      : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * na * 64 / t / 1e9); }
    { size_t nt = (bytes / 56) / 1024 * 1024;
      t = timeit([&] { k_rows_tiled<64><<<nt / 128, 128>>>(buf, nt); }, reps); printf("%5zu MiB tiled W=64  bs=128     : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * nt * 56 / t / 1e9);
      t = timeit([&] { k_rows_tiled<128><<<nt / 128, 128>>>(buf, nt); }, reps); printf("%5zu MiB tiled W=128 bs=128     : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * nt * 56 / t / 1e9);
      t = timeit([&] { k_rows_tiled<256><<<nt / 256, 256>>>(buf, nt); }, reps); printf("%5zu MiB tiled W=256 bs=256     : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * nt * 56 / t / 1e9);
      t = timeit([&] { k_rows_tiled<1024><<<nt / 128, 128>>>(buf, nt); }, reps); printf("%5zu MiB tiled W=1024 bs=128    : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * nt * 56 / t / 1e9); }
    size_t nd = bytes / 56; size_t ld = ((nd + 63) / 64) * 64 + 256; if (14 * ld * 4 <= (1024u << 20)) {
      for (int b2 : {128, 256}) { t = timeit([&] { k_rows<<<(nd + b2 - 1) / b2, b2>>>(buf, nd, ld); }, reps); printf("%5zu MiB 14-row rmw bs=%3d     : %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, b2, t * 1e6, 2.0 * nd * 56 / t / 1e9); }
      { bool flip = false; t = timeit([&] { flip = !flip; k_rows_copy<<<(nd + 127) / 128, 128>>>(flip ? buf2 : buf, flip ? buf : buf2, nd, ld); }, reps); printf("%5zu MiB 14-row ping-pong bs=128: %7.2f us  %7.1f GB/s (R+W)\n", bytes >> 20, t * 1e6, 2.0 * nd * 56 / t / 1e9); } }
  }
  return 0; }
