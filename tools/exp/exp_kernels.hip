// Experimental launch geometries / access shapes for the step kernel.  NOT part of the product:
// built into tools/exp/libfpv_exp.so by tools/exp/run_exp.py and timed A/B in one process.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fpv_abi.h"
#include "../../fpyv_amd/csrc/fpv_derive.h"
#include "../../fpyv_amd/csrc/fpv_math.h"

struct Buf {
    float* state; int64_t ld; const float4* action; float* reward; uint8_t* done; float wx, wy, wz;
};

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

#define LD(row) st[(int64_t)(row) * ld + i]

__device__ __forceinline__ void ld_drone(const float* __restrict__ st, int64_t ld, int64_t i, FpvDroneState& s)
{
    s.px = LD(0); s.py = LD(1); s.pz = LD(2); s.vx = LD(3); s.vy = LD(4); s.vz = LD(5);
    s.q.w = LD(6); s.q.x = LD(7); s.q.y = LD(8); s.q.z = LD(9); s.rx = LD(10); s.ry = LD(11); s.rz = LD(12); s.thrust = LD(13);
}
__device__ __forceinline__ void st_drone(float* __restrict__ st, int64_t ld, int64_t i, const FpvDroneState& s)
{
    LD(0) = s.px; LD(1) = s.py; LD(2) = s.pz; LD(3) = s.vx; LD(4) = s.vy; LD(5) = s.vz;
    LD(6) = s.q.w; LD(7) = s.q.x; LD(8) = s.q.y; LD(9) = s.q.z; LD(10) = s.rx; LD(11) = s.ry; LD(12) = s.rz; LD(13) = s.thrust;
}

// V0: product shape, block size BS, optional passthrough (no physics) and nt hints
template <int BS, bool PASS, bool NT>
__global__ __launch_bounds__(BS) void k_base(const FpvK K, const Buf B, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    FpvDroneState s;
    float4 a = NT ? nt_load4(&B.action[i]) : B.action[i];
    ld_drone(B.state, B.ld, i, s);
    float reward; bool done;
    if (PASS) {
        s.px += a.x * 1e-9f; s.py += a.y * 1e-9f; s.pz += a.z * 1e-9f; s.thrust += a.w * 1e-9f;
        reward = s.vx; done = s.vy > 1e30f;
    } else {
        FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
        reward = o.reward; done = o.done;
    }
    st_drone(B.state, B.ld, i, s);
    if (NT) { __builtin_nontemporal_store(reward, &B.reward[i]); __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &B.done[i]); }
    else { B.reward[i] = reward; B.done[i] = done ? 1 : 0; }
}

// V1: 4 consecutive drones per lane, float4 row loads/stores; actions as 4 float4 loads (64 B per lane)
template <bool PASS>
__global__ __launch_bounds__(256) void k_vec4(const FpvK K, const Buf B, const int64_t n)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;   // n % 4 == 0 required
    float4 r[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) r[k] = *reinterpret_cast<const float4*>(&B.state[(int64_t)k * B.ld + i0]);
    float4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = B.action[i0 + j];
    float rew[4]; uint8_t dn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        FpvDroneState s;
        const float* f = reinterpret_cast<const float*>(r);
        s.px = f[0 * 4 + j]; s.py = f[1 * 4 + j]; s.pz = f[2 * 4 + j]; s.vx = f[3 * 4 + j]; s.vy = f[4 * 4 + j]; s.vz = f[5 * 4 + j];
        s.q.w = f[6 * 4 + j]; s.q.x = f[7 * 4 + j]; s.q.y = f[8 * 4 + j]; s.q.z = f[9 * 4 + j];
        s.rx = f[10 * 4 + j]; s.ry = f[11 * 4 + j]; s.rz = f[12 * 4 + j]; s.thrust = f[13 * 4 + j];
        if (PASS) {
            s.px += a[j].x * 1e-9f; s.py += a[j].y * 1e-9f; s.pz += a[j].z * 1e-9f; s.thrust += a[j].w * 1e-9f;
            rew[j] = s.vx; dn[j] = s.vy > 1e30f;
        } else {
            FpvStepOut o = fpv_drone_step_lane<false>(K, s, a[j].x, a[j].y, a[j].z, a[j].w, B.wx, B.wy, B.wz);
            rew[j] = o.reward; dn[j] = o.done;
        }
        float* g = reinterpret_cast<float*>(r);
        g[0 * 4 + j] = s.px; g[1 * 4 + j] = s.py; g[2 * 4 + j] = s.pz; g[3 * 4 + j] = s.vx; g[4 * 4 + j] = s.vy; g[5 * 4 + j] = s.vz;
        g[6 * 4 + j] = s.q.w; g[7 * 4 + j] = s.q.x; g[8 * 4 + j] = s.q.y; g[9 * 4 + j] = s.q.z;
        g[10 * 4 + j] = s.rx; g[11 * 4 + j] = s.ry; g[12 * 4 + j] = s.rz; g[13 * 4 + j] = s.thrust;
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) *reinterpret_cast<float4*>(&B.state[(int64_t)k * B.ld + i0]) = r[k];
    *reinterpret_cast<float4*>(&B.reward[i0]) = make_float4(rew[0], rew[1], rew[2], rew[3]);
    *reinterpret_cast<uchar4*>(&B.done[i0]) = make_uchar4(dn[0], dn[1], dn[2], dn[3]);
}

// V2: persistent waves, register double buffer: loads of tile t+1 are in flight while tile t computes
template <bool PASS>
__global__ __launch_bounds__(256) void k_prefetch(const FpvK K, const Buf B, const int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    FpvDroneState s, sn;
    float4 a, an;
    if (i < n) { a = B.action[i]; ld_drone(B.state, B.ld, i, s); }
    while (i < n) {
        const int64_t inext = i + stride;
        if (inext < n) { an = B.action[inext]; ld_drone(B.state, B.ld, inext, sn); }
        float reward; bool done;
        if (PASS) {
            s.px += a.x * 1e-9f; s.py += a.y * 1e-9f; s.pz += a.z * 1e-9f; s.thrust += a.w * 1e-9f;
            reward = s.vx; done = s.vy > 1e30f;
        } else {
            FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
            reward = o.reward; done = o.done;
        }
        st_drone(B.state, B.ld, i, s);
        B.reward[i] = reward; B.done[i] = done ? 1 : 0;
        s = sn; a = an; i = inext;
    }
}

// V3: plain grid-stride loop (no explicit prefetch), G blocks
template <bool PASS>
__global__ __launch_bounds__(256) void k_gridstride(const FpvK K, const Buf B, const int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        FpvDroneState s;
        float4 a = B.action[i];
        ld_drone(B.state, B.ld, i, s);
        float reward; bool done;
        if (PASS) {
            s.px += a.x * 1e-9f; s.py += a.y * 1e-9f; s.pz += a.z * 1e-9f; s.thrust += a.w * 1e-9f;
            reward = s.vx; done = s.vy > 1e30f;
        } else {
            FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
            reward = o.reward; done = o.done;
        }
        st_drone(B.state, B.ld, i, s);
        B.reward[i] = reward; B.done[i] = done ? 1 : 0;
    }
}


// V4: generalised vector shape: V consecutive drones per lane (V = 1, 2, 4), block BS, nt hints on the
// streamed operands (action in; reward/done out)
template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float __attribute__((ext_vector_type(2))) T; };
template <> struct VecT<4> { typedef float __attribute__((ext_vector_type(4))) T; };

template <int V, int BS, bool NT, bool NTS, int MINW>
__global__ __launch_bounds__(BS, MINW) void k_vecg(const FpvK K, const Buf B, const int64_t n)
{
    typedef typename VecT<V>::T vt;
    const int64_t i0 = ((int64_t)blockIdx.x * BS + threadIdx.x) * V;
    if (i0 >= n) return;   // n % V == 0 required
    vt r[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) r[k] = *reinterpret_cast<const vt*>(&B.state[(int64_t)k * B.ld + i0]);
    v4f a[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const v4f* ap = reinterpret_cast<const v4f*>(&B.action[i0 + j]);
        a[j] = NT ? __builtin_nontemporal_load(ap) : *ap;
    }
    float rew[V]; uint8_t dn[V];
    float* f = reinterpret_cast<float*>(r);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        FpvDroneState s;
        s.px = f[0 * V + j]; s.py = f[1 * V + j]; s.pz = f[2 * V + j]; s.vx = f[3 * V + j]; s.vy = f[4 * V + j]; s.vz = f[5 * V + j];
        s.q.w = f[6 * V + j]; s.q.x = f[7 * V + j]; s.q.y = f[8 * V + j]; s.q.z = f[9 * V + j];
        s.rx = f[10 * V + j]; s.ry = f[11 * V + j]; s.rz = f[12 * V + j]; s.thrust = f[13 * V + j];
        FpvStepOut o = fpv_drone_step_lane<false>(K, s, a[j].x, a[j].y, a[j].z, a[j].w, B.wx, B.wy, B.wz);
        rew[j] = o.reward; dn[j] = o.done;
        f[0 * V + j] = s.px; f[1 * V + j] = s.py; f[2 * V + j] = s.pz; f[3 * V + j] = s.vx; f[4 * V + j] = s.vy; f[5 * V + j] = s.vz;
        f[6 * V + j] = s.q.w; f[7 * V + j] = s.q.x; f[8 * V + j] = s.q.y; f[9 * V + j] = s.q.z;
        f[10 * V + j] = s.rx; f[11 * V + j] = s.ry; f[12 * V + j] = s.rz; f[13 * V + j] = s.thrust;
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        vt* dp = reinterpret_cast<vt*>(&B.state[(int64_t)k * B.ld + i0]);
        if (NTS) __builtin_nontemporal_store(r[k], dp); else *dp = r[k];
    }
    vt rv; float* rf = reinterpret_cast<float*>(&rv);
#pragma unroll
    for (int j = 0; j < V; ++j) rf[j] = rew[j];
    if (NT) __builtin_nontemporal_store(rv, reinterpret_cast<vt*>(&B.reward[i0])); else *reinterpret_cast<vt*>(&B.reward[i0]) = rv;
    if (V == 4) {
        const uint32_t w = dn[0] | (dn[1] << 8) | (dn[V > 2 ? 2 : 0] << 16) | (dn[V > 3 ? 3 : 0] << 24);
        if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t*>(&B.done[i0])); else *reinterpret_cast<uint32_t*>(&B.done[i0]) = w;
    } else if (V == 2) {
        const uint16_t w = dn[0] | (dn[V > 1 ? 1 : 0] << 8);
        if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint16_t*>(&B.done[i0])); else *reinterpret_cast<uint16_t*>(&B.done[i0]) = w;
    } else {
        if (NT) __builtin_nontemporal_store(dn[0], &B.done[i0]); else B.done[i0] = dn[0];
    }
}

#define VG(id, V, BS, NT, NTS, W) case id: hipLaunchKernelGGL((k_vecg<V, BS, NT, NTS, W>), dim3((unsigned)((n / V + BS - 1) / BS)), dim3(BS), 0, s, K, B, n); break;


// V5: AoS [n][16] state rows (p3 v3 q4 rates3 T, reward, done) moved as ONE float4 stream per wave
// instruction (64 lanes x 16 B contiguous) and transposed through LDS (pitch 17) so that each lane
// owns one drone in registers: 64 B read + 64 B write + 16 B action, no separate reward/done stores.
template <int BS, bool PASS>
__global__ __launch_bounds__(BS) void k_aos_lds(const FpvK K, float* __restrict__ st16, const float4* __restrict__ action, const int64_t n,
                                                float wx, float wy, float wz)
{
    constexpr int P = 17;
    __shared__ float tile[BS / 64][64 * P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave_first = ((int64_t)blockIdx.x * BS + wave * 64);       // first drone of this wave
    if (wave_first >= n) return;                                            // n % 64 == 0 assumed in this experiment
    float4* g = reinterpret_cast<float4*>(st16) + wave_first * 4;
    const float4 a = action[wave_first + lane];
    float* t = tile[wave];
    float4 r0 = g[lane], r1 = g[64 + lane], r2 = g[128 + lane], r3 = g[192 + lane];
    {   // float4 index f = j*64+lane -> drone f>>2, column (f&3)*4
        float4 rr[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int f = j * 64 + lane; float* d = &t[(f >> 2) * P + (f & 3) * 4]; d[0] = rr[j].x; d[1] = rr[j].y; d[2] = rr[j].z; d[3] = rr[j].w; }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float* row = &t[lane * P];
    FpvDroneState s;
    s.px = row[0]; s.py = row[1]; s.pz = row[2]; s.vx = row[3]; s.vy = row[4]; s.vz = row[5];
    s.q.w = row[6]; s.q.x = row[7]; s.q.y = row[8]; s.q.z = row[9]; s.rx = row[10]; s.ry = row[11]; s.rz = row[12]; s.thrust = row[13];
    float reward; bool done;
    if (PASS) { s.px += a.x * 1e-9f; s.py += a.y * 1e-9f; s.pz += a.z * 1e-9f; s.thrust += a.w * 1e-9f; reward = s.vx; done = s.vy > 1e30f; }
    else { FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, wx, wy, wz); reward = o.reward; done = o.done; }
    row[0] = s.px; row[1] = s.py; row[2] = s.pz; row[3] = s.vx; row[4] = s.vy; row[5] = s.vz;
    row[6] = s.q.w; row[7] = s.q.x; row[8] = s.q.y; row[9] = s.q.z; row[10] = s.rx; row[11] = s.ry; row[12] = s.rz; row[13] = s.thrust;
    row[14] = reward; row[15] = done ? 1.0f : 0.0f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = j * 64 + lane; const float* d = &t[(f >> 2) * P + (f & 3) * 4]; g[f] = make_float4(d[0], d[1], d[2], d[3]); }
}


// V6: phase de-correlation experiments on the product shape (BS=128, nt): odd blocks sleep SLEEP*64
// cycles before their loads; LDSB bytes of dummy LDS cap the resident blocks per CU (more "rounds").
template <int SLEEP, int LDSB>
__global__ __launch_bounds__(128) void k_stagger(const FpvK K, const Buf B, const int64_t n)
{
    __shared__ float pad[LDSB / 4 + 1];
    if (LDSB > 0 && threadIdx.x == 0 && B.wx == 12345.f) pad[0] = 1.f;          // keep the allocation alive
    if (SLEEP > 0 && (blockIdx.x & 1)) {
#pragma unroll
        for (int k = 0; k < SLEEP / 100; ++k) __builtin_amdgcn_s_sleep(100);
        if (SLEEP % 100) __builtin_amdgcn_s_sleep(SLEEP % 100);
    }
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    FpvDroneState s;
    float4 a = nt_load4(&B.action[i]);
    ld_drone(B.state, B.ld, i, s);
    FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
    st_drone(B.state, B.ld, i, s);
    __builtin_nontemporal_store(o.reward, &B.reward[i]); __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), &B.done[i]);
    if (LDSB > 0 && B.wx == 54321.f) B.reward[i] = pad[0];
}
// V6: "grouped" SoA: the 14 state floats of a drone in FOUR group rows - three float4 rows (px py pz vx) (vy vz qw qx)
// (qy qz rx ry) and one float2 row (rz thrust) - so that a lane moves its drone with 4 loads + 4 stores of 16 / 8
// bytes (each wave instruction 1 KiB / 512 B contiguous) instead of 14 + 14 dword accesses: the same bytes in 7
// memory streams instead of 17.  Group g starts at state + 4 g ld floats.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int BS>
__global__ __launch_bounds__(BS) void k_grp(const FpvK K, const Buf B, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    v4f* g0 = reinterpret_cast<v4f*>(B.state) + i;
    v4f* g1 = reinterpret_cast<v4f*>(B.state + 4 * B.ld) + i;
    v4f* g2 = reinterpret_cast<v4f*>(B.state + 8 * B.ld) + i;
    v2f* g3 = reinterpret_cast<v2f*>(B.state + 12 * B.ld) + i;
    const float4 a = nt_load4(&B.action[i]);
    const v4f r0 = *g0, r1 = *g1, r2 = *g2;
    const v2f r3 = *g3;
    __builtin_amdgcn_sched_barrier(0);
    FpvDroneState s;
    s.px = r0.x; s.py = r0.y; s.pz = r0.z; s.vx = r0.w; s.vy = r1.x; s.vz = r1.y; s.q.w = r1.z; s.q.x = r1.w;
    s.q.y = r2.x; s.q.z = r2.y; s.rx = r2.z; s.ry = r2.w; s.rz = r3.x; s.thrust = r3.y;
    const FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
    *g0 = v4f{s.px, s.py, s.pz, s.vx}; *g1 = v4f{s.vy, s.vz, s.q.w, s.q.x}; *g2 = v4f{s.q.y, s.q.z, s.rx, s.ry}; *g3 = v2f{s.rz, s.thrust};
    __builtin_nontemporal_store(o.reward, &B.reward[i]);
    __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), &B.done[i]);
}

// V7: first-generation ramp.  The dispatcher starts every resident wave of the first generation at once: their loads
// queue up together (a read-only phase, nothing to write yet) and complete together.  Here block b of the first G
// blocks sleeps b * T / G sleep units (64 clocks each) before its loads - waves that would otherwise wait in the
// memory queue wait in s_sleep instead - so completions, and with them the first stores and the second generation's
// loads, spread over the ramp.  Blocks >= G (later generations) are untouched.
template <int G, int T>
__global__ __launch_bounds__(128) void k_ramp(const FpvK K, const Buf B, const int64_t n)
{
    if (blockIdx.x < (unsigned)G) {
        const int units = (int)(blockIdx.x * (unsigned)T / (unsigned)G);
        for (int k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(1);
    }
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    FpvDroneState s;
    float4 a = nt_load4(&B.action[i]);
    ld_drone(B.state, B.ld, i, s);
    FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
    st_drone(B.state, B.ld, i, s);
    __builtin_nontemporal_store(o.reward, &B.reward[i]); __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), &B.done[i]);
}
// V8: issue priority.  The six resident waves of a SIMD receive their data at about the same time and then share the
// vector ALU, so in the first generation each finishes its 231 instructions six times later than it could and no
// store leaves before ~2 us.  MODE 1: four priority classes by (block + wave) so that a SIMD's waves finish one
// after the other; MODE 2: by block only; MODE 3: everything at priority 3 (control: no relative difference).
template <int MODE>
__global__ __launch_bounds__(128) void k_prio(const FpvK K, const Buf B, const int64_t n)
{
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    FpvDroneState s;
    float4 a = nt_load4(&B.action[i]);
    ld_drone(B.state, B.ld, i, s);
    const unsigned cls = MODE == 1 ? ((blockIdx.x + (threadIdx.x >> 6)) & 3u) : MODE == 2 ? (blockIdx.x & 3u) : 3u;
    switch (cls) {                      // s_setprio takes an immediate
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
    FpvStepOut o = fpv_drone_step_lane<false>(K, s, a.x, a.y, a.z, a.w, B.wx, B.wy, B.wz);
    st_drone(B.state, B.ld, i, s);
    __builtin_nontemporal_store(o.reward, &B.reward[i]); __builtin_nontemporal_store((uint8_t)(o.done ? 1 : 0), &B.done[i]);
}
#define PRI(id, M) case id: hipLaunchKernelGGL((k_prio<M>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, K, B, n); break;

#define RMP(id, G, T) case id: hipLaunchKernelGGL((k_ramp<G, T>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, K, B, n); break;

#define STG(id, SL, LB) case id: hipLaunchKernelGGL((k_stagger<SL, LB>), G(128), dim3(128), 0, s, K, B, n); break;

extern "C" int exp_step(const fpv_params_t* P, float* state, int64_t ld, const float* action, float* reward,
                        uint8_t* done, int64_t n, int variant, int grid_blocks, void* stream)
{
    FpvK K; const char* why;
    if (fpv_derive_constants(P, &K, &why) != 0) return -1;
    Buf B{state, ld, reinterpret_cast<const float4*>(action), reward, done, 0.f, 0.f, 0.f};
    hipStream_t s = (hipStream_t)stream;
    auto G = [&](int bs) { return dim3((unsigned)((n + bs - 1) / bs)); };
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_base<256, false, false>), G(256), dim3(256), 0, s, K, B, n); break;
        case 1: hipLaunchKernelGGL((k_base<256, true, false>), G(256), dim3(256), 0, s, K, B, n); break;
        case 2: hipLaunchKernelGGL((k_base<64, false, false>), G(64), dim3(64), 0, s, K, B, n); break;
        case 3: hipLaunchKernelGGL((k_base<128, false, false>), G(128), dim3(128), 0, s, K, B, n); break;
        case 4: hipLaunchKernelGGL((k_base<512, false, false>), G(512), dim3(512), 0, s, K, B, n); break;
        case 5: hipLaunchKernelGGL((k_base<1024, false, false>), G(1024), dim3(1024), 0, s, K, B, n); break;
        case 6: hipLaunchKernelGGL((k_base<256, false, true>), G(256), dim3(256), 0, s, K, B, n); break;
        case 7: hipLaunchKernelGGL((k_base<256, true, true>), G(256), dim3(256), 0, s, K, B, n); break;
        case 10: hipLaunchKernelGGL((k_vec4<false>), G(1024), dim3(256), 0, s, K, B, n); break;
        case 11: hipLaunchKernelGGL((k_vec4<true>), G(1024), dim3(256), 0, s, K, B, n); break;
        case 20: hipLaunchKernelGGL((k_prefetch<false>), dim3(grid_blocks), dim3(256), 0, s, K, B, n); break;
        case 21: hipLaunchKernelGGL((k_prefetch<true>), dim3(grid_blocks), dim3(256), 0, s, K, B, n); break;
        case 30: hipLaunchKernelGGL((k_gridstride<false>), dim3(grid_blocks), dim3(256), 0, s, K, B, n); break;
        case 31: hipLaunchKernelGGL((k_gridstride<true>), dim3(grid_blocks), dim3(256), 0, s, K, B, n); break;
        VG(100, 1, 128, false, false, 1) VG(101, 1, 128, true, false, 1) VG(102, 1, 256, true, false, 1) VG(103, 1, 64, true, false, 1)
        VG(110, 2, 64, false, false, 1) VG(111, 2, 128, false, false, 1) VG(112, 2, 256, false, false, 1) VG(113, 2, 128, true, false, 1) VG(114, 2, 64, true, false, 1)
        VG(120, 4, 64, false, false, 1) VG(121, 4, 128, false, false, 1) VG(122, 4, 256, false, false, 1) VG(123, 4, 128, true, false, 1) VG(124, 4, 64, true, false, 1)
        VG(125, 4, 256, true, false, 1) VG(126, 4, 128, true, true, 1) VG(127, 4, 128, true, false, 4) VG(128, 4, 64, true, false, 4) VG(129, 4, 256, true, false, 4)
        VG(130, 2, 128, true, true, 1) VG(131, 1, 128, true, true, 1)
        case 300: hipLaunchKernelGGL((k_aos_lds<128, false>), G(128), dim3(128), 0, s, K, state, B.action, n, 0.f, 0.f, 0.f); break;
        case 301: hipLaunchKernelGGL((k_aos_lds<128, true>), G(128), dim3(128), 0, s, K, state, B.action, n, 0.f, 0.f, 0.f); break;
        case 302: hipLaunchKernelGGL((k_aos_lds<256, false>), G(256), dim3(256), 0, s, K, state, B.action, n, 0.f, 0.f, 0.f); break;
        case 303: hipLaunchKernelGGL((k_aos_lds<64, false>), G(64), dim3(64), 0, s, K, state, B.action, n, 0.f, 0.f, 0.f); break;
        case 500: hipLaunchKernelGGL((k_grp<128>), G(128), dim3(128), 0, s, K, B, n); break;
        case 501: hipLaunchKernelGGL((k_grp<256>), G(256), dim3(256), 0, s, K, B, n); break;
        case 502: hipLaunchKernelGGL((k_grp<64>), G(64), dim3(64), 0, s, K, B, n); break;
        STG(400, 0, 0) STG(401, 20, 0) STG(402, 50, 0) STG(403, 100, 0) STG(404, 200, 0) STG(405, 400, 0)
        STG(410, 0, 20480) STG(411, 0, 40960) STG(412, 0, 10240) STG(413, 50, 20480)
        PRI(700, 0) PRI(701, 1) PRI(702, 2) PRI(703, 3)
        RMP(600, 3072, 0) RMP(601, 3072, 16) RMP(602, 3072, 32) RMP(603, 3072, 48) RMP(604, 3072, 64) RMP(605, 3072, 96) RMP(606, 3072, 128)
        RMP(610, 4096, 32) RMP(611, 4096, 64) RMP(612, 4096, 96) RMP(613, 2048, 32) RMP(614, 2048, 64) RMP(615, 8192, 64) RMP(616, 8192, 128)
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// K steps, S chunks on S streams: chunk c's step t+1 only follows chunk c's step t; no global
// barrier between steps, so one chunk's tail overlaps another chunk's head.
static hipStream_t g_streams[8];
static hipEvent_t g_ev0, g_evs[8];
static bool g_init = false;
extern "C" int exp_rollout_pipelined(const fpv_params_t* P, float* state, int64_t ld, const float* actions, int64_t action_stride,
                                     float* reward, uint8_t* done, int64_t n, int k, int S, void* stream)
{
    FpvK K; const char* why;
    if (fpv_derive_constants(P, &K, &why) != 0) return -1;
    if (S < 1 || S > 8) return -2;
    if (!g_init) { for (int i = 0; i < 8; ++i) { hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking); hipEventCreateWithFlags(&g_evs[i], hipEventDisableTiming); }
                   hipEventCreateWithFlags(&g_ev0, hipEventDisableTiming); g_init = true; }
    hipStream_t s0 = (hipStream_t)stream;
    hipEventRecord(g_ev0, s0);
    const int64_t chunk = ((n / S + 127) / 128) * 128;
    for (int c = 0; c < S; ++c) hipStreamWaitEvent(g_streams[c], g_ev0, 0);
    for (int t = 0; t < k; ++t)
        for (int c = 0; c < S; ++c) {
            const int64_t lo = c * chunk, cnt = (lo + chunk <= n) ? chunk : (n - lo);
            if (cnt <= 0) continue;
            Buf B{state + lo, ld, reinterpret_cast<const float4*>(actions + t * action_stride) + lo, reward + lo, done + lo, 0.f, 0.f, 0.f};
            hipLaunchKernelGGL((k_vecg<1, 128, true, false, 1>), dim3((unsigned)((cnt + 127) / 128)), dim3(128), 0, g_streams[c], K, B, cnt);
        }
    for (int c = 0; c < S; ++c) { hipEventRecord(g_evs[c], g_streams[c]); hipStreamWaitEvent(s0, g_evs[c], 0); }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
