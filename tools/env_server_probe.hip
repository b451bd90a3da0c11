// Feasibility probe (DESIGN 10 "not built"): a register-resident closed-loop ENV SERVER.
//
// One persistent kernel holds D drones per lane in registers for the whole episode batch.  Per step it
//   waits for a doorbell word (`bell >= t + 1`: the sticks of step t are in the action buffer),
//   reads its 16-byte action rows, advances its drones with the SAME lane function as the shipped kernels
//   (fpv_drone_step_lane, fpv_math.h -> bit-identical states), writes the 13 observation rows + reward + done,
//   and signals completion: the last wave to arrive publishes `ready = t + 1`.
// The state never leaves the registers: 73 bytes per env-step cross HBM instead of 133, and there is no launch between steps.
// A policy runs beside it on another stream as ordinary kernels, gated per step by hipStreamWaitValue32(ready >= t) /
// hipStreamWriteValue32(bell = t + 1) or by a one-wave gate kernel (tools/env_server_ab.py).
//
// EXIT CONDITIONS every wave reaches (a kernel here must never outlive its limit): each wait polls at most `spin_cap` times AND
// at most `wait_cap_ticks` of the 100 MHz wall clock, and reads a global abort word; whoever times out sets the abort word,
// every other wave sees it at its next poll, and all leave through the same epilogue (final state stored, `ready` set to
// 0x7fffffff so that every stream-side wait is released too).  The host refuses a grid that cannot be fully resident
// (occupancy query, with a margin) - a wave that is not resident can never arrive at the step barrier.
//
// NOT a product path: built only by tools/env_server_ab.py --build into tools/_variants/.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../include/fpv_abi.h"
#include "../fpyv_amd/csrc/fpv_derive.h"

#define SRV_OK 0
// SRV_BYPASS = 1: the bytes handed between the two persistent kernels travel with write-through stores (sc0 sc1) and sc1
// loads, drained with s_waitcnt before the counter - no cache write-back / invalidate per wave (with the fences of the
// default build every wave's release writes the whole L2 back and every acquire invalidates it: ~0.1 us each, thousands per step)
#ifndef SRV_BYPASS
#define SRV_BYPASS 0
#endif
#if SRV_BYPASS
#define X_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define X_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define X_RELEASE() __builtin_amdgcn_s_waitcnt(0)
#define X_ACQUIRE() ((void)0)
#define X_ORDER __ATOMIC_RELAXED
#else
#define X_ST(p, v) (*(p) = (v))
#define X_LD(p) (*(p))
#define X_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define X_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define X_ORDER __ATOMIC_ACQ_REL
#endif
struct SrvArgs {
    FpvK K;
    float* state;            // [14][ld]: rows 0..12 are rewritten every step (the policy's observation), all 14 at the end
    int64_t ld;
    const float4* action;    // [n]
    float* reward; uint8_t* done;
    uint32_t* bell;          // sticks of step t are ready when *bell >= t + 1
    uint32_t* ready;         // observation after step t is complete when *ready >= t + 1
    uint32_t* arrive;        // waves that finished the current step
    uint32_t* abort_word;    // != 0: leave now
    uint32_t n, steps, waves;
    uint32_t spin_cap; uint64_t wait_cap_ticks;
    uint32_t poll_sleep;     // 0: s_sleep 1 (64 cycles) between polls, 1: s_sleep 8, 2: s_sleep 32
    float wx, wy, wz;
};

__device__ __forceinline__ bool srv_wait_ge(const uint32_t* word, uint32_t want, const SrvArgs& A)
{
    // lane 0 polls with RELAXED loads (an acquire in the loop would invalidate the caches of whatever runs beside this kernel on
    // every poll - the first version of this probe did, and the policy's kernels ran 8x slower); one agent-scope acquire
    // fence after the word has the value: the loads that follow see what the signaller wrote.  Everyone gets the verdict.
    int ok = 0;
    if (threadIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        for (uint32_t k = 0; k < A.spin_cap; ++k) {
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = 1; break; }
            if ((k & 15u) == 15u) {
                if (__hip_atomic_load(A.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > A.wait_cap_ticks) break;
            }
            if (A.poll_sleep >= 2) __builtin_amdgcn_s_sleep(32); else if (A.poll_sleep == 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(A.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    X_ACQUIRE();
    return ok != 0;
}

template <int D>
__global__ __launch_bounds__(64) void fpv_env_server_kernel(SrvArgs A)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    FpvDroneState s[D];
    uint32_t idx[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        idx[d] = (wave * D + d) * 64u + lane;                  // consecutive lanes -> consecutive drones: coalesced rows
        const uint32_t i = idx[d] < A.n ? idx[d] : A.n - 1;
        const float* st = A.state;
        s[d].px = st[0 * A.ld + i]; s[d].py = st[1 * A.ld + i]; s[d].pz = st[2 * A.ld + i];
        s[d].vx = st[3 * A.ld + i]; s[d].vy = st[4 * A.ld + i]; s[d].vz = st[5 * A.ld + i];
        s[d].q.w = st[6 * A.ld + i]; s[d].q.x = st[7 * A.ld + i]; s[d].q.y = st[8 * A.ld + i]; s[d].q.z = st[9 * A.ld + i];
        s[d].rx = st[10 * A.ld + i]; s[d].ry = st[11 * A.ld + i]; s[d].rz = st[12 * A.ld + i]; s[d].thrust = st[13 * A.ld + i];
    }
    uint32_t t = 0;
    for (; t < A.steps; ++t) {
        if (!srv_wait_ge(A.bell, t + 1u, A)) break;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (idx[d] < A.n) {
                const uint32_t i = idx[d];
                const float* ap_ = reinterpret_cast<const float*>(A.action + i);
                const float4 a = make_float4(X_LD(ap_), X_LD(ap_ + 1), X_LD(ap_ + 2), X_LD(ap_ + 3));
                FpvStepOut o = fpv_drone_step_lane<false, true>(A.K, s[d], a.x, a.y, a.z, a.w, A.wx, A.wy, A.wz);
                if ((A.K.flags & FPV_FLAG_AUTO_RESET) && o.done) fpv_drone_reset_lane(A.K, s[d]);
                float* st = A.state;
                X_ST(st + 0 * A.ld + i, s[d].px); X_ST(st + 1 * A.ld + i, s[d].py); X_ST(st + 2 * A.ld + i, s[d].pz);
                X_ST(st + 3 * A.ld + i, s[d].vx); X_ST(st + 4 * A.ld + i, s[d].vy); X_ST(st + 5 * A.ld + i, s[d].vz);
                X_ST(st + 6 * A.ld + i, s[d].q.w); X_ST(st + 7 * A.ld + i, s[d].q.x); X_ST(st + 8 * A.ld + i, s[d].q.y); X_ST(st + 9 * A.ld + i, s[d].q.z);
                X_ST(st + 10 * A.ld + i, s[d].rx); X_ST(st + 11 * A.ld + i, s[d].ry); X_ST(st + 12 * A.ld + i, s[d].rz);
                A.reward[i] = o.reward; A.done[i] = o.done ? 1 : 0;
            }
        }
        // this wave's rows are written back before it counts itself in (agent-scope release), the last one publishes the step
        X_RELEASE();
        if (lane == 0) {
            const uint32_t before = __hip_atomic_fetch_add(A.arrive, 1u, X_ORDER, __HIP_MEMORY_SCOPE_AGENT);
            if (before == A.waves - 1u) {
                __hip_atomic_store(A.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(A.ready, t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // epilogue (normal end or abort): the whole state goes back, and every stream-side wait is released
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (idx[d] < A.n) {
            const uint32_t i = idx[d];
            float* st = A.state;
            st[0 * A.ld + i] = s[d].px; st[1 * A.ld + i] = s[d].py; st[2 * A.ld + i] = s[d].pz;
            st[3 * A.ld + i] = s[d].vx; st[4 * A.ld + i] = s[d].vy; st[5 * A.ld + i] = s[d].vz;
            st[6 * A.ld + i] = s[d].q.w; st[7 * A.ld + i] = s[d].q.x; st[8 * A.ld + i] = s[d].q.y; st[9 * A.ld + i] = s[d].q.z;
            st[10 * A.ld + i] = s[d].rx; st[11 * A.ld + i] = s[d].ry; st[12 * A.ld + i] = s[d].rz; st[13 * A.ld + i] = s[d].thrust;
        }
    }
    if (t < A.steps && lane == 0) __hip_atomic_store(A.ready, 0x7fffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a one-wave gate: leaves as soon as *word >= want (or at the caps): what a stream without stream memory operations
// puts in front of the policy's kernels
__global__ __launch_bounds__(64) void fpv_env_gate_kernel(const uint32_t* word, uint32_t want, uint32_t* abort_word, uint32_t spin_cap, uint64_t cap_ticks)
{
    if (threadIdx.x != 0) return;
    const uint64_t t0 = wall_clock64();
    for (uint32_t k = 0; k < spin_cap; ++k) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return; }
        if ((k & 15u) == 15u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
            if (wall_clock64() - t0 > cap_ticks) break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(abort_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void fpv_env_set_word_kernel(uint32_t* word, uint32_t v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- a persistent POLICY beside the server (what a fused policy megakernel would be): linear 13 -> 4 + tanh + bias per drone,
// the same arithmetic as fpv_env_policy_once_kernel below (the per-step launch of the shipped closed loop's stand-in)
struct PolArgs {
    const float* state; int64_t ld; float4* action;
    float W[13][4]; float bias[4];
    uint32_t* bell; uint32_t* ready; uint32_t* arrive; uint32_t* abort_word;
    uint32_t n, steps, waves, per_lane;
    uint32_t spin_cap; uint64_t wait_cap_ticks;
};

__device__ __forceinline__ float4 pol_eval(const PolArgs& P, uint32_t i)
{
    float o[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) o[k] = X_LD(P.state + k * P.ld + i);
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 13; ++k) acc = fmaf(o[k], P.W[k][j], acc);
        h[j] = tanhf(acc) + P.bias[j];
    }
    return make_float4(h[0], h[1], h[2], h[3]);
}

__global__ __launch_bounds__(64) void fpv_env_policy_persistent_kernel(PolArgs P)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    for (uint32_t t = 0; t < P.steps; ++t) {
        if (t) {
            int ok = 0;
            if (lane == 0) {
                const uint64_t t0 = wall_clock64();
                for (uint32_t k = 0; k < P.spin_cap; ++k) {
                    if (__hip_atomic_load(P.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= t) { ok = 1; break; }
                    if ((k & 15u) == 15u) {
                        if (__hip_atomic_load(P.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                        if (wall_clock64() - t0 > P.wait_cap_ticks) break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (!ok) __hip_atomic_store(P.abort_word, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            X_ACQUIRE();
            if (!ok) return;
        }
        for (uint32_t d = 0; d < P.per_lane; ++d) {
            const uint32_t i = (wave * P.per_lane + d) * 64u + lane;
            if (i < P.n) {
                const float4 v = pol_eval(P, i);
                float* ap_ = reinterpret_cast<float*>(P.action + i);
                X_ST(ap_, v.x); X_ST(ap_ + 1, v.y); X_ST(ap_ + 2, v.z); X_ST(ap_ + 3, v.w);
            }
        }
        X_RELEASE();
        if (lane == 0) {
            const uint32_t before = __hip_atomic_fetch_add(P.arrive, 1u, X_ORDER, __HIP_MEMORY_SCOPE_AGENT);
            if (before == P.waves - 1u) {
                __hip_atomic_store(P.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(P.bell, t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(128) void fpv_env_policy_once_kernel(PolArgs P)
{
    const uint32_t i = blockIdx.x * 128u + threadIdx.x;
    if (i < P.n) P.action[i] = pol_eval(P, i);
}

static char g_err[256] = "";
extern "C" const char* srv_last_error(void) { return g_err; }
#define SRV_FAIL(...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return -1; } while (0)

template <int D> static int launch_d(const SrvArgs& A, hipStream_t stream, int* waves_per_cu, int* resident_limit)
{
    int dev = 0, cus = 0, blocks = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) SRV_FAIL("device query failed");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fpv_env_server_kernel<D>, 64, 0) != hipSuccess) SRV_FAIL("occupancy query failed");
    // all waves must be resident at once (a wave that is not can never arrive at the step barrier).  These kernels are
    // VGPR-bound at 1 - 4 waves per SIMD (the occupancy query's known one-block overshoot concerns SGPR-bound kernels); if
    // the query is wrong anyway, the step barrier never completes, every wait runs into its cap and the kernel leaves
    const int limit = blocks * cus;
    *waves_per_cu = (int)((A.waves + cus - 1) / cus); *resident_limit = limit;
    if ((int64_t)A.waves > (int64_t)limit) SRV_FAIL("grid of %u waves cannot be fully resident with a margin (%d blocks per CU x %d CUs by the occupancy query): use more drones per lane or fewer drones", A.waves, blocks, cus);
    hipLaunchKernelGGL(fpv_env_server_kernel<D>, dim3(A.waves), dim3(64), 0, stream, A);
    if (hipGetLastError() != hipSuccess) SRV_FAIL("launch failed");
    return 0;
}

// words: device pointers bell, ready (8-byte signal memory for the stream-operation gate, or plain device words), arrive, abort
extern "C" int srv_launch(const fpv_params_t* params, int64_t n, int drones_per_lane, float* state, int64_t ld, const float* action,
                          float* reward, uint8_t* done, uint32_t* bell, uint32_t* ready, uint32_t* arrive, uint32_t* abort_word,
                          int steps, double wait_cap_ms, const float* wind, void* stream, int* waves_per_cu, int* resident_limit, int poll_sleep)
{
    SrvArgs A;
    const char* why = "";
    if (fpv_derive_constants(params, &A.K, &why) != FPV_OK) SRV_FAIL("parameters: %s", why);
    if (params->mode != FPV_MODE_DRONE || (params->flags & (FPV_FLAG_FP16_STATE | FPV_FLAG_STICK_NOISE))) SRV_FAIL("the probe serves fp32 drones with caller-supplied sticks");
    if (n <= 0 || n > (1 << 28) || ld < n || steps <= 0 || steps > 100000) SRV_FAIL("bad sizes");
    A.state = state; A.ld = ld; A.action = reinterpret_cast<const float4*>(action); A.reward = reward; A.done = done;
    A.bell = bell; A.ready = ready; A.arrive = arrive; A.abort_word = abort_word;
    A.n = (uint32_t)n; A.steps = (uint32_t)steps;
    A.waves = (uint32_t)((n + 64 * drones_per_lane - 1) / (64 * drones_per_lane));
    A.spin_cap = 1u << 22;                                     // ~4 M polls: far beyond any wait that the wall-clock cap allows
    A.wait_cap_ticks = (uint64_t)(wait_cap_ms * 1e5);          // wall_clock64 ticks at 100 MHz
    A.poll_sleep = (uint32_t)poll_sleep;
    A.wx = wind ? wind[0] : 0.f; A.wy = wind ? wind[1] : 0.f; A.wz = wind ? wind[2] : 0.f;
    switch (drones_per_lane) {
        case 1: return launch_d<1>(A, (hipStream_t)stream, waves_per_cu, resident_limit);
        case 2: return launch_d<2>(A, (hipStream_t)stream, waves_per_cu, resident_limit);
        case 4: return launch_d<4>(A, (hipStream_t)stream, waves_per_cu, resident_limit);
        case 8: return launch_d<8>(A, (hipStream_t)stream, waves_per_cu, resident_limit);
        default: SRV_FAIL("drones_per_lane is 1, 2, 4 or 8");
    }
}

extern "C" int srv_gate(const uint32_t* word, uint32_t want, uint32_t* abort_word, double cap_ms, void* stream)
{
    hipLaunchKernelGGL(fpv_env_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, want, abort_word, 1u << 22, (uint64_t)(cap_ms * 1e5));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int srv_set_word(uint32_t* word, uint32_t v, void* stream)
{
    hipLaunchKernelGGL(fpv_env_set_word_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, v);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int srv_stream_wait_ge(void* stream, uint32_t* word, uint32_t want)
{
    return (int)hipStreamWaitValue32((hipStream_t)stream, word, want, hipStreamWaitValueGte, 0xffffffffu);
}

extern "C" int srv_stream_write(void* stream, uint32_t* word, uint32_t v)
{
    return (int)hipStreamWriteValue32((hipStream_t)stream, word, v, 0);
}

extern "C" int srv_signal_alloc(void** p)
{
    return (int)hipExtMallocWithFlags(p, 8, hipMallocSignalMemory);
}

extern "C" int srv_can_stream_wait(void)
{
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) return 0;
    return v;
}

static void pol_fill(PolArgs& P, const float* state, int64_t ld, float* action, const float* W, const float* bias, int64_t n)
{
    P.state = state; P.ld = ld; P.action = reinterpret_cast<float4*>(action); P.n = (uint32_t)n;
    for (int k = 0; k < 13; ++k) for (int j = 0; j < 4; ++j) P.W[k][j] = W[k * 4 + j];
    for (int j = 0; j < 4; ++j) P.bias[j] = bias[j];
}

// W, bias: HOST pointers ([13][4], [4])
extern "C" int srv_policy_once(const float* state, int64_t ld, float* action, const float* W, const float* bias, int64_t n, void* stream)
{
    PolArgs P = {};
    pol_fill(P, state, ld, action, W, bias, n);
    hipLaunchKernelGGL(fpv_env_policy_once_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int srv_policy_persistent(const float* state, int64_t ld, float* action, const float* W, const float* bias, int64_t n, int waves, int steps,
                                     uint32_t* bell, uint32_t* ready, uint32_t* arrive, uint32_t* abort_word, double wait_cap_ms, void* stream)
{
    PolArgs P = {};
    pol_fill(P, state, ld, action, W, bias, n);
    P.bell = bell; P.ready = ready; P.arrive = arrive; P.abort_word = abort_word;
    P.steps = (uint32_t)steps; P.waves = (uint32_t)waves; P.per_lane = (uint32_t)((n + 64 * (int64_t)waves - 1) / (64 * (int64_t)waves));
    P.spin_cap = 1u << 22; P.wait_cap_ticks = (uint64_t)(wait_cap_ms * 1e5);
    int dev = 0, cus = 0, blocks = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) SRV_FAIL("device query failed");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fpv_env_policy_persistent_kernel, 64, 0) != hipSuccess) SRV_FAIL("occupancy query failed");
    if (waves > blocks * cus) SRV_FAIL("policy grid cannot be resident");
    hipLaunchKernelGGL(fpv_env_policy_persistent_kernel, dim3(waves), dim3(64), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
