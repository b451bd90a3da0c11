// fpv_addr.h - lane addressing of the SoA / row buffers.
//
// Every global access of the step kernels is written as "uniform 64-bit row base + 32-bit byte offset of the lane"
// (row_at(), csrc/fpv_hip.hip): the lane's part of an address is ONE 32-bit product, never a 64-bit multiply.  What the
// compiler makes of it at the shipped flags (profiles/archive/r04_hot_kernel_isa.txt, checked by tests/test_isa_claims.py):
//   * the action row, the SoA sticks and the fp16 kernel's pair rows use the saddr form
//     `global_load_dword(x4) v, v_off, s[base:base+1]` - SGPR base, 32-bit VGPR offset, no vector address arithmetic;
//   * the 14 fp32 state rows of the plain kernel become 14 `v_lshl_add_u64 v[a:b], s[base], 0, v[off]` into VGPR pairs,
//     each used by `global_load_dword v, v[a:b], off` and AGAIN by the row's store at the end of the kernel (the row bases
//     advance by scalar adds of the stride).  Forcing the saddr form on the state rows too was built and measured
//     (`off32` in profiles/archive/r02_exp_state_cache_policy.log: 42 VGPRs, 8 waves per SIMD) and was no faster: rejected.
// The widest element addressed by a lane offset is the 16-byte action row, so the offset 16*i must fit
// 32 bits: a handle holds at most 2^28 drones (fpv_create refuses more; a GPU with 288 GB would hold
// ~2^31 fp32 drones, so larger populations are split over handles - they are independent).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define FPV_ADDR_HD __host__ __device__ __forceinline__
#else
#define FPV_ADDR_HD static inline
#endif

#define FPV_MAX_DRONES_LOG2 28
#define FPV_MAX_DRONES ((int64_t)1 << FPV_MAX_DRONES_LOG2)
#define FPV_MAX_ELEM_BYTES 16u       // widest element addressed by a lane offset (float4 action row)

// byte offset of element i of a row of `elem_bytes`-byte elements; exact for i < FPV_MAX_DRONES and
// elem_bytes <= FPV_MAX_ELEM_BYTES (the product stays below 2^32)
FPV_ADDR_HD uint32_t fpv_lane_offset(uint32_t i, uint32_t elem_bytes) { return i * elem_bytes; }
