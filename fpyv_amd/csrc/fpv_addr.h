// fpv_addr.h - lane addressing of the SoA / row buffers.
//
// Every global access of the step kernels is "uniform 64-bit base (SGPR pair) + 32-bit byte offset of
// the lane": the compiler then emits `global_load_dword v, v_off, s[base:base+1]` instead of 64-bit
// vector address arithmetic per row (that arithmetic was ~15 % of the kernel's VALU instructions).
// The widest element addressed this way is the 16-byte action row, so the offset 16*i must fit
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
