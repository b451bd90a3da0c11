// fpv_exp.h - the two build-time switches left for A/B builds (tools/ab_variants.py); the shipped library is built
// with neither defined.  Every other experiment hook of rounds 1-5 is closed and gone from the sources; what each
// one measured is in profiles/HISTORY.md.
#pragma once

// drones per workgroup of the step and k-step kernels (whole wave64s; the rotation's block, an XCD's share of a row)
#ifndef FPV_EXP_BLOCK
#define FPV_EXP_BLOCK 128
#endif

// occupancy bound of the single-step kernel: -DFPV_EXP_STEP_WAVES=N (or =MIN,MAX) waves per SIMD
#ifdef FPV_EXP_STEP_WAVES
#define FPV_EXP_STEP_ATTR __attribute__((amdgpu_waves_per_eu(FPV_EXP_STEP_WAVES)))
#else
#define FPV_EXP_STEP_ATTR
#endif
