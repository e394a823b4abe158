// Common definitions for the priorflow HIP library (gfx950 / CDNA4 only).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PF_HD __host__ __device__ __forceinline__
#else
// host-only build: used exclusively by tests/emu (logic validation without a GPU)
#define PF_HD inline
#endif

// The reference evaluates every sampler step as a separate rounded fp32 torch op;
// all translation units are built with -ffp-contract=off so a*b+c is never fused.

#define PF_CORR_LEVELS 4
#define PF_CORR_RADIUS 4
#define PF_TAPS 81           // (2r+1)^2
#define PF_CORR_CH 324       // levels * taps

// Error codes returned by the C-ABI on top of hipError_t values (which are > 0).
#define PF_OK 0
#define PF_ERR_BAD_ARG (-1)
#define PF_ERR_BAD_SHAPE (-2)
