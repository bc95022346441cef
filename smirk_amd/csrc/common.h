// Shared helpers for the gfx950 kernels of libsmirk_hip.so (wave64, MFMA, LDS).  CDNA4 only: no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smirk_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SMIRK_WAVE 64

static inline size_t smirk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int smirk_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMIRK_OK : SMIRK_ERR_LAUNCH;
}

// row of a 32x32 MFMA accumulator register r (0..15) for this lane: (r&3) + 8*(r>>2) + 4*(lane>>5); column = lane&31
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
