// Shared helpers for the gfx950 kernels of libsmirk_hip.so (wave64, MFMA, LDS).  CDNA4 only: no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smirk_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 -> split-fp16 pair (value = hi + lo * 2^-11).  The empty asm pins v as a ROUNDED fp32 register value.  Without it the compiler folds a
// preceding multiply into the conversion (v_fma_mixlo_f16: one rounding of the exact product) while the residual is still taken from the fp32-rounded
// product; at an exact tie the two disagree about which fp16 neighbour `hi` is and the pair lands a whole fp16 ulp away from v (measured: 1 element in
// ~16k off by 6e-4 relative in the BatchNorm backward, tests/test_train_ops_gpu.py).
__device__ __forceinline__ void smirk_split1(float v, _Float16& hi, _Float16& lo) {
    asm("" : "+v"(v));
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

// Two values per conversion instruction (gfx950 v_cvt_pk_f16_f32; same round-to-nearest-even as v_cvt_f16_f32, so the SAME (hi, lo) pairs as smirk_split1):
// 8 VALU per pair instead of 12.  hi / lo are returned packed (a in the low half).  Scalar arithmetic on purpose: no packed-FP32 instructions (build.py).
typedef _Float16 smirk_half2 __attribute__((ext_vector_type(2)));
typedef float smirk_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void smirk_split2(float a, float b, smirk_half2& hi, smirk_half2& lo) {
    asm("" : "+v"(a));
    asm("" : "+v"(b));
    const smirk_float2 v = {a, b};
    hi = __builtin_convertvector(v, smirk_half2);
    const float da = (a - (float)hi.x) * 2048.0f, db = (b - (float)hi.y) * 2048.0f;
    const smirk_float2 d = {da, db};
    lo = __builtin_convertvector(d, smirk_half2);
}

#define SMIRK_WAVE 64

static inline size_t smirk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int smirk_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMIRK_OK : SMIRK_ERR_LAUNCH;
}

// row of a 32x32 MFMA accumulator register r (0..15) for this lane: (r&3) + 8*(r>>2) + 4*(lane>>5); column = lane&31
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- launch profiler (smirk_profile_start / _stop in capi.hip) -------------------------------------------------------------------
// Every launch of the library goes through SMIRK_LAUNCH: with the profiler armed the launch is bracketed by two hipEvents on its stream
// and labelled with the kernel's instantiated name; dispatchers that know the launch's algorithmic work state it just before through
// smirk_prof_next().  Disarmed (the normal case) the cost is one load and a predictable branch.
extern bool g_smirk_prof_on;
void smirk_prof_begin(const char* name, hipStream_t st);
void smirk_prof_end(hipStream_t st);
// name (nullable: keep the stringified kernel), algorithmic flop and bytes of the NEXT launch made by this thread
void smirk_prof_next(const char* name, double flop, double bytes);

struct SmirkLaunchScope {
    bool on;
    hipStream_t st;
    SmirkLaunchScope(const char* name, hipStream_t s) : on(g_smirk_prof_on), st(s) { if (on) smirk_prof_begin(name, s); }
    ~SmirkLaunchScope() { if (on) smirk_prof_end(st); }
};
#define SMIRK_LAUNCH(kernel, grid, block, lds, st, ...)                                   \
    do {                                                                                  \
        SmirkLaunchScope smirk_launch_scope_(#kernel, (hipStream_t)(st));                 \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                    \
    } while (0)
