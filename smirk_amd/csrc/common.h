// Shared helpers for the gfx950 kernels of libsmirk_hip.so (wave64, MFMA, LDS).  CDNA4 only: no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

// ---- split-fp16 range audit (fail loudly instead of propagating inf / NaN) ----------------------------------------------------------------------------------
// hi is an fp16: a value of magnitude >= 65520 becomes +-inf when it is stored, and every later layer turns it into inf / NaN pixels without any error (a badly
// scaled checkpoint is the case nobody can test offline).  Every site that SPLITS fp32 results into the storage format audits the eight values it is about to
// store: four v_max3_f32 on |v| + one compare + one never-taken branch per 8-channel group; the cold path stores a 1 into ONE host-pinned, device-mapped word
// (capi.hip) that the host reads — without any synchronisation — at the start of the NEXT forward call of a module (smirk_amd/_lib.py raise_if_range_tripped).
// fmaxf drops NaNs, which is fine here: a NaN inside the network is the child of an inf that was flagged where it was produced; NaNs arriving from OUTSIDE are caught
// by smirk_range_audit1 (`!(|v| < limit)` is true for NaN) in the kernels that convert images / parameters into the storage format.
// The pointer to that word is a per-translation-unit __device__ variable bound lazily by the first launch of each unit on each device (SMIRK_LAUNCH below).
static __device__ __attribute__((unused)) unsigned* smirk_range_flag_dev;
#define SMIRK_F16_RANGE_LIMIT 65520.0f
__device__ __forceinline__ void smirk_range_trip() {
    unsigned* f = smirk_range_flag_dev;
    if (f) __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void smirk_range_audit8(const float* v) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                          fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    if (__builtin_expect(!(m < SMIRK_F16_RANGE_LIMIT), 0)) smirk_range_trip();
}
__device__ __forceinline__ void smirk_range_audit4(const float* v) {
    const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    if (__builtin_expect(!(m < SMIRK_F16_RANGE_LIMIT), 0)) smirk_range_trip();
}
// The hot epilogues (MFMA convolution kernels, the encoder's fused blocks) do not branch per group: a branch in the middle of an unrolled epilogue keeps the scheduler from
// overlapping one item's LDS reads with the previous item's stores (measured, round 6: +3-7 % on the 224^2 / 112^2 generator kernels, +47 % on the register-bound
// mbconv_image instantiation).  They keep ONE running maximum per lane (v_max3_f32: 4 instructions per 8-channel group, one VGPR) and test it once, before the kernel ends.
struct SmirkRangeAcc {
    float m;
    __device__ __forceinline__ SmirkRangeAcc() : m(0.f) {}
    __device__ __forceinline__ void see8(const float* v) {
        m = fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])); m = fmaxf(fmaxf(m, fabsf(v[2])), fabsf(v[3]));
        m = fmaxf(fmaxf(m, fabsf(v[4])), fabsf(v[5])); m = fmaxf(fmaxf(m, fabsf(v[6])), fabsf(v[7]));
    }
    __device__ __forceinline__ void see1(float v) { m = fmaxf(m, fabsf(v)); }
    __device__ __forceinline__ void commit() const { if (__builtin_expect(!(m < SMIRK_F16_RANGE_LIMIT), 0)) smirk_range_trip(); }
};
// The same audit for kernels that sit AT their VGPR budget (enc1_fused, conv3x3_ring<2,pool>, mbconv_image<7,4>: one more live VGPR is a spill): the per-group maximum
// is compared at once and the wave's verdict is OR-ed into a scalar register pair — 4 v_max3 + v_cmp + s_or_b64 per group, no VGPR held, no branch.
struct SmirkRangeAccS {
    unsigned long long bad;
    __device__ __forceinline__ SmirkRangeAccS() : bad(0ull) {}
    __device__ __forceinline__ void see8(const float* v) {
        const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                              fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
        bad |= __builtin_amdgcn_ballot_w64(!(m < SMIRK_F16_RANGE_LIMIT));
    }
    __device__ __forceinline__ void see1(float v) { bad |= __builtin_amdgcn_ballot_w64(!(fabsf(v) < SMIRK_F16_RANGE_LIMIT)); }
    __device__ __forceinline__ void commit() const { if (__builtin_expect(bad != 0ull, 0)) smirk_range_trip(); }
};
__device__ __forceinline__ void smirk_range_audit1(float v) {                  // NaN-catching form (inputs from outside the library)
    if (__builtin_expect(!(fabsf(v) < SMIRK_F16_RANGE_LIMIT), 0)) smirk_range_trip();
}
unsigned* smirk_range_flag_device_ptr();                                        // capi.hip: device address of the host-pinned word (nullptr if it cannot be mapped)
static inline void smirk_range_bind_tu() {
    static std::atomic<unsigned long long> bound{0};                            // one bit per device, per translation unit
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return; }
    if ((bound.load(std::memory_order_relaxed) >> dev) & 1ull) return;
    unsigned* p = smirk_range_flag_device_ptr();
    if (p && hipMemcpyToSymbol(HIP_SYMBOL(smirk_range_flag_dev), &p, sizeof(p)) != hipSuccess) (void)hipGetLastError();   // (a unit without audited kernels has no such symbol)
    bound.fetch_or(1ull << dev, std::memory_order_relaxed);
}

static inline size_t smirk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int smirk_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SMIRK_OK : SMIRK_ERR_LAUNCH;
}

// row of a 32x32 MFMA accumulator register r (0..15) for this lane: (r&3) + 8*(r>>2) + 4*(lane>>5); column = lane&31
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- BatchNorm statistics from the MFMA accumulators (train mode) ----------------------------------------------------------------------------------------
// In the 32x32 accumulator layout a lane holds ONE output channel (column lane & 31) and 16 pixel rows (mfma32_row): the per-channel sums of a wave's
// [32 x 32] block are 16 adds + 16 FMAs on values the epilogue computes anyway (acc0 + acc1 * 2^-11) plus one exchange between lanes l and l + 32.
// `valid` = number of rows of this block that exist (tiles ragged at the end of M repeat their last row: those must not be counted).
__device__ __forceinline__ void stats_block(const float* x16, int lane, int valid, float& s1, float& s2) {
    if (valid >= 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1 += x16[r]; s2 = fmaf(x16[r], x16[r], s2); }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = mfma32_row(r, lane) < valid ? x16[r] : 0.f;
            s1 += x; s2 = fmaf(x, x, s2);
        }
    }
}
// sums of lanes l and l + 32 (the two row halves of a column), identical in both afterwards
__device__ __forceinline__ float stats_pair(float v) { return v + __shfl_xor(v, 32, 64); }

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
        smirk_range_bind_tu();                                                            \
        SmirkLaunchScope smirk_launch_scope_(#kernel, (hipStream_t)(st));                 \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                    \
    } while (0)
