// Canary kernels: detect a co-resident kernel (from another stream / hardware queue) damaging this kernel's LDS or registers.
//   lds_canary : every workgroup fills `LDS_WORDS` words of LDS with a pattern and re-verifies them in a loop for `iters` rounds
//   vgpr_canary: every lane parks 96 pattern values in VGPRs and re-verifies them in a loop
// Mismatches (capped) are appended to `log` as (kind, block, index, got, want, round) and counted in log[0].
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LDS_WORDS 8192   /* 32 KB */

__device__ __forceinline__ uint32_t pat(uint32_t block, uint32_t i) { return 0xC0DE0000u ^ (block * 2654435761u) ^ (i * 40503u + 17u); }

extern "C" __global__ __launch_bounds__(256) void lds_canary_kernel(uint32_t* log, int cap, int iters) {
    __shared__ uint32_t smem[LDS_WORDS];
    for (int i = threadIdx.x; i < LDS_WORDS; i += 256) smem[i] = pat(blockIdx.x, i);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < LDS_WORDS; i += 256) {
            const uint32_t got = ((volatile uint32_t*)smem)[i], want = pat(blockIdx.x, i);
            if (got != want) {
                const uint32_t n = atomicAdd(log, 1u);
                if ((int)n < cap) {
                    uint32_t* r = log + 8 + n * 8;
                    r[0] = 1; r[1] = blockIdx.x; r[2] = i; r[3] = got; r[4] = want; r[5] = it;
                }
                ((volatile uint32_t*)smem)[i] = want;                // repair so one hit is logged once
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

extern "C" __global__ __launch_bounds__(256) void vgpr_canary_kernel(uint32_t* log, int cap, int iters) {
    uint32_t r[96];
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 96; ++k) r[k] = pat(id, k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 96; ++k) asm volatile("" : "+v"(r[k]));   // keep every value live in a VGPR across the loop
        __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int k = 0; k < 96; ++k) {
        if (r[k] != pat(id, k)) {
            const uint32_t n = atomicAdd(log, 1u);
            if ((int)n < cap) {
                uint32_t* q = log + 8 + n * 8;
                q[0] = 2; q[1] = blockIdx.x; q[2] = threadIdx.x * 256 + k; q[3] = r[k]; q[4] = pat(id, k); q[5] = iters;
            }
        }
    }
}

extern "C" int canary_run(int kind, uint32_t* log, int cap, int blocks, int iters, void* stream) {
    if (kind == 1) hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, cap, iters);
    else hipLaunchKernelGGL(vgpr_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, cap, iters);
    return (int)hipGetLastError();
}

// LDS hog: occupies 64 KB of LDS per workgroup (two per CU) and either sleeps (mode 0), streams ds_read_b128 over it (mode 1) or only
// writes it once (mode 2) — an aggressor with the igemm's LDS footprint but none of its other ingredients.
typedef float hog_f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256, 2) void lds_hog_kernel(float* out, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) float smem[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = (float)i;
    __syncthreads();
    hog_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const hog_f4 v = *(const volatile hog_f4*)(smem + ((threadIdx.x * 4 + k * 1024 + it * 64) & 16380));
                acc += v;
            }
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
extern "C" int lds_hog_run(float* out, int blocks, int iters, int mode, void* stream) {
    hipLaunchKernelGGL(lds_hog_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, mode);
    return (int)hipGetLastError();
}

// Barrier canary (kind 3): 4 waves; every round each wave publishes the round number in LDS, the workgroup barriers, every thread checks that all
// four waves have published THIS round (a barrier released early shows an old number), barriers again.  ~16 VGPRs, 64 B of LDS: co-resides with anything.
extern "C" __global__ __launch_bounds__(256) void barrier_canary_kernel(uint32_t* log, int cap, int iters) {
    __shared__ uint32_t flags[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();
    for (int it = 1; it <= iters; ++it) {
        if (lane == 0) ((volatile uint32_t*)flags)[wave] = (uint32_t)it;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t got = ((volatile uint32_t*)flags)[w];
            if (got != (uint32_t)it) {
                const uint32_t n = atomicAdd(log, 1u);
                if ((int)n < cap) { uint32_t* r = log + 8 + n * 8; r[0] = 3; r[1] = blockIdx.x; r[2] = threadIdx.x * 4 + w; r[3] = got; r[4] = it; r[5] = it; }
            }
        }
        __syncthreads();
    }
}

// VALU canary (kind 4): a short dependent integer/float chain recomputed from the same inputs every round; any round whose result differs from
// the first round's is logged.  ~24 VGPRs, no LDS.
extern "C" __global__ __launch_bounds__(256) void valu_canary_kernel(uint32_t* log, int cap, int iters) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    float ref[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it <= iters; ++it) {
        float a = 1.0f + (float)(id & 1023) * 0.001f, b = 0.5f + (float)(id >> 10) * 0.002f, c = 0.25f, d = 2.0f;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            a = fmaf(a, 0.999f, b * 0.001f); b = fmaf(b, 1.001f, -c * 0.002f); c = fmaf(c, a, d * 1e-4f); d = fmaf(d, 0.9995f, a * b * 1e-3f);
        }
        if (it == 0) { ref[0] = a; ref[1] = b; ref[2] = c; ref[3] = d; }
        else if (a != ref[0] || b != ref[1] || c != ref[2] || d != ref[3]) {
            const uint32_t n = atomicAdd(log, 1u);
            if ((int)n < cap) { uint32_t* r = log + 8 + n * 8; r[0] = 4; r[1] = blockIdx.x; r[2] = threadIdx.x; r[3] = __float_as_uint(a); r[4] = __float_as_uint(ref[0]); r[5] = it; }
        }
    }
}

extern "C" int canary_run2(int kind, uint32_t* log, int cap, int blocks, int iters, void* stream) {
    if (kind == 3) hipLaunchKernelGGL(barrier_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, cap, iters);
    else hipLaunchKernelGGL(valu_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, log, cap, iters);
    return (int)hipGetLastError();
}
