// Microbenchmark replicas of the conv_igemm hot loop (128x128 tile, 4 waves, split-fp16: 24 MFMAs + 16 ds_read_b128 + 8 DMA per 32-k chunk),
// adding one ingredient at a time, to find which one costs the ~1000 cycles per chunk that the real kernel spends beyond its 768 MFMA cycles.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define STAGE (256 * 32)      // dwords: 128 A rows + 128 B rows of 32 dwords

// MODE bit0: LDS fragment reads, bit1: s_barrier per chunk, bit2: operand DMA (global_load_lds) per chunk, bit3: vmcnt(0) before the barrier
template <int MODE>
__global__ __launch_bounds__(256, 2) void igemm_micro(const float* __restrict__ src, float* out, int chunks, size_t wg_stride_dw, size_t region_dw) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, hb = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = (float)((i * 7) & 15) * 0.001f;
    __syncthreads();
    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) { ah[s][i][k] = (_Float16)(0.01f * (k + i)); al[s][i][k] = (_Float16)0.5f; bh[s][i][k] = (_Float16)(0.02f * k); bl[s][i][k] = (_Float16)0.25f; }
    size_t goff = ((size_t)blockIdx.x * wg_stride_dw) % region_dw;      // walks the region in 32 KB steps, wrapping: region size sets L2 / MALL / HBM residency
    const float* g = src + goff + (size_t)tid * 4;
    for (int c = 0; c < chunks; ++c) {
        const float* As = smem + (c & 1) * STAGE;
        const float* Bs = As + 128 * 32;
        if (MODE & 2) {
            if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (MODE & 4) {
            float* Ad = smem + ((c + 1) & 1) * STAGE;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                __builtin_amdgcn_global_load_lds((gptr_t)(g + p * 1024), (lptr_t)(Ad + (wave * 8 + 32 * p) * 32), 16, 0, 0);
            }
            goff += 8 * 1024; if (goff >= region_dw) goff -= region_dw;
            g = src + goff + (size_t)tid * 4;
        }
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pc = 2 * (2 * s + hb);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (wm * 2 + i) * 32 + fr;
                    ah[s][i] = *(const half8*)(As + row * 32 + ((pc ^ ((row >> 1) & 7)) << 2));
                    al[s][i] = *(const half8*)(As + row * 32 + (((pc + 1) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = (wn * 2 + j) * 32 + fr;
                    bh[s][j] = *(const half8*)(Bs + row * 32 + ((pc ^ ((row >> 1) & 7)) << 2));
                    bl[s][j] = *(const half8*)(Bs + row * 32 + (((pc + 1) ^ ((row >> 1) & 7)) << 2));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc1[i][j], 0, 0, 0);
                }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) sum += acc0[i][j][0] + acc1[i][j][5];
    out[blockIdx.x * 256 + tid] = sum;
}

extern "C" int igemm_micro_run(int mode, const float* src, float* out, int blocks, int chunks, size_t wg_stride_dw, size_t region_dw, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define CASE(M) case M: hipLaunchKernelGGL(igemm_micro<M>, dim3(blocks), dim3(256), 0, st, src, out, chunks, wg_stride_dw, region_dw); break;
    switch (mode) { CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(14) CASE(15) default: return -1; }
    return (int)hipGetLastError();
}
