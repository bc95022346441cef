// Does a wave's VALU stream slow down while its SIMD partner streams MFMAs (and vice versa)?  512-thread workgroups: waves 0-3 run role A, waves 4-7 role B.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_mfma_share valu_mfma_share.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP8(x) x x x x x x x x
// role: 0 idle, 1 = 512 x 8 independent v_fma_f32 per iteration x 32, 2 = MFMA stream (4 independent accumulators), 3 = ds_read_b128 stream
template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* o, long long* t) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? RA : RB;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f;
    half8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (threadIdx.x + i)); y[i] = (_Float16)(0.002f * i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int it = 0; it < 32; ++it)
            asm volatile(REP8(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (role == 2) {
        for (int it = 0; it < 512; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c3, 0, 0, 0);
        }
    } else if (role == 3) {
        const float* p = lds + (threadIdx.x & 63) * 4;
        float s = 0;
        for (int it = 0; it < 256; ++it) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 v0 = *(const f4*)(p + ((it * 256) & 4095)), v1 = *(const f4*)(p + ((it * 256 + 1024) & 4095));
            f4 v2 = *(const f4*)(p + ((it * 256 + 2048) & 4095)), v3 = *(const f4*)(p + ((it * 256 + 3072) & 4095));
            asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));
            s += v0.x;
        }
        a0 += s;
    }
    long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    o[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) t[wave] = t1 - t0;
}
template <int RA, int RB>
void run(const char* name, float* o, long long* t) {
    hipMemset(t, 0, 64);
    k<RA, RB><<<256, 512>>>(o, t);
    long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    auto per = [](int role, long long c) { return role == 1 ? c / (32.0 * 512.0) : role == 2 ? c / 2048.0 : role == 3 ? c / 1024.0 : 0.0; };
    printf("%-44s  A (waves 0-3): %7.2f cycles/op   B (waves 4-7): %7.2f cycles/op\n", name, per(RA, h[0]), per(RB, h[4]));
}
int main() {
    float* o; long long* t; hipMalloc(&o, 256 * 512 * 4); hipMalloc(&t, 64);
    run<1, 0>("A = VALU alone", o, t);
    run<2, 0>("A = MFMA alone", o, t);
    run<3, 0>("A = ds_read_b128 alone", o, t);
    run<1, 1>("A = VALU, B = VALU", o, t);
    run<2, 2>("A = MFMA, B = MFMA", o, t);
    run<2, 1>("A = MFMA, B = VALU", o, t);
    run<1, 2>("A = VALU, B = MFMA", o, t);
    run<2, 3>("A = MFMA, B = ds_read_b128", o, t);
    run<1, 3>("A = VALU, B = ds_read_b128", o, t);
    return 0;
}
