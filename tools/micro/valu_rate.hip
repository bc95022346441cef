// VALU issue cost on gfx950 (cycles per wave64 instruction), one wave per SIMD and two: which of the epilogue's instructions are slow?
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define BODY(INSN)                                                                         \
    long long t0 = __builtin_readcyclecounter();                                           \
    for (int it = 0; it < 64; ++it) {                                                      \
        asm volatile(REP8(REP8(INSN)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sm)); \
    }                                                                                      \
    long long t1 = __builtin_readcyclecounter();
template <int K>
__global__ void k(float* o, long long* t) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f;
    unsigned long long sm = 0x5555555555555555ull;
    long long dt;
    if (K == 0) { BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") dt = t1 - t0; }
    if (K == 1) { BODY("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0\n") dt = t1 - t0; }
    if (K == 2) { BODY("v_cvt_f32_f16_e32 %0, %1\n v_cvt_f32_f16_e32 %1, %2\n v_cvt_f32_f16_e32 %2, %3\n v_cvt_f32_f16_e32 %3, %4\n v_cvt_f32_f16_e32 %4, %5\n v_cvt_f32_f16_e32 %5, %6\n v_cvt_f32_f16_e32 %6, %7\n v_cvt_f32_f16_e32 %7, %0\n") dt = t1 - t0; }
    if (K == 3) { BODY("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n") dt = t1 - t0; }
    if (K == 4) { BODY("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n") dt = t1 - t0; }
    if (K == 5) { BODY("v_cndmask_b32_e64 %0, 0, %0, %9\n v_cndmask_b32_e64 %1, 0, %1, %9\n v_cndmask_b32_e64 %2, 0, %2, %9\n v_cndmask_b32_e64 %3, 0, %3, %9\n v_cndmask_b32_e64 %4, 0, %4, %9\n v_cndmask_b32_e64 %5, 0, %5, %9\n v_cndmask_b32_e64 %6, 0, %6, %9\n v_cndmask_b32_e64 %7, 0, %7, %9\n") dt = t1 - t0; }
    if (K == 6) { BODY("v_fmamk_f32 %0, %0, 0x3a000000, %8\n v_fmamk_f32 %1, %1, 0x3a000000, %8\n v_fmamk_f32 %2, %2, 0x3a000000, %8\n v_fmamk_f32 %3, %3, 0x3a000000, %8\n v_fmamk_f32 %4, %4, 0x3a000000, %8\n v_fmamk_f32 %5, %5, 0x3a000000, %8\n v_fmamk_f32 %6, %6, 0x3a000000, %8\n v_fmamk_f32 %7, %7, 0x3a000000, %8\n") dt = t1 - t0; }
    if (K == 7) { BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n") dt = t1 - t0; }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[K] = dt;
}
int main() {
    float* o; long long* t; hipMalloc(&o, 4 << 20); hipMalloc(&t, 64);
    const char* names[8] = {"v_fma_f32 (8 independent)", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_max_f32", "v_cndmask_b32 (sgpr mask)", "v_fmamk_f32 (literal)", "v_fma_f32 (dependent chain)"};
    for (int threads = 256; threads <= 512; threads += 256) {
        hipMemset(t, 0, 64);
        k<0><<<256, threads>>>(o, t); k<1><<<256, threads>>>(o, t); k<2><<<256, threads>>>(o, t); k<3><<<256, threads>>>(o, t);
        k<4><<<256, threads>>>(o, t); k<5><<<256, threads>>>(o, t); k<6><<<256, threads>>>(o, t); k<7><<<256, threads>>>(o, t);
        long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i) printf("%d waves/SIMD  %-30s %6.2f cycles per instruction (per wave)\n", threads / 256, names[i], (double)h[i] / (64.0 * 512.0));
    }
    return 0;
}
