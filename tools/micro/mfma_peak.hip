// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate with every SIMD busy (no memory traffic): the achievable matrix-pipe peak under
// the board's power / clock management, to compare with the 2.5 PFLOP/s data-sheet figure.   hipcc --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clocks) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(threadIdx.x * 0.001f + k); b[k] = (_Float16)(0.5f - k * 0.01f); }
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = w1 - w0; }
}

extern "C" int mfma_peak_run(float* out, int blocks, int iters, int nacc, long long* clocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nacc == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, st, out, iters, clocks);
    else if (nacc == 8) hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, st, out, iters, clocks);
    else hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, st, out, iters, clocks);
    return (int)hipGetLastError();
}
