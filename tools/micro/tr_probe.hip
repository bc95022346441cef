// What does ds_read_b64_tr_b16 return?  LDS holds halfs whose value is their own index; every lane reads at a caller-given byte address.
#include <hip/hip_runtime.h>
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void tr_probe_kernel(const int* addr, float* out) {
    __shared__ __attribute__((aligned(256))) __fp16 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (__fp16)(float)i;
    __syncthreads();
    fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)((char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
extern "C" void tr_probe_run(const int* addr, float* out, void* stream) { hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out); }
