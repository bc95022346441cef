// Does `buffer_load_dwordx4 ... offen lds` write zeros to LDS for out-of-range lanes, or skip the write?
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lptr_t;
extern "C" __global__ void oob_lds_kernel(const float* src, float* out, int nbytes) {
    __shared__ float smem[256];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.0f;          // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nbytes, 0x00020000);
    const unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
extern "C" int oob_lds_run(const float* src, float* out, int nbytes, void* stream) {
    hipLaunchKernelGGL(oob_lds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, out, nbytes);
    return (int)hipGetLastError();
}
