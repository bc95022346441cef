// probe: what do DPP quad_perm [1,0,3,2] and v_permlane16_swap do on gfx950?  (tools/micro; build: hipcc --offload-arch=gfx950 -O3 -o pool_probe pool_probe.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* o) {
    const int lane = threadIdx.x;
    const float y = (float)lane;
    const float h = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, true));
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, y), __builtin_bit_cast(unsigned, y + 100.f), false, false);
    o[lane] = h;
    o[64 + lane] = __builtin_bit_cast(float, r[0]);
    o[128 + lane] = __builtin_bit_cast(float, r[1]);
}
int main() {
    float* d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    float h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; ++s) { printf("%s:", s == 0 ? "dpp" : s == 1 ? "swap[0] (first = lane)" : "swap[1] (second = lane+100)"); for (int i = 0; i < 64; ++i) printf(" %g", h[s * 64 + i]); printf("\n"); }
    return 0;
}
