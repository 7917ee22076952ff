// Checks that DPP wave_shr:1 on gfx950 moves lane i-1 -> lane i across all 64 lanes (lane 0 <- 0 with bound_ctrl).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) { int v = threadIdx.x + 100; o[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
int main()
{
    int *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = h[0] != 0;
    for (int i = 1; i < 64; i++) bad += h[i] != 99 + i;
    printf("wave_shr:1 %s (lane0=%d lane1=%d lane16=%d lane32=%d lane63=%d)\n", bad ? "WRONG" : "ok", h[0], h[1], h[16], h[32], h[63]);
    return bad;
}
