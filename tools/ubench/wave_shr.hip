// Checks the DPP whole-wave shifts the RS decoder relies on (gfx950): wave_shr:1 moves lane i-1 -> lane i across all 64
// lanes (lane 0 <- 0 with bound_ctrl); with row masks, wave_shl:1 on rows 2-3 only and wave_shr:1 on rows 0-1 only leave
// the other rows untouched and zero-fill lanes 63 / 0.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
    const int v = threadIdx.x + 100;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xC, 0xf, true);
    o[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x138, 0x3, 0xf, true);
}
int main()
{
    int *d, h[192];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = h[0] != 0;
    for (int i = 1; i < 64; i++) bad += h[i] != 99 + i;
    for (int i = 0; i < 64; i++) bad += h[64 + i] != (i < 32 ? 100 + i : (i == 63 ? 0 : 101 + i));
    for (int i = 0; i < 64; i++) bad += h[128 + i] != (i >= 32 ? 100 + i : (i == 0 ? 0 : 99 + i));
    printf("DPP wave shifts %s (shr: %d %d %d %d | shl rows2-3: %d %d %d %d | shr rows0-1: %d %d %d %d)\n", bad ? "WRONG" : "ok",
           h[0], h[1], h[32], h[63], h[64 + 31], h[64 + 32], h[64 + 62], h[64 + 63], h[128], h[128 + 1], h[128 + 31], h[128 + 32]);
    return bad;
}
