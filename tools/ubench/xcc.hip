// which XCC does workgroup i of a 2-D launch run on?  (HW_REG_XCC_ID vs the linear id)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    const unsigned lid = blockIdx.x + gridDim.x * blockIdx.y;
    if (threadIdx.x == 0) out[lid] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) | (__builtin_amdgcn_s_getreg(20 | (15 << 11)) << 8);
}
int main()
{
    const unsigned gx = 110, gy = 16;
    unsigned *d, h[110 * 16];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int agree = 0, hist[16] = {0};
    for (unsigned i = 0; i < gx * gy; ++i) { hist[h[i] & 15]++; }
    printf("first 32 WGs: xcc_id (raw reg>>8 in hex)\n");
    for (unsigned i = 0; i < 32; ++i) printf("%u:%u(%x) ", i, h[i] & 15, h[i] >> 8);
    printf("\nhistogram of xcc ids:");
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    // is xcc == (lid + c) % 8 for a constant c?
    for (unsigned c = 0; c < 8; ++c) { int ok = 0; for (unsigned i = 0; i < gx * gy; ++i) ok += ((h[i] & 15) == ((i + c) & 7)); if (ok > agree) agree = ok; }
    printf("\nbest agreement with (lid + c) %% 8: %d of %u\n", agree, gx * gy);
    return 0;
}
