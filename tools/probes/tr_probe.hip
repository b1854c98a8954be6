// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (exact integer data).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short *out)
{
    __shared__ __attribute__((aligned(16))) short m[64][32];   // [row][col], value = row*100 + col
    for (int i = threadIdx.x; i < 64 * 32; i += 64) m[i / 32][i % 32] = (short)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // group g: block rows = 8*g + q, cols = 4*p .. 4*p+3
    const short *addr = &m[8 * g + q][4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
