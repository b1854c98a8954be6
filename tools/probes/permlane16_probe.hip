// What v_permlane16_swap_b32 does on gfx950, checked on hardware (the conv epilogues rely on it):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane16_probe.hip -o /tmp/p16 && /tmp/p16
// Every lane passes a = 100 + lane, b = 200 + lane.  Expected output: rows 0 / 2 (lanes 0-15, 32-47) hold (own a, the NEXT row's a),
// rows 1 / 3 (lanes 16-31, 48-63) hold (the PREVIOUS row's b, own b).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;   // named lvalues: see c2_swap_rows in csrc/orn_conv2_bf16.hip
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int row = 0; row < 4; ++row)
        printf("row %d: lane %2d holds (%u, %u) .. lane %2d holds (%u, %u)\n", row, row * 16, h[row * 16], h[64 + row * 16], row * 16 + 15,
               h[row * 16 + 15], h[64 + row * 16 + 15]);
    return 0;
}
