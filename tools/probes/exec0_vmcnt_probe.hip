// Does a vector-memory instruction issued with EXEC = 0 take part in vmcnt on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/exec0_vmcnt_probe.hip -o /tmp/e0 && /tmp/e0
// One wave: a real (cold, slow) load A, then three loads under EXEC = 0, then s_waitcnt vmcnt(1).  If the masked loads count,
// that wait has to see A complete (in-order return) and takes A's latency; if they are dropped, A alone is outstanding and the
// wait falls through.  Prints cycles to pass vmcnt(1) and vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *big, unsigned long long *out, float *sink)
{
    const float *p = big + (size_t)threadIdx.x * 4096 + (size_t)blockIdx.x * 1048576;   // one cache line per lane, never touched before
    float a, b0, b1, b2;
    unsigned long long t0, t1, t2;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                 "s_memtime %0\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "global_load_dword %3, %7, off\n\t"
                 "s_mov_b64 exec, 0\n\t"
                 "global_load_dword %4, %7, off offset:64\n\t"
                 "global_load_dword %5, %7, off offset:128\n\t"
                 "global_load_dword %6, %7, off offset:192\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "s_waitcnt vmcnt(1)\n\t"
                 "s_memtime %1\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_memtime %2\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2) : "v"(p) : "memory");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t0; }
    sink[threadIdx.x] = a;                     // (b0..b2 are undefined under EXEC = 0: not used)
}
int main()
{
    float *big, *sink; unsigned long long *out, h[2];
    if (hipMalloc(&big, (size_t)64 * 1048576 * 4) != hipSuccess || hipMalloc(&out, 16) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, big + (size_t)rep * 8 * 1048576, out, sink);
        if (hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("rep %d: vmcnt(1) passed after %llu cycles, vmcnt(0) after %llu cycles\n", rep, h[0], h[1]);
    }
    return 0;
}
