// Does a hipGraph with a forked branch hide a chain of small latency-bound kernels under a big one on this stack (ROCm 7.2, gfx950)?
//   serial:  big1 -> small x N -> big2 -> tail
//   forked:  big1 -> { small x N  ||  big2 } -> tail          (side stream forked / joined with events during capture)
// Reports the replay time of both graphs (us) for a few chain lengths, with and without a high-priority side stream, and with the
// big kernel leaving LDS room for the small ones or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_big(float *out, int iters)
{
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = fmaf(a, b, 1e-3f); b = fmaf(b, 0.9999f, 1e-4f); }
    lds[threadIdx.x] = a + b;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 7) & 255];
}
__global__ void __launch_bounds__(256) k_small(float *buf, int iters)
{
    __shared__ float s[256];
    float a = buf[threadIdx.x];
    for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 1e-3f);
    s[threadIdx.x] = a;
    __syncthreads();
    buf[threadIdx.x] = s[255 - threadIdx.x];
}

static float replay_us(hipGraphExec_t ge, hipStream_t st, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main()
{
    float *out, *buf;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    hipStream_t st, side, side_hi;
    CK(hipStreamCreate(&st)); CK(hipStreamCreate(&side));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&side_hi, hipStreamNonBlocking, hi));
    CK(hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    const int big_iters = 20000, small_iters = 1500;
    for (int lds_kb : {68, 40}) {
        for (int nsmall : {0, 4, 8, 16}) {
            for (int mode = 0; mode < 3; ++mode) {        // 0 serial, 1 forked, 2 forked with a high-priority side stream
                if (nsmall == 0 && mode > 0) continue;
                hipStream_t sd = mode == 2 ? side_hi : side;
                hipGraph_t g; hipGraphExec_t ge;
                hipEvent_t fork, join;
                CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), lds_kb * 1024, st, out, big_iters);
                if (mode == 0) {
                    for (int i = 0; i < nsmall; ++i) hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, st, buf, small_iters);
                    hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), lds_kb * 1024, st, out + 4096, big_iters);
                } else {
                    CK(hipEventRecord(fork, st));
                    CK(hipStreamWaitEvent(sd, fork, 0));
                    for (int i = 0; i < nsmall; ++i) hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, sd, buf, small_iters);
                    CK(hipEventRecord(join, sd));
                    hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), lds_kb * 1024, st, out + 4096, big_iters);
                    CK(hipStreamWaitEvent(st, join, 0));
                }
                hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, st, buf + 4096, 10);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                const float us = replay_us(ge, st, 40);
                printf("big LDS %d KB, %2d small kernels, %s: %.1f us per replay\n", lds_kb, nsmall,
                       mode == 0 ? "serial" : (mode == 1 ? "forked" : "forked (high-priority side)"), us);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
        }
    }
    return 0;
}
