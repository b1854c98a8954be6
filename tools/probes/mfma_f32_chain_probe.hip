// Cycles per dependent fp32 MFMA (one accumulation chain per wave) on gfx950: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32,
// 1 / 2 / 4 independent accumulators per wave, one wave per SIMD (256 threads per CU, 1 block).  s_memtime ticks and wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k32(float *out, unsigned long long *ticks, int n)
{
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float *out, unsigned long long *ticks, int n)
{
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <class F>
static int run(const char *name, F launch, int nacc, float *out, unsigned long long *ticks, int blocks)
{
    const int n = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(out, ticks, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(out, ticks, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t;
    CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double per = (double)t / ((double)n * 8 * nacc);
    printf("%s x%d acc, %4d blocks: %.1f s_memtime ticks per MFMA (chain: %.1f per dependent step), wall %.1f us -> %.2f ticks/ns\n", name, nacc, blocks, per,
           per * nacc, ms * 1e3, (double)t / (ms * 1e6));
    return 0;
}

int main()
{
    float *out; unsigned long long *ticks;
    CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&ticks, 64));
    for (int blocks : {1, 256, 1024}) {
#define RUN32(N) run("32x32x2 f32", [&](float *o, unsigned long long *t, int n) { hipLaunchKernelGGL(k32<N>, dim3(blocks), dim3(256), 0, 0, o, t, n); }, N, out, ticks, blocks)
#define RUN16(N) run("16x16x4 f32", [&](float *o, unsigned long long *t, int n) { hipLaunchKernelGGL(k16<N>, dim3(blocks), dim3(256), 0, 0, o, t, n); }, N, out, ticks, blocks)
        RUN32(1); RUN32(2); RUN16(1); RUN16(2); RUN16(4);
    }
    return 0;
}
