// Shared host/device helpers for liborn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/orn.h"
#include "../../include/orn_debug.h"

#define ORN_WAVE 64

// ---- error plumbing --------------------------------------------------------------------------
void orn_set_error(const char *fmt, ...);
// Kernel-form switches of tools/probes (A/B runs) and of the tests that pin the non-default forms.  Every one that is SET is
// reported on stderr the first time it is read ("liborn: probe switch ORN_X=.. is active"), so a stray environment variable can
// not silently change what a fit runs; switches that make results wrong exist only in diagnostic builds (-DORN_PROBE_BUILD).
const char *orn_probe_env(const char *name);
int orn_probe_env_int(const char *name, int dflt);

#define ORN_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            orn_set_error(__VA_ARGS__);                \
            return ORN_E_ARG;                          \
        }                                              \
    } while (0)

#define ORN_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            orn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                \
        }                                                                   \
    } while (0)

#define ORN_HIP(expr)                                                       \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) {                                            \
            orn_set_error("%s: %s", #expr, hipGetErrorString(e__));         \
            return (int)e__;                                                \
        }                                                                   \
    } while (0)

#define ORN_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != 0) return rc__;   \
    } while (0)

// Wave priority of the kernels on the caller's stream that run beside the side branch's full-chip weight-gradient launch in the
// pipelined step (orn_engine.hip): where a SIMD hosts a wave of each, this one issues first -- the caller's stream is the critical
// path, the side branch has slack (same box, 720p step: 1.045 -> 1.036 ms).  Alone on the chip (serial steps) it changes nothing.
#ifdef ORN_PRIO_LEVEL      /* tools/probes: tagged builds with another level (0: none) */
#define ORN_PRIO_HIGH() __builtin_amdgcn_s_setprio(ORN_PRIO_LEVEL)
#else
#define ORN_PRIO_HIGH() __builtin_amdgcn_s_setprio(2)
#endif
static inline int orn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t orn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- device helpers --------------------------------------------------------------------------
// fast variants (bf16 path): v_exp_f32 + v_rcp_f32, ~1 ulp each -- far below bf16 resolution
__device__ __forceinline__ float orn_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float orn_silu(float z) { return z * orn_sigmoid(z); }
// d/dz [z*sigmoid(z)] = s*(1 + z*(1-s))
__device__ __forceinline__ float orn_silu_grad(float z)
{
    const float s = orn_sigmoid(z);
    return s * (1.0f + z * (1.0f - s));
}

// exact-mode variants (accurate expf): used by the fp32 path.  The quotient 1 / (1 + e^-z) is v_rcp_f32 + one Newton step (error
// below 1 ulp of the reciprocal) instead of the IEEE division sequence (~10 instructions): the kernels that evaluate SiLU per
// element (fp32 head forward / fused backward, conv epilogues) are bound by exactly these instructions.  d = +inf (z < -88) keeps
// the plain reciprocal 0: the Newton step would form inf * 0.
__device__ __forceinline__ float orn_sigmoid_exact(float z)
{
    const float d = 1.0f + expf(-z);
    const float r = __builtin_amdgcn_rcpf(d);
    return d < 3.0e38f ? fmaf(r, fmaf(-d, r, 1.0f), r) : r;
}
__device__ __forceinline__ float orn_silu_exact(float z) { return z * orn_sigmoid_exact(z); }
__device__ __forceinline__ float orn_silu_grad_exact(float z)
{
    const float s = orn_sigmoid_exact(z);
    return s * (1.0f + z * (1.0f - s));
}

__device__ __forceinline__ float orn_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Last stem kernel (first linear layer's dW / db from `nslab` partial rows of dh1): one output row per 128 threads.  Shared by
// k_linear_silu_bwd_w_slabs and the wgrad batch launch that carries it as trailing work-groups (Adam is its only consumer).
// B == 1, second linear layer of the stem backward, 16 output rows per 256-thread work-group (k_stem_bwd_l2, or trailing
// work-groups of a later launch: the 16-bit engine runs it inside the batched wgrad launch, since only Adam and the stem's last
// kernel -- which then trails the slab-reduction launch -- consume its results):
//   d[o] = (sum_s dy_slabs[s][o]) * silu'(pre[o]);  db[o] = d[o];  dw[o][k] = d[o] * x[k];  partial[wg][k] = sum_o w[o][k] * d[o]
#define ORN_STEM_ROWS 16
struct OrnStemL2Job {
    const float *w, *x, *pre, *dy_slabs; int nslab; size_t slab_ld; int K, N; float *dw, *db, *partial;
};
// t in [0, 256); sd: 16 floats of LDS; contains a work-group barrier (every thread of the work-group must call)
__device__ __forceinline__ void orn_stem_l2_block(const OrnStemL2Job &j, int blk, int t, float *sd)
{
    const int o0 = blk * ORN_STEM_ROWS, wave = t >> 6, lane = t & 63, N = j.N, K = j.K;
    {   // this wave's four rows together: their slab loads (and pre) are in flight at once
        float v[ORN_STEM_ROWS / 4], pr[ORN_STEM_ROWS / 4];
        int oc[ORN_STEM_ROWS / 4];
#pragma unroll
        for (int i = 0; i < ORN_STEM_ROWS / 4; ++i) {
            const int o = o0 + wave + 4 * i;
            oc[i] = o < N ? o : N - 1;
            pr[i] = j.pre[oc[i]];
            v[i] = 0.f;
        }
        for (int sl = lane; sl < j.nslab; sl += 64) {
#pragma unroll
            for (int i = 0; i < ORN_STEM_ROWS / 4; ++i) v[i] += j.dy_slabs[(size_t)sl * j.slab_ld + oc[i]];
        }
#pragma unroll
        for (int i = 0; i < ORN_STEM_ROWS / 4; ++i) {
            const int r = wave + 4 * i, o = o0 + r;
            const float tot = orn_wave_sum(v[i]);
            if (lane == 0) {
                const float d = o < N ? tot * orn_silu_grad_exact(pr[i]) : 0.f;
                sd[r] = d;
                if (o < N) j.db[o] = d;
            }
        }
    }
    __syncthreads();
    for (int k = t; k < K; k += 256) {
        const float xk = j.x[k];
        float wv[ORN_STEM_ROWS];
#pragma unroll
        for (int r = 0; r < ORN_STEM_ROWS; ++r) {      // unconditional (clamped) loads: all sixteen in flight together
            const int o = o0 + r < N ? o0 + r : N - 1;
            wv[r] = j.w[(size_t)o * K + k];
        }
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < ORN_STEM_ROWS; ++r) {
            const int o = o0 + r;
            if (o < N) j.dw[(size_t)o * K + k] = sd[r] * xk;
            acc = fmaf(wv[r], sd[r], acc);             // sd[r] = 0 past the last row
        }
        j.partial[(size_t)blk * K + k] = acc;
    }
}
struct OrnStemW0Job {
    const float *x; const int *row_idx; size_t row_stride; const float *pre, *dy_slabs; int nslab, K, N; float *dpre, *dw, *db;
};
// t in [0, 128); sh: 2 floats of LDS private to the row's two waves; contains a work-group barrier (every thread must call)
__device__ __forceinline__ void orn_stem_w0_row(const OrnStemW0Job &j, int o, int t, float *sh)
{
    const bool ok = o < j.N;
    const float *x = j.x;
    if (j.row_idx) x += (size_t)(*j.row_idx) * j.row_stride;
    float v = 0.f;
    if (ok)
        for (int r = t; r < j.nslab; r += 128) v += j.dy_slabs[(size_t)r * j.N + o];
    v = orn_wave_sum(v);
    if ((t & 63) == 0) sh[t >> 6] = v;
    __syncthreads();
    if (!ok) return;
    const float d = (sh[0] + sh[1]) * orn_silu_grad_exact(j.pre[o]);
    if (t == 0) { j.dpre[o] = d; j.db[o] = d; }
    for (int k = t; k < j.K; k += 128) j.dw[(size_t)o * j.K + k] = d * x[k];
}

// One output neuron of y = silu(W x + b) per wave (B rows of x); shared by k_linear_silu and the merge launches that carry
// the stem's linear layers as extra work-groups (orn_merge.hip).
struct OrnLinearJob {
    const float *x; const int *row_idx; size_t row_stride; const float *w, *bias; int B, K, N; float *pre, *y;
};
__device__ __forceinline__ void orn_linear_silu_wave(const OrnLinearJob &j, int wave, int lane)
{
    if (wave >= j.N) return;
    const float *x = j.x;
    if (j.row_idx) x += (size_t)(*j.row_idx) * j.row_stride;
    const float *wr = j.w + (size_t)wave * j.K;
    for (int b = 0; b < j.B; ++b) {
        float acc = 0.f;
        for (int k = lane; k < j.K; k += 64) acc = fmaf(wr[k], x[(size_t)b * j.K + k], acc);
        acc = orn_wave_sum(acc);
        if (lane == 0) {
            const float p = acc + j.bias[wave];
            j.pre[(size_t)b * j.N + wave] = p;
            j.y[(size_t)b * j.N + wave] = p / (1.0f + expf(-p));
        }
    }
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0. Deterministic.
__device__ __forceinline__ float orn_block_sum(float v, float *smem /* >= 16 floats */)
{
    v = orn_wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

// Device-side dynamic loss scale + non-finite guard of the engine (one per engine, in its workspace; graph-safe: every
// kernel reads it from memory, nothing is captured by value).  The 16-bit gradient tensors travel multiplied by `gs`
// (fp16: 2^20 at the start; bf16 and fp32: 1).  Any kernel that meets a non-finite gradient on its way into the gradient
// arena raises `flag`; Adam then leaves parameters and moments untouched (the step is skipped, as torch.cuda.amp.GradScaler
// does) and the next schedule advance halves the scale.  After ORN_SCALE_GROWTH_INTERVAL clean steps it doubles again, up
// to its initial value.
// The engine keeps ORN_SCALE_SLOTS of these, one per step of its unrolled graph: the kernels of step r read the scale from and
// raise the flag of entry r, so an overflowing step skips ITSELF only; entry 0 is also the master record (skipped, good,
// backoffs, gs_max, launched), and k_advance folds the flags into the scale and copies it to every entry.
#define ORN_SCALE_SLOTS 4
struct OrnScaleState {
    float gs, inv_gs;      // scale carried by the 16-bit gradient tensors, and its reciprocal
    float gs_max;          // initial value: the scale never grows beyond it
    int32_t flag;          // a non-finite loss or gradient was seen in this entry's step since the last advance
    int32_t skipped;       // (entry 0) optimiser steps skipped so far (Adam's bias corrections do not count them)
    int32_t good;          // (entry 0) clean steps since the last change of scale, credited once they have run
    int32_t backoffs;      // (entry 0) times the scale was halved
    int32_t launched;      // (entry 0) steps of the group launched at the last advance
};
#define ORN_SCALE_GROWTH_INTERVAL 2000
__device__ __forceinline__ void orn_flag_nonfinite(OrnScaleState *sc, float v)
{
    if (sc && !(fabsf(v) <= 3.0e38f)) sc->flag = 1;      // NaN and +-inf; plain store: every writer stores the same value
}

// Device-side state of the step in flight (engine): schedule entry + derived Adam scalars.
struct OrnStepCur {
    int32_t frame, step;
    float lr;
    float step_size;   // lr / (1 - beta1^step), formed in double
    float sqrt_bc2;    // sqrt(1 - beta2^step), formed in double
    int32_t slot;
    int32_t pad[2];
};

// Last stage of the loss (fixed-order sum of the per-tile partials -> loss, L1, MSE, SSIM, PSNR, the stats ring, the non-finite
// flag): one work-group, needed only by Adam at the end of the step -- so the 16-bit engine lets it ride on the head's backward
// launch instead of paying a launch of its own (5 us + a boundary).
struct OrnLossFinalJob {
    const float *part_ssim; int n_ssim; const float *part_l1; int n_l1;    // n_l1 == 0: no job
    double n_elem, n_map; int loss_type; float loss_scale;
    float *stats; const OrnStepCur *cur; float *ring; OrnScaleState *sc;
    // optional (engine, deferred last block): this step's schedule state and scale, copied for the side branch, which still
    // reads them after the main stream has advanced to the next step; the copy's flag starts clear
    OrnStepCur *cur_copy; OrnScaleState *sc_copy;
};
// every thread of the work-group calls it (barriers inside); sd: 3 * blockDim.x doubles of LDS; blockDim.x a power of two
__device__ __forceinline__ void orn_loss_finalize_block(const OrnLossFinalJob &j, double *sd)
{
    const int t = threadIdx.x, nt = blockDim.x;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = t; i < j.n_l1; i += nt) { a += (double)j.part_l1[2 * i]; b += (double)j.part_l1[2 * i + 1]; }
    for (int i = t; i < j.n_ssim; i += nt) c += (double)j.part_ssim[i];
    sd[t] = a; sd[nt + t] = b; sd[2 * nt + t] = c;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (t < s) { sd[t] += sd[t + s]; sd[nt + t] += sd[nt + t + s]; sd[2 * nt + t] += sd[2 * nt + t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        const float l1 = (float)(sd[0] / j.n_elem);
        const float mse = (float)(sd[nt] / j.n_elem);
        const float ss = (j.loss_type == 2 /* ORN_LOSS_FUSION6 */) ? (float)(sd[2 * nt] / j.n_map) : 0.f;
        float loss;
        if (j.loss_type == 0 /* L2 */) loss = mse;
        else if (j.loss_type == 1 /* L1 */) loss = l1;
        else loss = 0.7f * l1 + 0.3f * (1.0f - ss);
        orn_flag_nonfinite(j.sc, loss);                // a NaN / inf forward pass: no update from this step
        const float psnr = -10.0f * log10f(mse);
        j.stats[0] = loss * j.loss_scale; j.stats[1] = l1; j.stats[2] = mse; j.stats[3] = ss; j.stats[4] = psnr;
        j.stats[5] = 0.f; j.stats[6] = 0.f; j.stats[7] = 0.f;
        if (j.cur_copy && j.cur) *j.cur_copy = *j.cur;
        if (j.sc_copy && j.sc) { j.sc_copy->gs = j.sc->gs; j.sc_copy->inv_gs = j.sc->inv_gs; j.sc_copy->gs_max = j.sc->gs_max; j.sc_copy->flag = 0; }
        if (j.ring) {                                  // engine: publish into the per-step ring (slot from the cursor)
            float *r = j.ring + (size_t)j.cur->slot * 8;
            r[0] = loss * j.loss_scale; r[1] = l1; r[2] = mse; r[3] = ss; r[4] = psnr;
            r[5] = j.cur->lr; r[6] = (float)j.cur->frame; r[7] = (float)j.cur->step;
        }
    }
}

// ---- internal cross-file entry points (not exported) -----------------------------------------
// Generic deterministic column reduce: out[j] = sum_{i<rows} in[i*ld + j], fixed order.
int orn_launch_reduce_rows(const float *in, int rows, size_t ld, size_t n, float *out, hipStream_t st);
