// Small HBM-/latency-bound kernels of the hot path: positional encoding (A1), MLP stem (A2),
// 1x1 head (A5), Adam (A9) and the deterministic reduction helpers everything else shares.
#include "orn_internal.h"
#include <stdlib.h>
#include <math.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------
// error text (thread-local)
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void orn_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *orn_probe_env(const char *name)
{
    const char *v = getenv(name);
    if (v) fprintf(stderr, "liborn: probe switch %s=%s is active (non-default kernel form)\n", name, v);
    return v;
}

int orn_probe_env_int(const char *name, int dflt)
{
    const char *v = orn_probe_env(name);
    return v ? atoi(v) : dflt;
}

extern "C" int orn_version(void) { return ORN_VERSION; }

extern "C" int orn_last_error(char *buf, size_t n)
{
    if (!buf || n == 0) return (int)strlen(g_err);
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return (int)strlen(buf);
}

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
// out[j] = sum_i in[i*ld + j], i ascending (fixed order).
__global__ void k_reduce_rows(const float *__restrict__ in, int rows, size_t ld, size_t n, float *__restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    // eight independent partial sums keep eight loads in flight (fixed order: still deterministic)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= rows; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += in[(size_t)(i + k) * ld + j];
    }
    for (; i < rows; ++i) a8[0] += in[(size_t)i * ld + j];
    out[j] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// Many rows, few columns: 32 columns x 32 row-lanes per block; each lane sums rows r = lane (mod 32) in
// ascending order, then the 32 lane sums are added in fixed order through LDS.  Deterministic.
__global__ void __launch_bounds__(1024) k_reduce_rows_tall(const float *__restrict__ in, int rows, size_t ld, size_t n,
                                                          float *__restrict__ out)
{
    __shared__ float sm[32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const size_t j = (size_t)blockIdx.x * 32 + cx;
    float acc = 0.f;
    if (j < n)
        for (int i = ry; i < rows; i += 32) acc += in[(size_t)i * ld + j];
    sm[ry][cx] = acc;
    __syncthreads();
    if (ry < 4) {                                   // 4 partial sums of 8 lanes each, then a fixed 4-way add
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += sm[ry * 8 + k][cx];
        sm[ry * 8][cx] = r;
    }
    __syncthreads();
    if (ry == 0 && j < n) out[j] = (sm[0][cx] + sm[8][cx]) + (sm[16][cx] + sm[24][cx]);
}

int orn_launch_reduce_rows(const float *in, int rows, size_t ld, size_t n, float *out, hipStream_t st)
{
    if (n == 0) return 0;
    if (rows >= 32 && n <= 65536) {
        hipLaunchKernelGGL(k_reduce_rows_tall, dim3(orn_cdiv((long)n, 32)), dim3(1024), 0, st, in, rows, ld, n, out);
        ORN_LAUNCH_CHECK("reduce_rows_tall");
        return 0;
    }
    hipLaunchKernelGGL(k_reduce_rows, dim3(orn_cdiv((long)n, 256)), dim3(256), 0, st, in, rows, ld, n, out);
    ORN_LAUNCH_CHECK("reduce_rows");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// A1  positional encoding                                                  utils.py:121-129
// ------------------------------------------------------------------------------------------------
__global__ void k_pe_fwd(const float *__restrict__ pos, int B, const float *__restrict__ lbase_pow, int levels,
                         float *__restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * levels) return;
    const int b = idx / levels, i = idx % levels;
    // fp32(fp32(pos * fp32(lbase^i)) * fp32(pi)); __fmul_rn blocks any contraction / reassociation.
    const float arg = __fmul_rn(__fmul_rn(pos[b], lbase_pow[i]), 3.14159265358979323846f);
    // accurate (range-reduced) sinf/cosf: arguments reach 1.9e4 rad, no __sinf here.
    out[(size_t)b * 2 * levels + 2 * i] = sinf(arg);
    out[(size_t)b * 2 * levels + 2 * i + 1] = cosf(arg);
}

extern "C" int orn_pe_fwd(const float *pos, int B, const float *lbase_pow, int levels, float *out, void *stream)
{
    ORN_REQUIRE(pos && lbase_pow && out && B > 0 && levels > 0, "pe_fwd: bad arguments");
    hipLaunchKernelGGL(k_pe_fwd, dim3(orn_cdiv((long)B * levels, 128)), dim3(128), 0, (hipStream_t)stream, pos, B,
                       lbase_pow, levels, out);
    ORN_LAUNCH_CHECK("pe_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// A2  stem                                                                 model.py:174-188
// ------------------------------------------------------------------------------------------------
// y[b][o] = silu(pre), pre = bias[o] + sum_k w[o][k]*x[b][k]; one wave per output neuron.
// x row selected through an optional device index (engine: frame index of the current step).
__global__ void k_linear_silu(const float *__restrict__ x, const int *__restrict__ row_idx, size_t row_stride,
                              const float *__restrict__ w, const float *__restrict__ bias, int B, int K, int N,
                              float *__restrict__ pre, float *__restrict__ y)
{
    const OrnLinearJob j = {x, row_idx, row_stride, w, bias, B, K, N, pre, y};
    orn_linear_silu_wave(j, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, threadIdx.x & 63);
}

int orn_launch_linear_silu(const float *x, const int *row_idx, size_t row_stride, const float *w, const float *b,
                           int B, int K, int N, float *pre, float *y, hipStream_t st)
{
    hipLaunchKernelGGL(k_linear_silu, dim3(orn_cdiv((long)N * 64, 256)), dim3(256), 0, st, x, row_idx, row_stride,
                       w, b, B, K, N, pre, y);
    ORN_LAUNCH_CHECK("linear_silu");
    return 0;
}

extern "C" int orn_stem_fwd(const float *embed, const float *w0, const float *b0, const float *w1, const float *b1,
                            int B, int E, int Hd, int Nout, float *pre1, float *h1, float *pre2, float *h2,
                            void *stream)
{
    ORN_REQUIRE(embed && w0 && b0 && w1 && b1 && pre1 && h1 && pre2 && h2, "stem_fwd: null pointer");
    ORN_REQUIRE(B > 0 && E > 0 && Hd > 0 && Nout > 0, "stem_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    ORN_TRY(orn_launch_linear_silu(embed, nullptr, 0, w0, b0, B, E, Hd, pre1, h1, st));
    ORN_TRY(orn_launch_linear_silu(h1, nullptr, 0, w1, b1, B, Hd, Nout, pre2, h2, st));
    return 0;
}

// dpre[b][o] = dy[b][o]*silu'(pre[b][o]);  db[o] = sum_b dpre;  dw[o][k] = sum_b dpre[b][o]*x[b][k]
__global__ void k_linear_silu_bwd_w(const float *__restrict__ x, const int *__restrict__ row_idx, size_t row_stride,
                                    const float *__restrict__ pre, const float *__restrict__ dy, int B, int K, int N,
                                    float *__restrict__ dpre, float *__restrict__ dw, float *__restrict__ db)
{
    // one output row per wave (a row per work-group makes thousands of tiny work-groups: dispatcher-bound)
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= N) return;
    if (row_idx) x += (size_t)(*row_idx) * row_stride;
    float dsum = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(size_t)b * N + o] * orn_silu_grad_exact(pre[(size_t)b * N + o]);
        dsum += d;
        if (lane == 0) dpre[(size_t)b * N + o] = d;
    }
    if (lane == 0) db[o] = dsum;
    for (int k = lane; k < K; k += 64) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = dy[(size_t)b * N + o] * orn_silu_grad_exact(pre[(size_t)b * N + o]);
            acc = fmaf(d, x[(size_t)b * K + k], acc);
        }
        dw[(size_t)o * K + k] = acc;
    }
}

// B == 1 variant whose upstream gradient arrives as `nslab` partial rows (k_linear_bwd_x_partial): the block sums its
// column in fixed order itself, so no standalone reduction launch is needed.
__global__ void __launch_bounds__(128)
k_linear_silu_bwd_w_slabs(const float *__restrict__ x, const int *__restrict__ row_idx, size_t row_stride,
                          const float *__restrict__ pre, const float *__restrict__ dy_slabs, int nslab, int K, int N,
                          float *__restrict__ dpre, float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float sred[16];
    __shared__ float sd;
    const int o = blockIdx.x;
    if (row_idx) x += (size_t)(*row_idx) * row_stride;
    float v = 0.f;
    for (int r = threadIdx.x; r < nslab; r += blockDim.x) v += dy_slabs[(size_t)r * N + o];
    const float tot = orn_block_sum(v, sred);
    if (threadIdx.x == 0) {
        const float d = tot * orn_silu_grad_exact(pre[o]);
        sd = d;
        dpre[o] = d;
        db[o] = d;
    }
    __syncthreads();
    const float d = sd;
    for (int k = threadIdx.x; k < K; k += blockDim.x) dw[(size_t)o * K + k] = d * x[k];
}

// partial[chunk][b][k] = sum_{o in chunk} w[o][k]*dpre[b][o]
__global__ void k_linear_bwd_x_partial(const float *__restrict__ w, const float *__restrict__ dpre, int B, int K,
                                       int N, int rows_per_chunk, float *__restrict__ partial)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int chunk = blockIdx.y;
    if (k >= K) return;
    const int o0 = chunk * rows_per_chunk, o1 = min(N, o0 + rows_per_chunk);
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int o = o0; o < o1; ++o) acc = fmaf(w[(size_t)o * K + k], dpre[(size_t)b * N + o], acc);
        partial[((size_t)chunk * B + b) * K + k] = acc;
    }
}

#define ORN_STEM_CHUNKS 256

// B == 1, second linear layer, one launch (orn_stem_l2_block, orn_common.h): the upstream gradient arrives as `nslab` partial rows
// (the first block's per-work-group dx shares, or one finished row), 16 output rows per work-group
// (was: slab reduction + k_linear_silu_bwd_w + k_linear_bwd_x_partial, three graph nodes of ~5-9 us each)
__global__ void __launch_bounds__(256) k_stem_bwd_l2(OrnStemL2Job j)
{
    __shared__ float sd[ORN_STEM_ROWS];
    orn_stem_l2_block(j, (int)blockIdx.x, (int)threadIdx.x, sd);
}

int orn_launch_stem_bwd(const float *embed, const int *row_idx, size_t row_stride, const float *w1, const float *pre1,
                        const float *h1, const float *pre2, const float *dh2, int B, int E, int Hd, int Nout,
                        float *dw0, float *db0, float *dw1, float *db1, float *ws, hipStream_t st, int dh2_nslab, OrnStemW0Job *defer_w0,
                        OrnStemL2Job *defer_l2)
{
    // ws: dpre2 [B*Nout] | dpre1 [B*Hd] | dh1 [B*Hd] | partial [CHUNKS*B*Hd]
    float *dpre2 = ws;
    float *dpre1 = dpre2 + (size_t)B * Nout;
    float *dh1 = dpre1 + (size_t)B * Hd;
    float *partial = dh1 + (size_t)B * Hd;
    if (B == 1) {
        const int nwg = orn_cdiv(Nout, ORN_STEM_ROWS);
        const OrnStemL2Job l2 = {w1, h1, pre2, dh2, dh2_nslab, (size_t)Nout, Hd, Nout, dw1, db1, partial};
        if (defer_l2 && defer_w0) *defer_l2 = l2;      // the caller runs it as trailing work-groups of a later launch (and w0 behind that one)
        else {
            hipLaunchKernelGGL(k_stem_bwd_l2, dim3(nwg), dim3(256), 0, st, l2);
            ORN_LAUNCH_CHECK("stem_bwd_l2");
        }
        if (defer_w0) {      // the caller appends this job to a later launch (orn_stem_w0_row)
            *defer_w0 = OrnStemW0Job{embed, row_idx, row_stride, pre1, partial, nwg, E, Hd, dpre1, dw0, db0};
            return 0;
        }
        hipLaunchKernelGGL(k_linear_silu_bwd_w_slabs, dim3(Hd), dim3(128), 0, st, embed, row_idx, row_stride, pre1, partial, nwg, E, Hd,
                           dpre1, dw0, db0);
        ORN_LAUNCH_CHECK("stem_bwd_w0");
        return 0;
    }
    ORN_REQUIRE(dh2_nslab == 1, "stem_bwd: gradient slabs need B == 1");
    hipLaunchKernelGGL(k_linear_silu_bwd_w, dim3(orn_cdiv(Nout, 4)), dim3(256), 0, st, h1, nullptr, 0, pre2, dh2, B, Hd, Nout,
                       dpre2, dw1, db1);
    ORN_LAUNCH_CHECK("stem_bwd_w1");
    const int rpc = orn_cdiv(Nout, ORN_STEM_CHUNKS);
    hipLaunchKernelGGL(k_linear_bwd_x_partial, dim3(orn_cdiv(Hd, 128), ORN_STEM_CHUNKS), dim3(128), 0, st, w1, dpre2, B,
                       Hd, Nout, rpc, partial);
    ORN_LAUNCH_CHECK("stem_bwd_x");
    ORN_TRY(orn_launch_reduce_rows(partial, ORN_STEM_CHUNKS, (size_t)B * Hd, (size_t)B * Hd, dh1, st));
    hipLaunchKernelGGL(k_linear_silu_bwd_w, dim3(orn_cdiv(Hd, 2)), dim3(128), 0, st, embed, row_idx, row_stride, pre1, dh1, B, E, Hd,
                       dpre1, dw0, db0);
    ORN_LAUNCH_CHECK("stem_bwd_w0");
    return 0;
}

size_t orn_stem_bwd_ws_floats(int B, int Hd, int Nout)
{
    const size_t rows = (size_t)orn_cdiv(Nout, ORN_STEM_ROWS) > ORN_STEM_CHUNKS ? (size_t)orn_cdiv(Nout, ORN_STEM_ROWS) : ORN_STEM_CHUNKS;
    return (size_t)B * Nout + 2 * (size_t)B * Hd + rows * B * Hd;
}

extern "C" int orn_stem_bwd(const float *embed, const float *w1, const float *pre1, const float *h1,
                            const float *pre2, const float *dh2, int B, int E, int Hd, int Nout, float *dw0,
                            float *db0, float *dw1, float *db1, float *ws, void *stream)
{
    ORN_REQUIRE(embed && w1 && pre1 && h1 && pre2 && dh2 && dw0 && db0 && dw1 && db1 && ws, "stem_bwd: null pointer");
    ORN_REQUIRE(B > 0 && E > 0 && Hd > 0 && Nout > 0, "stem_bwd: bad sizes");
    return orn_launch_stem_bwd(embed, nullptr, 0, w1, pre1, h1, pre2, dh2, B, E, Hd, Nout, dw0, db0, dw1, db1, ws,
                               (hipStream_t)stream, 1, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// A5  head                                                                 model.py:621-622
// ------------------------------------------------------------------------------------------------
#define ORN_HEAD_MAXC 512

// ZIN: the input is the pre-activation z and a = SiLU(z) is formed here (fp32 engine's training step: the last block stores z only)
template <int V, bool ZIN = false>
__global__ void k_head_fwd(const float *__restrict__ a, const float *__restrict__ w, const float *__restrict__ bias, int C,
                           size_t HW, int sigmoid, float *__restrict__ out)
{
    __shared__ float sw[3 * ORN_HEAD_MAXC + 3];
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sw[i] = w[i];
    if (threadIdx.x < 3) sw[3 * C + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y;
    const size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (p >= HW) return;
    const float *ab = a + (size_t)b * C * HW + p;
    float acc[3][V];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[k][v] = 0.f;
    for (int c = 0; c < C; ++c) {
        float av[V];
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(ab + (size_t)c * HW);
            av[0] = t.x; av[1 % V] = t.y; av[2 % V] = t.z; av[3 % V] = t.w;
        } else {
            av[0] = ab[(size_t)c * HW];
        }
        if (ZIN) {
#pragma unroll
            for (int v = 0; v < V; ++v) av[v] = orn_silu_exact(av[v]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[k][v] = fmaf(sw[k * C + c], av[v], acc[k][v]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float r[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float u = acc[k][v] + sw[3 * C + k];
            r[v] = sigmoid ? 1.0f / (1.0f + expf(-u)) : (tanhf(u) + 1.0f) * 0.5f;
        }
        float *o = out + ((size_t)b * 3 + k) * HW + p;
        if (V == 4) *reinterpret_cast<float4 *>(o) = make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]);
        else o[0] = r[0];
    }
}

int orn_launch_head_fwd(const float *a, const float *w, const float *b, int B, int C, size_t HW, int sigmoid,
                        float *out, hipStream_t st, bool z_input)
{
    ORN_REQUIRE(C <= ORN_HEAD_MAXC, "head: C=%d > %d unsupported", C, ORN_HEAD_MAXC);
    if (z_input) {
        if (HW % 4 == 0)
            hipLaunchKernelGGL((k_head_fwd<4, true>), dim3(orn_cdiv((long)HW / 4, 256), B), dim3(256), 0, st, a, w, b, C, HW, sigmoid, out);
        else
            hipLaunchKernelGGL((k_head_fwd<1, true>), dim3(orn_cdiv((long)HW, 256), B), dim3(256), 0, st, a, w, b, C, HW, sigmoid, out);
        ORN_LAUNCH_CHECK("head_fwd(z)");
        return 0;
    }
    if (HW % 4 == 0)
        hipLaunchKernelGGL(k_head_fwd<4>, dim3(orn_cdiv((long)HW / 4, 256), B), dim3(256), 0, st, a, w, b, C, HW, sigmoid, out);
    else
        hipLaunchKernelGGL(k_head_fwd<1>, dim3(orn_cdiv((long)HW, 256), B), dim3(256), 0, st, a, w, b, C, HW, sigmoid, out);
    ORN_LAUNCH_CHECK("head_fwd");
    return 0;
}

extern "C" int orn_head_fwd(const float *a, const float *w, const float *b, int B, int C, int H, int W, int sigmoid,
                            float *out, void *stream)
{
    ORN_REQUIRE(a && w && b && out && B > 0 && C > 0 && H > 0 && W > 0, "head_fwd: bad arguments");
    return orn_launch_head_fwd(a, w, b, B, C, (size_t)H * W, sigmoid, out, (hipStream_t)stream, false);
}

// Backward.  Block = 256 threads x PPT pixels of one batch item.  du kept in registers; per channel c:
// da[c][p] = sum_k w[k][c]*du[k][p] (stored), s_k = sum_p du[k][p]*a[c][p] -> wave reduce -> LDS -> partial.
// partial layout: [block][3*C + 3]  (dw[k][c] at k*C+c, db[k] at 3*C+k); reduced over blocks afterwards.
#define ORN_HEAD_PPT 8
__global__ void __launch_bounds__(256)
k_head_bwd(const float *__restrict__ a, const float *__restrict__ w, const float *__restrict__ out,
           const float *__restrict__ dout, int C, size_t HW, int sigmoid, float *__restrict__ da,
           float *__restrict__ partial)
{
    __shared__ float sw[3 * ORN_HEAD_MAXC];
    __shared__ float sred[4][3 * ORN_HEAD_MAXC + 3];
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t p0 = (size_t)blockIdx.x * 256 * ORN_HEAD_PPT;
    float du[3][ORN_HEAD_PPT];
    size_t pp[ORN_HEAD_PPT];
#pragma unroll
    for (int i = 0; i < ORN_HEAD_PPT; ++i) {
        pp[i] = p0 + (size_t)i * 256 + threadIdx.x;
        const bool ok = pp[i] < HW;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float d = 0.f;
            if (ok) {
                const float o = out[((size_t)b * 3 + k) * HW + pp[i]];
                const float g = dout[((size_t)b * 3 + k) * HW + pp[i]];
                // o = (tanh u + 1)/2 -> do/du = (1 - tanh^2)/2 = 2 o (1-o);  sigmoid: o (1-o)
                d = g * (sigmoid ? o * (1.0f - o) : 2.0f * o * (1.0f - o));
            }
            du[k][i] = d;
        }
    }
    float dbs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ORN_HEAD_PPT; ++i) s += du[k][i];
        dbs[k] = orn_wave_sum(s);
    }
    if (lane == 0) {
        sred[wave][3 * C + 0] = dbs[0];
        sred[wave][3 * C + 1] = dbs[1];
        sred[wave][3 * C + 2] = dbs[2];
    }
    for (int c = 0; c < C; ++c) {
        const float w0 = sw[c], w1 = sw[C + c], w2 = sw[2 * C + c];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < ORN_HEAD_PPT; ++i) {
            if (pp[i] < HW) {
                const size_t idx = ((size_t)b * C + c) * HW + pp[i];
                const float av = a[idx];
                s0 = fmaf(du[0][i], av, s0);
                s1 = fmaf(du[1][i], av, s1);
                s2 = fmaf(du[2][i], av, s2);
                da[idx] = fmaf(w2, du[2][i], fmaf(w1, du[1][i], w0 * du[0][i]));
            }
        }
        s0 = orn_wave_sum(s0);
        s1 = orn_wave_sum(s1);
        s2 = orn_wave_sum(s2);
        if (lane == 0) {
            sred[wave][c] = s0;
            sred[wave][C + c] = s1;
            sred[wave][2 * C + c] = s2;
        }
    }
    __syncthreads();
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 3 * C + 3; i += blockDim.x)
        partial[blk * (3 * C + 3) + i] = (sred[0][i] + sred[1][i]) + (sred[2][i] + sred[3][i]);
}

__global__ void k_head_split_dw(const float *__restrict__ red, int C, float *__restrict__ dw, float *__restrict__ db)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * C) dw[i] = red[i];
    else if (i < 3 * C + 3) db[i - 3 * C] = red[i];
}

extern "C" size_t orn_head_bwd_ws_bytes(int B, int C, int H, int W)
{
    const size_t HW = (size_t)H * W;
    const size_t nblk = (size_t)orn_cdiv((long)HW, 256 * ORN_HEAD_PPT) * B;
    return orn_align((nblk + 1) * (3 * (size_t)C + 3) * sizeof(float));
}

int orn_launch_head_bwd(const float *a, const float *w, const float *out, const float *dout, int B, int C, size_t HW,
                        int sigmoid, float *da, float *dw, float *db, float *ws, hipStream_t st)
{
    ORN_REQUIRE(C <= ORN_HEAD_MAXC, "head: C=%d > %d unsupported", C, ORN_HEAD_MAXC);
    const int gx = orn_cdiv((long)HW, 256 * ORN_HEAD_PPT);
    const size_t nblk = (size_t)gx * B, n = 3 * (size_t)C + 3;
    float *partial = ws, *red = ws + nblk * n;
    hipLaunchKernelGGL(k_head_bwd, dim3(gx, B), dim3(256), 0, st, a, w, out, dout, C, HW, sigmoid, da, partial);
    ORN_LAUNCH_CHECK("head_bwd");
    ORN_TRY(orn_launch_reduce_rows(partial, (int)nblk, n, n, red, st));
    hipLaunchKernelGGL(k_head_split_dw, dim3(orn_cdiv((long)n, 128)), dim3(128), 0, st, red, C, dw, db);
    ORN_LAUNCH_CHECK("head_split");
    return 0;
}

// fp32 engine, last block with a stride-2 PixelShuffle: head backward + SiLU' + un-shuffle in ONE pass over z.
//   du = dout * act'(out);  a = SiLU(z) (recomputed: bit-identical to the forward's);  dz = (W^T du) * SiLU'(z)
//   dy[o = 4 n + 2 si + sj][h][w] = dz[n][2 h + si][2 w + sj];  head dW[k][n] += du[k] a;  head db[k] += du[k];  conv dbias[o] += dy
// The unfused pair (k_head_bwd: read a, write da; k_silu_bwd_unshuffle: read da and z, write dy) moved 5 activation-sized
// tensors through HBM (1.77 GB at 720p, 750 us); this one moves 2 (z in, dy out).
// Block = 256 threads x 2 low-res pixels; partials in fixed order.
#define HF_PPT 2                      /* 512 low-res pixels per work-group: 450 work-groups at 720p (2048 pixels per work-group left */
#define HF_THREADS 256                /* half the CUs idle: 306 us) */
__global__ void __launch_bounds__(HF_THREADS)
k_head_bwd_fused_f32(const float *__restrict__ z, const float *__restrict__ w, const float *__restrict__ out,
                     const float *__restrict__ dout, int Cn, int H, int W, int sigmoid, float *__restrict__ dy,
                     float *__restrict__ dbp, float *__restrict__ hpart)
{
    extern __shared__ float hf_smem[];
    float *sw = hf_smem;                              // [3 Cn]
    float *sred = hf_smem + 3 * Cn;                   // [8 waves][7 Cn + 3]: head dW (3 Cn), conv dbias (4 Cn), head db (3)
    const int nred = 7 * Cn + 3;
    for (int i = threadIdx.x; i < 3 * Cn; i += HF_THREADS) sw[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ws = 2 * W;
    const size_t HW = (size_t)H * W, HWs = 4 * HW;
    float du[3][HF_PPT][4];
    int zoff[HF_PPT];                                 // offset of (2h, 2w) in a high-res plane; < 0: no such pixel
    unsigned pix[HF_PPT];
#pragma unroll
    for (int i = 0; i < HF_PPT; ++i) {
        const size_t pp = (size_t)blockIdx.x * (HF_THREADS * HF_PPT) + (size_t)i * HF_THREADS + threadIdx.x;
        const bool ok = pp < HW;
        const int h = ok ? (int)(pp / W) : 0, ww = ok ? (int)(pp - (size_t)h * W) : 0;
        zoff[i] = ok ? (2 * h) * Ws + 2 * ww : -1;
        pix[i] = (unsigned)pp;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                float2 o = make_float2(0.f, 0.f), g = make_float2(0.f, 0.f);
                if (ok) {
                    o = *reinterpret_cast<const float2 *>(out + (size_t)k * HWs + zoff[i] + si * Ws);
                    g = *reinterpret_cast<const float2 *>(dout + (size_t)k * HWs + zoff[i] + si * Ws);
                }
                // o = (tanh u + 1)/2 -> do/du = 2 o (1-o);  sigmoid: o (1-o)
                du[k][i][2 * si + 0] = g.x * (sigmoid ? o.x * (1.0f - o.x) : 2.0f * o.x * (1.0f - o.x));
                du[k][i][2 * si + 1] = g.y * (sigmoid ? o.y * (1.0f - o.y) : 2.0f * o.y * (1.0f - o.y));
            }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float sdu = 0.f;
#pragma unroll
        for (int i = 0; i < HF_PPT; ++i) sdu += (du[k][i][0] + du[k][i][1]) + (du[k][i][2] + du[k][i][3]);
        sdu = orn_wave_sum(sdu);
        if (lane == 0) sred[wave * nred + 7 * Cn + k] = sdu;
    }
    for (int n = 0; n < Cn; ++n) {
        const float w0 = sw[n], w1 = sw[Cn + n], w2 = sw[2 * Cn + n];
        const float *zn = z + (size_t)n * HWs;
        float2 zv[HF_PPT][2];
#pragma unroll
        for (int i = 0; i < HF_PPT; ++i)
#pragma unroll
            for (int si = 0; si < 2; ++si)
                zv[i][si] = zoff[i] >= 0 ? *reinterpret_cast<const float2 *>(zn + zoff[i] + si * Ws) : make_float2(0.f, 0.f);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, db4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < HF_PPT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float zz = (q & 1) ? zv[i][q >> 1].y : zv[i][q >> 1].x;
                const float sg = orn_sigmoid_exact(zz);
                const float a = zz * sg;                                           // == orn_silu_exact(zz): what the forward's head read
                const float da = fmaf(w2, du[2][i][q], fmaf(w1, du[1][i][q], w0 * du[0][i][q]));
                const float dz = da * (sg * (1.0f + zz * (1.0f - sg)));             // orn_silu_grad_exact
                s0 = fmaf(du[0][i][q], a, s0);
                s1 = fmaf(du[1][i][q], a, s1);
                s2 = fmaf(du[2][i][q], a, s2);
                db4[q] += dz;
                if (zoff[i] >= 0) dy[((size_t)n * 4 + q) * HW + pix[i]] = dz;
            }
        s0 = orn_wave_sum(s0); s1 = orn_wave_sum(s1); s2 = orn_wave_sum(s2);
#pragma unroll
        for (int q = 0; q < 4; ++q) db4[q] = orn_wave_sum(db4[q]);
        if (lane == 0) {
            float *r = sred + wave * nred;
            r[n] = s0; r[Cn + n] = s1; r[2 * Cn + n] = s2;
            r[3 * Cn + 4 * n + 0] = db4[0]; r[3 * Cn + 4 * n + 1] = db4[1]; r[3 * Cn + 4 * n + 2] = db4[2]; r[3 * Cn + 4 * n + 3] = db4[3];
        }
    }
    __syncthreads();
    const int O = 4 * Cn;
    for (int i = threadIdx.x; i < nred; i += HF_THREADS) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < HF_THREADS / 64; ++wv) v += sred[wv * nred + i];
        if (i < 3 * Cn) hpart[(size_t)blockIdx.x * (3 * Cn + 3) + i] = v;
        else if (i < 7 * Cn) dbp[(size_t)blockIdx.x * O + (i - 3 * Cn)] = v;
        else hpart[(size_t)blockIdx.x * (3 * Cn + 3) + 3 * Cn + (i - 7 * Cn)] = v;
    }
}

int orn_head_bwd_fused_f32_blocks(int H, int W) { return orn_cdiv((long)H * W, HF_THREADS * HF_PPT); }

// dy [4 Cn][H][W] and dbp [blocks][4 Cn] belong to the caller (the conv backward's workspace); hws: (blocks + 1) x (3 Cn + 3) floats
// column sums of the head's per-work-group partials straight into dW [3][C] and db [3] (was: reduce_rows + a split kernel);
// 32 columns x 32 row-lanes per work-group, rows r = lane (mod 32) ascending, then a fixed tree: deterministic
__global__ void __launch_bounds__(1024) k_head_partials_finish(const float *__restrict__ in, int rows, int n, int C,
                                                              float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float sm[32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int j = (int)blockIdx.x * 32 + cx;
    float acc = 0.f;
    if (j < n)
        for (int i = ry; i < rows; i += 32) acc += in[(size_t)i * n + j];
    sm[ry][cx] = acc;
    __syncthreads();
    if (ry < 4) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += sm[ry * 8 + k][cx];
        sm[ry * 8][cx] = r;
    }
    __syncthreads();
    if (ry == 0 && j < n) {
        const float v = (sm[0][cx] + sm[8][cx]) + (sm[16][cx] + sm[24][cx]);
        if (j < 3 * C) dw[j] = v;
        else db[j - 3 * C] = v;
    }
}

int orn_launch_head_bwd_fused_f32(const float *z, const float *w, const float *out, const float *dout, int Cn, int H, int W,
                                  int sigmoid, float *dy, float *dbp, float *dw, float *db, float *hws, hipStream_t st)
{
    ORN_REQUIRE(Cn <= ORN_HEAD_MAXC, "head: C=%d > %d unsupported", Cn, ORN_HEAD_MAXC);
    const int nblk = orn_head_bwd_fused_f32_blocks(H, W);
    const size_t n = 3 * (size_t)Cn + 3;
    const size_t lds = ((size_t)3 * Cn + (size_t)(HF_THREADS / 64) * (7 * Cn + 3)) * sizeof(float);
    float *partial = hws, *red = hws + (size_t)nblk * n;
    hipLaunchKernelGGL(k_head_bwd_fused_f32, dim3(nblk), dim3(HF_THREADS), lds, st, z, w, out, dout, Cn, H, W, sigmoid, dy, dbp, partial);
    ORN_LAUNCH_CHECK("head_bwd_fused_f32");
    (void)red;
    hipLaunchKernelGGL(k_head_partials_finish, dim3(orn_cdiv((long)n, 32)), dim3(1024), 0, st, partial, nblk, (int)n, Cn, dw, db);
    ORN_LAUNCH_CHECK("head_partials_finish");
    return 0;
}

extern "C" int orn_head_bwd(const float *a, const float *w, const float *out, const float *dout, int B, int C, int H,
                            int W, int sigmoid, float *da, float *dw, float *db, void *ws, size_t ws_bytes,
                            void *stream)
{
    ORN_REQUIRE(a && w && out && dout && da && dw && db && ws, "head_bwd: null pointer");
    ORN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "head_bwd: bad sizes");
    if (ws_bytes < orn_head_bwd_ws_bytes(B, C, H, W)) {
        orn_set_error("head_bwd: workspace %zu < %zu", ws_bytes, orn_head_bwd_ws_bytes(B, C, H, W));
        return ORN_E_WS;
    }
    return orn_launch_head_bwd(a, w, out, dout, B, C, (size_t)H * W, sigmoid, da, dw, db, (float *)ws,
                               (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// A9  Adam                                                                 main_train.py:196,250
// ------------------------------------------------------------------------------------------------
// Matches torch.optim.Adam's algebra:
//   m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// step_size / sqrt_bc2 come by value (per-op API) or from the engine's device-side step state.
template <bool MASK>
__global__ void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                       size_t n, float step_size_v, float sqrt_bc2_v, const OrnStepCur *__restrict__ sp, float beta1,
                       float omb1, float beta2, float omb2, float eps, float inv_gscale, const float *__restrict__ gmask,
                       OrnScaleState *sc, OrnScaleState *sc_master, OrnScaleState *mirror)
{
    ORN_PRIO_HIGH();
    // non-finite gradients somewhere in this step: leave parameters and moments alone (the whole step is skipped)
    if (sc && sc->flag) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (sc_master) sc_master->skipped += 1;
            if (mirror) mirror->flag = 1;       // the side branch's Adam launch of this step follows the same decision
        }
        return;
    }
    // gmask (optional, 0/1 per parameter): the gradient is multiplied by it -- the prune fine-tune of main_eval.py,
    // where a masked weight (weight_orig * mask) only ever receives the masked gradient and a frozen tensor none
    float step_size = step_size_v, sqrt_bc2 = sqrt_bc2_v;
    if (sp) { step_size = sp->step_size; sqrt_bc2 = sp->sqrt_bc2; }
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    if (i0 + 4 <= n) {
        float4 pv = *reinterpret_cast<float4 *>(p + i0);
        const float4 gv = *reinterpret_cast<const float4 *>(g + i0);
        float4 mv = *reinterpret_cast<float4 *>(m + i0);
        float4 vv = *reinterpret_cast<float4 *>(v + i0);
        float *pp = &pv.x, *mp = &mv.x, *vp = &vv.x;
        const float *gp = &gv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * inv_gscale * (MASK ? gmask[i0 + k] : 1.0f);
            mp[k] = __fadd_rn(__fmul_rn(beta1, mp[k]), __fmul_rn(omb1, gg));
            vp[k] = __fadd_rn(__fmul_rn(beta2, vp[k]), __fmul_rn(__fmul_rn(omb2, gg), gg));
            pp[k] -= step_size * (mp[k] / (sqrtf(vp[k]) / sqrt_bc2 + eps));
        }
        *reinterpret_cast<float4 *>(p + i0) = pv;
        *reinterpret_cast<float4 *>(m + i0) = mv;
        *reinterpret_cast<float4 *>(v + i0) = vv;
    } else {
        for (size_t i = i0; i < n; ++i) {
            const float gg = g[i] * inv_gscale * (MASK ? gmask[i] : 1.0f);
            const float mm = __fadd_rn(__fmul_rn(beta1, m[i]), __fmul_rn(omb1, gg));
            const float vv = __fadd_rn(__fmul_rn(beta2, v[i]), __fmul_rn(__fmul_rn(omb2, gg), gg));
            m[i] = mm;
            v[i] = vv;
            p[i] -= step_size * (mm / (sqrtf(vv) / sqrt_bc2 + eps));
        }
    }
}

int orn_launch_adam(float *p, const float *g, float *m, float *v, size_t n, double lr, int step, const OrnStepCur *sp,
                    double beta1, double beta2, double eps, float inv_gscale, hipStream_t st, const float *gmask, OrnScaleState *sc,
                    OrnScaleState *sc_master, OrnScaleState *mirror, bool count_skip)
{
    OrnScaleState *const master = count_skip ? (sc_master ? sc_master : sc) : nullptr;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const dim3 gr(orn_cdiv((long)orn_cdiv((long)n, 4), 256));
    if (gmask)
        hipLaunchKernelGGL(k_adam<true>, gr, dim3(256), 0, st, p, g, m, v, n, (float)(lr / bc1), (float)sqrt(bc2), sp, (float)beta1,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, inv_gscale, gmask, sc, master, mirror);
    else
        hipLaunchKernelGGL(k_adam<false>, gr, dim3(256), 0, st, p, g, m, v, n, (float)(lr / bc1), (float)sqrt(bc2), sp, (float)beta1,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, inv_gscale, gmask, sc, master, mirror);
    ORN_LAUNCH_CHECK("adam");
    return 0;
}

extern "C" int orn_adam_step(float *p, const float *g, float *m, float *v, size_t n, double lr, double beta1,
                             double beta2, double eps, int step, void *stream)
{
    ORN_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    ORN_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adam_step: arenas must be 16-byte aligned");
    return orn_launch_adam(p, g, m, v, n, lr, step, nullptr, beta1, beta2, eps, 1.0f, (hipStream_t)stream, nullptr);
}
