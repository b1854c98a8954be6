// A4  NeRVBlock conv path in exact fp32 (model.py:539,567 and its autograd backward):
//   fwd   : y = conv3x3(x, Wf, bf, pad 1)  ->  z = PixelShuffle_s(y)  ->  a = SiLU(z)
//   bwd   : dy = unshuffle(da * SiLU'(z));  dbf = sum_hw dy;  dWf = wgrad(x, dy);  dx = dgrad(dy, Wf)
//
// All three contractions are implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32, one rounding per
// product, fp32 accumulate), operands staged through LDS, tensors NCHW fp32 exactly as PyTorch holds
// them.  MFMA bound: roofline = 157.3 TFLOP/s (fp32 matrix).
//
// Lane maps (MI355X guide): A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include "orn_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ================================================================================================
// Implicit-GEMM 3x3 conv (pad 1, stride 1).  D rows = output channel, D cols = 32 pixels of a row.
// Work-group tile: 64 output channels x (8 rows x 32 cols); 4 waves, wave w owns rows 2w, 2w+1.
// K loop: chunks of 8 input channels; MFMA step t=(cp,ij) contracts channels {cp, cp+4} (lane half
// picks one) at tap ij -- any K permutation is a legal fp32 summation order for the conv.
// ================================================================================================
#define CV_BO 64
#define CV_TH 8
#define CV_TW 32
#define CV_CC 8
#define CV_XW (CV_TW + 2)
#define CV_XH (CV_TH + 2)
#define CV_WLD (CV_CC * 9 + 1)

enum { EPI_PLAIN = 0, EPI_PS_SILU = 1 };

struct ConvP {
    const float *x;     // [B, C, H, W]
    const float *w;     // [O, C, 3, 3]
    const float *bias;  // [O] or null
    float *out;         // EPI_PLAIN: [B,O,H,W];  EPI_PS_SILU: a [B,Cn,H*s,W*s] (may be null: z only -- the fp32 engine's last block, whose
                        //            only consumer, the head, recomputes SiLU(z))
    float *z;           // EPI_PS_SILU: pre-activation (may be null)
    int B, C, O, H, W, s;
    int tiles_w, tiles_h;
    int nsplit, c_per_split;   // EPI_PLAIN: input channels split over blockIdx.z (partial slabs, reduced afterwards)
    int o_first;               // first output channel of this launch (a ragged O is covered by a second, narrower launch)
};

// SMALL = 1: 2-row tiles (64 pixels), each wave one 32x32 block -- 4x more work-groups for the small early layers,
// whose cost is latency, not throughput.
#ifndef F32_MINB
#define F32_MINB 2          /* two work-groups per CU: <= 256 VGPRs */
#endif
// NOB = 32-channel blocks of the work-group's output-channel tile: 2 (64 channels), or 1 for the last 32 channels of an O that is
// 32 mod 64 (the dgrad of every 96-channel layer: as a second 64-channel tile, half of its MFMA work was thrown away).
template <int EPI, int SMALL, int NOB = 2>
__global__ void __launch_bounds__(256, F32_MINB) k_conv3x3_f32(ConvP p)
{
    constexpr int TH = SMALL ? 2 : CV_TH;
    constexpr int XH = TH + 2;
    constexpr int RW = SMALL ? 1 : 2;          // rows per wave
    constexpr int OBW = SMALL ? 1 : NOB;       // 32-channel blocks per wave
    constexpr int BO = SMALL ? CV_BO : 32 * NOB;
    __shared__ float Xs[2][CV_CC][XH][CV_XW];
    __shared__ float Ws[2][BO][CV_WLD];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x;
    const int tw = tile % p.tiles_w, th = tile / p.tiles_w;
    const int h0 = th * TH, w0 = tw * CV_TW;
    const int o0 = p.o_first + blockIdx.y * BO;
    const int b = blockIdx.z / p.nsplit, split = blockIdx.z - b * p.nsplit;
    const int C = p.C, H = p.H, W = p.W;
    const float *xb = p.x + (size_t)b * C * H * W;
    const int c_begin = split * p.c_per_split, c_end = min(C, c_begin + p.c_per_split);

    f32x16 acc[OBW][RW];
#pragma unroll
    for (int i = 0; i < OBW; ++i)
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int r0 = SMALL ? (wave >> 1) : wave * 2;     // first output row of the wave (tile-local)
    const int ob0 = SMALL ? (wave & 1) : 0;            // first 32-channel block of the wave
    // Round 3: the K loop is double-buffered.  The global loads of chunk k+1 are issued into registers BEFORE the 36 MFMA k-steps
    // of chunk k and stored to the other LDS buffer behind them: one barrier per chunk and no exposed memory latency (the first
    // form staged a chunk, synchronised, computed, synchronised: two work-groups per CU took turns waiting for memory).
    // Per-thread source offsets of the input patch are chunk-invariant (relative to the chunk's first channel) and computed once.
    constexpr int XN = (CV_CC * XH * CV_XW + 255) / 256, WN = (BO * CV_CC * 9) / 256;
    int xoff[XN];                                      // < 0: zero padding (outside the image / the patch)
#pragma unroll
    for (int it = 0; it < XN; ++it) {
        const int idx = t + it * 256;
        const int c = idx / (XH * CV_XW);
        const int rem = idx - c * (XH * CV_XW);
        const int r = rem / CV_XW, xx = rem - r * CV_XW;
        const int gh = h0 + r - 1, gw = w0 + xx - 1;
        const bool ok = idx < CV_CC * XH * CV_XW && gh >= 0 && gh < H && gw >= 0 && gw < W;
        xoff[it] = ok ? (c * H + gh) * W + gw : -1 - c;        // (c kept for the channel-range test of the last chunk)
    }
    float xr[XN], wr[WN];
    auto gload = [&](const int c0) {
        const float *xc = xb + (size_t)c0 * H * W;
        const int cleft = c_end - c0;                  // channels of this chunk that exist
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = t + it * 256;
            const int c = idx / (XH * CV_XW);
            xr[it] = (xoff[it] >= 0 && c < cleft) ? xc[xoff[it]] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < WN; ++it) {
            const int idx = t + it * 256;
            const int o = idx / (CV_CC * 9), kk = idx - o * (CV_CC * 9);
            wr[it] = (o0 + o < p.O && kk < cleft * 9) ? p.w[(size_t)(o0 + o) * C * 9 + (size_t)c0 * 9 + kk] : 0.f;
        }
    };
    auto lstore = [&](float (*xs)[XH][CV_XW], float (*ws)[CV_WLD]) {
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = t + it * 256;
            if (idx < CV_CC * XH * CV_XW) (&xs[0][0][0])[idx] = xr[it];
        }
#pragma unroll
        for (int it = 0; it < WN; ++it) {
            const int idx = t + it * 256;
            const int o = idx / (CV_CC * 9), kk = idx - o * (CV_CC * 9);
            ws[o][kk] = wr[it];
        }
    };
    if (c_begin < c_end) { gload(c_begin); lstore(Xs[0], Ws[0]); }
    __syncthreads();
    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += CV_CC, buf ^= 1) {
        const bool has_next = c0 + CV_CC < c_end;
#if !defined(CVF_ABL) || !(CVF_ABL & 1)
        if (has_next) gload(c0 + CV_CC);
#endif
        // ---- 36 MFMA k-steps ---------------------------------------------------------------------
#pragma unroll
        for (int cp = 0; cp < CV_CC / 2; ++cp) {
            const int ch = cp + (CV_CC / 2) * hh;
            float xv[RW + 2][3];
#pragma unroll
            for (int r = 0; r < RW + 2; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) xv[r][j] = Xs[buf][ch][r0 + r][l31 + j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
#pragma unroll
                    for (int ob = 0; ob < OBW; ++ob) {
                        const float a = Ws[buf][(ob0 + ob) * 32 + l31][ch * 9 + i * 3 + j];
#pragma unroll
                        for (int rr = 0; rr < RW; ++rr)
                            acc[ob][rr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[i + rr][j], acc[ob][rr], 0, 0, 0);
                    }
                }
        }
        if (has_next) lstore(Xs[buf ^ 1], Ws[buf ^ 1]);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const int gw = w0 + l31;
    if (gw >= W) return;
#if defined(CVF_ABL) && (CVF_ABL & 2)
    if (acc[0][0][0] != 12345.f) return;          // timing-only ablation: no epilogue
#endif
#pragma unroll
    for (int ob = 0; ob < OBW; ++ob)
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int gh = h0 + r0 + rr;
            if (gh >= H) continue;
            if (EPI == EPI_PS_SILU && p.s == 2) {
                // stride-2 PixelShuffle: accumulator registers (reg, reg + 1), reg even, are output channels (o, o + 1) = the two
                // horizontal sub-pixels sj = 0, 1 of one (n, si): one 8-byte store per lane instead of two 4-byte stores 8 bytes
                // apart (the scattered form cost 0.5 ms of the 2.3 ms forward at 720p: tools/probes, CVF_ABL)
                const int Cn = p.O >> 2;
#pragma unroll
                for (int reg = 0; reg < 16; reg += 2) {
                    const int o = o0 + (ob0 + ob) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    if (o >= p.O) continue;                         // O % 4 == 0: o + 1 exists with o
                    float v0 = acc[ob][rr][reg], v1 = acc[ob][rr][reg + 1];
                    if (p.bias) { v0 += p.bias[o]; v1 += p.bias[o + 1]; }
                    const int n = o >> 2, si = (o >> 1) & 1;
                    const size_t idx = (((size_t)b * Cn + n) * (H * 2) + (gh * 2 + si)) * (size_t)(W * 2) + (size_t)gw * 2;
                    if (p.z) *reinterpret_cast<float2 *>(p.z + idx) = make_float2(v0, v1);
                    if (p.out) *reinterpret_cast<float2 *>(p.out + idx) = make_float2(orn_silu_exact(v0), orn_silu_exact(v1));
                }
                continue;
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = o0 + (ob0 + ob) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                if (o >= p.O) continue;
                float v = acc[ob][rr][reg];
                if (p.bias) v += p.bias[o];
                if (EPI == EPI_PLAIN) {
                    p.out[((((size_t)split * p.B + b) * p.O + o) * H + gh) * W + gw] = v;
                } else {
                    const int s = p.s, ss = s * s;
                    const int n = o / ss, rem = o - n * ss, si = rem / s, sj = rem - si * s;
                    const int Cn = p.O / ss;
                    const size_t idx = (((size_t)b * Cn + n) * (H * s) + (gh * s + si)) * (size_t)(W * s) + (gw * s + sj);
                    if (p.z) p.z[idx] = v;
                    if (p.out) p.out[idx] = orn_silu_exact(v);
                }
            }
        }
}

// Small images leave most CUs idle: split the input channels over work-groups (plain epilogue only).
static bool conv_small(int B, int O, int H, int W)
{
    return (long)orn_cdiv(W, CV_TW) * orn_cdiv(H, CV_TH) * orn_cdiv(O, CV_BO) * B < 256;
}

int orn_conv3x3_f32_nsplit(int B, int C, int O, int H, int W)
{
    const int th = conv_small(B, O, H, W) ? 2 : CV_TH;
    const long wgs = (long)orn_cdiv(W, CV_TW) * orn_cdiv(H, th) * orn_cdiv(O, CV_BO) * B;
    int ns = (int)(512 / (wgs > 0 ? wgs : 1));
    const int maxs = orn_cdiv(C, 2 * CV_CC);
    if (ns > maxs) ns = maxs;
    return ns < 1 ? 1 : ns;
}

int orn_launch_conv3x3_f32(const float *x, const float *w, const float *bias, int B, int C, int O, int H, int W,
                           int s, int epi, float *z, float *out, hipStream_t st, float *split_ws)
{
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.out = out; p.z = z;
    p.B = B; p.C = C; p.O = O; p.H = H; p.W = W; p.s = s;
    const bool small = conv_small(B, O, H, W);
    p.tiles_w = orn_cdiv(W, CV_TW);
    p.tiles_h = orn_cdiv(H, small ? 2 : CV_TH);
    p.nsplit = 1; p.c_per_split = C; p.o_first = 0;
    if (epi == EPI_PLAIN && split_ws && !bias) {
        const int ns = orn_conv3x3_f32_nsplit(B, C, O, H, W);
        if (ns > 1) {
            p.c_per_split = orn_cdiv(orn_cdiv(C, ns), CV_CC) * CV_CC;
            p.nsplit = orn_cdiv(C, p.c_per_split);
            p.out = split_ws;
            dim3 grid(p.tiles_w * p.tiles_h, orn_cdiv(O, CV_BO), B * p.nsplit);
            if (small) hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 0>), grid, dim3(256), 0, st, p);
            ORN_LAUNCH_CHECK("conv3x3_f32(split)");
            const size_t n = (size_t)B * O * H * W;
            return orn_launch_reduce_rows(split_ws, p.nsplit, n, n, out, st);
        }
    }
    if (!small && epi == EPI_PLAIN && O % CV_BO == 32) {
        // whole 64-channel tiles, then the last 32 channels on a 32-channel tile
        if (O >= CV_BO) {
            hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 0, 2>), dim3(p.tiles_w * p.tiles_h, O / CV_BO, B), dim3(256), 0, st, p);
            ORN_LAUNCH_CHECK("conv3x3_f32");
        }
        p.o_first = O / CV_BO * CV_BO;
        hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 0, 1>), dim3(p.tiles_w * p.tiles_h, 1, B), dim3(256), 0, st, p);
        ORN_LAUNCH_CHECK("conv3x3_f32(32)");
        return 0;
    }
    dim3 grid(p.tiles_w * p.tiles_h, orn_cdiv(O, CV_BO), B);
    if (epi == EPI_PS_SILU) {
        if (small) hipLaunchKernelGGL((k_conv3x3_f32<EPI_PS_SILU, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv3x3_f32<EPI_PS_SILU, 0>), grid, dim3(256), 0, st, p);
    } else {
        if (small) hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 1>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_conv3x3_f32<EPI_PLAIN, 0>), grid, dim3(256), 0, st, p);
    }
    ORN_LAUNCH_CHECK("conv3x3_f32");
    return 0;
}

extern "C" int orn_conv3x3_ps_silu_fwd(const float *x, const float *wf, const float *bf, int B, int C, int O, int H,
                                       int W, int s, float *z, float *a, void *stream)
{
    ORN_REQUIRE(x && wf && a, "conv3x3_ps_silu_fwd: null pointer");
    ORN_REQUIRE(B > 0 && C > 0 && O > 0 && H > 0 && W > 0 && s > 0, "conv3x3_ps_silu_fwd: bad sizes");
    ORN_REQUIRE(O % (s * s) == 0, "conv3x3_ps_silu_fwd: O=%d not divisible by s*s=%d", O, s * s);
    return orn_launch_conv3x3_f32(x, wf, bf, B, C, O, H, W, s, EPI_PS_SILU, z, a, (hipStream_t)stream, nullptr);
}

// ================================================================================================
// dy[b,o,h,w] = da[b,n,hs+i,ws+j] * SiLU'(z[b,n,hs+i,ws+j]),  o = n*s*s + i*s + j      (+ dbias partials)
// grid: (chunks, O, B); each block sums its chunk -> partial[(b*chunks + chunk)*O + o]
// ================================================================================================
#define DY_PPB 2048
__global__ void __launch_bounds__(256)
k_silu_bwd_unshuffle(const float *__restrict__ da, const float *__restrict__ z, int O, int H, int W, int s,
                     float *__restrict__ dy, float *__restrict__ partial)
{
    __shared__ float sred[16];
    const int o = blockIdx.y, b = blockIdx.z, chunk = blockIdx.x;
    const int ss = s * s, n = o / ss, rem = o - n * ss, si = rem / s, sj = rem - si * s;
    const int Cn = O / ss;
    const size_t HW = (size_t)H * W;
    const size_t Ws = (size_t)W * s;
    const float *dab = da + ((size_t)b * Cn + n) * HW * ss;
    const float *zb = z + ((size_t)b * Cn + n) * HW * ss;
    float *dyb = dy + ((size_t)b * O + o) * HW;
    float sum = 0.f;
    const size_t p0 = (size_t)chunk * DY_PPB;
    for (int i = threadIdx.x; i < DY_PPB; i += 256) {
        const size_t pix = p0 + i;
        if (pix < HW) {
            const int h = (int)(pix / W), w = (int)(pix - (size_t)h * W);
            const size_t src = (size_t)(h * s + si) * Ws + (size_t)(w * s + sj);
            const float v = dab[src] * orn_silu_grad_exact(zb[src]);
            dyb[pix] = v;
            sum += v;
        }
    }
    const float tot = orn_block_sum(sum, sred);
    if (threadIdx.x == 0) partial[((size_t)b * gridDim.x + chunk) * O + o] = tot;
}

// stride 2: output channels (o, o + 1), o even, are the horizontal sub-pixels sj = 0, 1 of one (n, si): one 8-byte load of da and
// of z per thread instead of two 4-byte loads 8 bytes apart.  grid: (chunks, O / 2, B)
__global__ void __launch_bounds__(256)
k_silu_bwd_unshuffle_s2(const float *__restrict__ da, const float *__restrict__ z, int O, int H, int W,
                        float *__restrict__ dy, float *__restrict__ partial)
{
    __shared__ float sred[16];
    const int o = 2 * blockIdx.y, b = blockIdx.z, chunk = blockIdx.x;
    const int n = o >> 2, si = (o >> 1) & 1;
    const int Cn = O >> 2;
    const size_t HW = (size_t)H * W;
    const size_t Ws = (size_t)W * 2;
    const float *dab = da + ((size_t)b * Cn + n) * HW * 4;
    const float *zb = z + ((size_t)b * Cn + n) * HW * 4;
    float *dy0 = dy + ((size_t)b * O + o) * HW, *dy1 = dy0 + HW;
    float sum0 = 0.f, sum1 = 0.f;
    const size_t p0 = (size_t)chunk * DY_PPB;
    for (int i = threadIdx.x; i < DY_PPB; i += 256) {
        const size_t pix = p0 + i;
        if (pix < HW) {
            const int h = (int)(pix / W), w = (int)(pix - (size_t)h * W);
            const size_t src = (size_t)(h * 2 + si) * Ws + (size_t)w * 2;
            const float2 d = *reinterpret_cast<const float2 *>(dab + src);
            const float2 zz = *reinterpret_cast<const float2 *>(zb + src);
            const float v0 = d.x * orn_silu_grad_exact(zz.x), v1 = d.y * orn_silu_grad_exact(zz.y);
            dy0[pix] = v0; dy1[pix] = v1;
            sum0 += v0; sum1 += v1;
        }
    }
    const float t0 = orn_block_sum(sum0, sred);
    __syncthreads();
    const float t1 = orn_block_sum(sum1, sred);
    if (threadIdx.x == 0) {
        partial[((size_t)b * gridDim.x + chunk) * O + o] = t0;
        partial[((size_t)b * gridDim.x + chunk) * O + o + 1] = t1;
    }
}

// ================================================================================================
// Weight transform for dgrad: Wd[c][o][i][j] = Wf[o][c][2-i][2-j]  (dx = conv3x3(dy, Wd, pad 1))
// ================================================================================================
__global__ void k_flip_transpose_w(const float *__restrict__ wf, int O, int C, float *__restrict__ wd)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * C * 9) return;
    const int ij = (int)(idx % 9);
    const long oc = idx / 9;
    const int c = (int)(oc % C), o = (int)(oc / C);
    wd[((long)c * O + o) * 9 + (8 - ij)] = wf[idx];
}

// all layers' dgrad weight transforms in one launch (blockIdx.y = layer): five 5-us graph nodes become one
struct FlipAll { int n; struct { const float *wf; float *wd; int O, C; } l[ORN_MAX_LAYERS]; };
__global__ void k_flip_transpose_w_all(FlipAll a)
{
    const auto &l = a.l[blockIdx.y];
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)l.O * l.C * 9) return;
    const int ij = (int)(idx % 9);
    const long oc = idx / 9;
    const int c = (int)(oc % l.C), o = (int)(oc / l.C);
    l.wd[((long)c * l.O + o) * 9 + (8 - ij)] = l.wf[idx];
}
int orn_launch_flip_transpose_all(int n, const float *const *wf, float *const *wd, const int *O, const int *C, hipStream_t st)
{
    if (n == 0) return 0;
    ORN_REQUIRE(n <= ORN_MAX_LAYERS, "flip_transpose_all: %d layers", n);
    FlipAll a;
    a.n = n;
    long mx = 0;
    for (int i = 0; i < n; ++i) {
        a.l[i].wf = wf[i]; a.l[i].wd = wd[i]; a.l[i].O = O[i]; a.l[i].C = C[i];
        const long w = (long)O[i] * C[i] * 9;
        if (w > mx) mx = w;
    }
    hipLaunchKernelGGL(k_flip_transpose_w_all, dim3(orn_cdiv(mx, 256), n), dim3(256), 0, st, a);
    ORN_LAUNCH_CHECK("flip_transpose_w_all");
    return 0;
}

// ================================================================================================
// wgrad: dW[o, n=(c,ij)] = sum_{b,h,w} dy[b,o,h,w] * x[b,c,h+i-1,w+j-1]
// D rows = o (A operand = dy), D cols = n (B operand = shifted x), K = pixels.
// Work-group tile 64 o x 128 n; K tile = 4 rows x 32 cols of pixels; split-K over pixel tiles with
// per-split partial slabs reduced afterwards in fixed order (deterministic, no atomics).
// ================================================================================================
#define WG_BO 64
#define WG_BN 128
#define WG_TH 4
#define WG_TW 32
#define WG_NPX (WG_TH * WG_TW)
#define WG_MAXC 16
#define WG_XH (WG_TH + 2)
#define WG_XW (WG_TW + 2)

struct WgradP {
    const float *x;    // [B,C,H,W]
    const float *dy;   // [B,O,H,W]
    float *partial;    // [S][O][C*9]
    int B, C, O, H, W;
    int tiles_w, tiles_h, n_ktiles, S, n_ntiles, n_otiles, n_items, items_per_xcd;
};

__global__ void __launch_bounds__(256, (F32_MINB > 3 ? 3 : F32_MINB)) k_wgrad_f32(WgradP p)
{
    __shared__ float Ds[WG_BO][WG_NPX + 1];
    __shared__ float Xs[WG_MAXC][WG_XH][WG_XW];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware decode (work-groups go round the 8 XCDs, each with its own L2): every XCD takes a contiguous range of the
    // (split, o tile, n tile) items with the n tile fastest, so the n tiles that share one dy tile -- and, next, the o tiles that
    // share one x patch -- run on one L2 instead of each fetching it from beyond.
    const int per_xcd = p.items_per_xcd;
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per_xcd || item >= p.n_items) return;
    const int n_on = p.n_otiles * p.n_ntiles;
    const int ksplit = item / n_on, on = item - ksplit * n_on;
    const int ot = on / p.n_ntiles, nt = on - ot * p.n_ntiles;
    const int o0 = ot * WG_BO, n0 = nt * WG_BN;
    const int C = p.C, H = p.H, W = p.W, N = C * 9;
    const int c_lo = n0 / 9;
    const int wo = wave >> 1, wn = wave & 1;

    // this lane's two output columns (n-blocks 2*wn, 2*wn+1) -> per-lane LDS base offsets
    int xoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int n = n0 + (2 * wn + q) * 32 + l31;
        if (n >= N) n = N - 1;                          // clamp (masked at the store)
        const int c = n / 9, ij = n - c * 9, i = ij / 3, j = ij - i * 3;
        xoff[q] = ((c - c_lo) * WG_XH + i) * WG_XW + j + hh;
    }
    const int doff = (wo * 32 + l31) * (WG_NPX + 1) + hh;

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int tiles_per_img = p.tiles_w * p.tiles_h;
    // Round 3: the loads of K tile k+1 are issued into registers before the 128 MFMAs of tile k and stored to LDS behind them
    // (single LDS buffer, two barriers per tile, no exposed memory latency).
    constexpr int DN = (WG_BO * WG_NPX) / 256, XN = (WG_MAXC * WG_XH * WG_XW + 255) / 256;
    float dr[DN], xr[XN];
    // Index arithmetic of the 45 staging loads, hoisted (it cost ~700 vector instructions per K tile, a third of the 128 MFMAs'
    // issue time): dy element idx = t + 256 it is (o = (t >> 7) + 2 it, pixel t & 127), i.e. a per-thread base + it * 2 H W; the
    // x patch offsets (relative to the tile origin) are tile-invariant.  Tiles that touch the image border keep the general form.
    const int dr_r = (t >> 5) & 3, dr_x = t & 31, dr_o = t >> 7;
    int xo[XN];                                           // x patch element -> offset from (channel c_lo, row h0, col w0); INT_MIN: not a patch element / no such channel
#pragma unroll
    for (int it = 0; it < XN; ++it) {
        const int idx = t + it * 256;
        const int c = idx / (WG_XH * WG_XW);
        const int rem2 = idx - c * (WG_XH * WG_XW);
        const int r = rem2 / WG_XW, xx = rem2 - r * WG_XW;
        xo[it] = (idx < WG_MAXC * WG_XH * WG_XW && c_lo + c < C) ? (c * H + r - 1) * W + xx - 1 : INT_MIN;       // (-1 is a real offset: c 0, r 1, xx 0)
    }
    const bool o_full = o0 + WG_BO <= p.O;
    auto gload = [&](const int kt) {
        const int b = kt / tiles_per_img;
        const int rem = kt - b * tiles_per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
        const int h0 = th * WG_TH, w0 = tw * WG_TW;
        const float *xb = p.x + (size_t)b * C * H * W;
        const float *dyb = p.dy + (size_t)b * p.O * H * W;
        if (o_full && h0 >= 1 && h0 + WG_TH + 1 <= H && w0 >= 1 && w0 + WG_TW + 1 <= W) {      // interior tile (wave-uniform)
            const float *dp = dyb + ((size_t)(o0 + dr_o) * H + h0 + dr_r) * W + w0 + dr_x;
            const size_t ostep = (size_t)2 * H * W;
#pragma unroll
            for (int it = 0; it < DN; ++it) dr[it] = dp[(size_t)it * ostep];
            const float *xp = xb + ((size_t)c_lo * H + h0) * W + w0;
#pragma unroll
            for (int it = 0; it < XN; ++it) xr[it] = xo[it] != INT_MIN ? xp[xo[it]] : 0.f;
            return;
        }
#pragma unroll
        for (int it = 0; it < DN; ++it) {
            const int idx = t + it * 256;
            const int o = idx / WG_NPX, px = idx - o * WG_NPX;
            const int r = px / WG_TW, xx = px - r * WG_TW;
            const int gh = h0 + r, gw = w0 + xx;
            dr[it] = 0.f;
            if (o0 + o < p.O && gh < H && gw < W) dr[it] = dyb[((size_t)(o0 + o) * H + gh) * W + gw];
        }
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = t + it * 256;
            const int c = idx / (WG_XH * WG_XW);
            const int rem2 = idx - c * (WG_XH * WG_XW);
            const int r = rem2 / WG_XW, xx = rem2 - r * WG_XW;
            const int gh = h0 + r - 1, gw = w0 + xx - 1, gc = c_lo + c;
            xr[it] = 0.f;
            if (idx < WG_MAXC * WG_XH * WG_XW && gc < C && gh >= 0 && gh < H && gw >= 0 && gw < W)
                xr[it] = xb[((size_t)gc * H + gh) * W + gw];
        }
    };
    if (ksplit < p.n_ktiles) gload(ksplit);
    for (int kt = ksplit; kt < p.n_ktiles; kt += p.S) {
        if (kt != ksplit) __syncthreads();        // everyone is done with the previous tile's LDS image
#pragma unroll
        for (int it = 0; it < DN; ++it) {
            const int idx = t + it * 256;
            const int o = idx / WG_NPX, px = idx - o * WG_NPX;
            Ds[o][px] = dr[it];
        }
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = t + it * 256;
            if (idx < WG_MAXC * WG_XH * WG_XW) (&Xs[0][0][0])[idx] = xr[it];
        }
        __syncthreads();
        if (kt + p.S < p.n_ktiles) gload(kt + p.S);
        const float *dsp = &Ds[0][0] + doff;
        const float *xs0 = &Xs[0][0][0] + xoff[0];
        const float *xs1 = &Xs[0][0][0] + xoff[1];
#pragma unroll
        for (int r = 0; r < WG_TH; ++r)
#pragma unroll
            for (int x2 = 0; x2 < WG_TW; x2 += 2) {
                const float a = dsp[r * WG_TW + x2];
                const float b0 = xs0[r * WG_XW + x2];
                const float b1 = xs1[r * WG_XW + x2];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }
    }

    float *out = p.partial + (size_t)ksplit * p.O * N;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = n0 + (2 * wn + q) * 32 + l31;
        if (n >= N) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = o0 + wo * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (o < p.O) out[(size_t)o * N + n] = acc[q][reg];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Second form of the fp32 wgrad for O % 128 == 0 and C % 32 == 0 (the 96-channel layers): work-group tile 128 o x 288 n (32 input
// channels x 9 taps), wave = 32 o x 288 n = NINE 32x32 accumulator tiles (the 16-bit wgrad's shape).  Per 64-pixel K tile a
// thread stages 49 floats for 288 MFMAs of its wave (first form: 45 for 128), dy is re-read by 3 n tiles instead of 7 and x by 3
// o tiles instead of 6: the first form spent a quarter of its time issuing staging loads (its K loop alone: 1.86 of 2.49 ms).
// ------------------------------------------------------------------------------------------------
#define W2F_BO 128
#define W2F_CN 32                      // input channels per n tile (288 = 32 x 9 output columns)
#define W2F_TH 1                       /* K tile = one row x 32 pixels: 29 staged floats per thread (two rows: 49, and the kernel spills) */
#define W2F_TW 32
#define W2F_NPX (W2F_TH * W2F_TW)
#define W2F_XH (W2F_TH + 2)
#define W2F_XW (W2F_TW + 2)
#define W2F_DLD (W2F_NPX + 1)
__global__ void __launch_bounds__(256, 2) k_wgrad_f32_v2(WgradP p)
{
    __shared__ float Ds[2][W2F_BO][W2F_DLD];            // double-buffered: tile k+1 is stored behind the MFMAs of tile k, one barrier per tile
    __shared__ float Xs[2][W2F_CN][W2F_XH][W2F_XW];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware item decode as in the first form: (split, o tile, n tile), n tile fastest
    const int per_xcd = p.items_per_xcd;
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per_xcd || item >= p.n_items) return;
    const int n_on = p.n_otiles * p.n_ntiles;
    const int ksplit = item / n_on, on = item - ksplit * n_on;
    const int ot = on / p.n_ntiles, nt = on - ot * p.n_ntiles;
    const int o0 = ot * W2F_BO, c_lo = nt * W2F_CN, n0 = c_lo * 9;
    const int C = p.C, H = p.H, W = p.W, N = C * 9;

    // per-lane LDS offsets of this lane's nine output columns n = n0 + 32 q + l31 -> (channel, tap)
    int xoff[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int nr = q * 32 + l31, c = nr / 9, ij = nr - c * 9, i = ij / 3, j = ij - i * 3;
        xoff[q] = (c * W2F_XH + i) * W2F_XW + j + hh;
    }
    const int doff = (wave * 32 + l31) * W2F_DLD + hh;

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    constexpr int DN = (W2F_BO * W2F_NPX) / 256;                              // 32
    constexpr int XN = (W2F_CN * W2F_XH * W2F_XW + 255) / 256;                // 17
    float dr[DN], xr[XN];
    int xo[XN];                              // x patch element -> offset from (channel c_lo, row h0, col w0); INT_MIN: not a patch element
#pragma unroll
    for (int it = 0; it < XN; ++it) {
        const int idx = t + it * 256;
        const int c = idx / (W2F_XH * W2F_XW);
        const int rem = idx - c * (W2F_XH * W2F_XW);
        const int r = rem / W2F_XW, xx = rem - r * W2F_XW;
        xo[it] = idx < W2F_CN * W2F_XH * W2F_XW ? (c * H + r - 1) * W + xx - 1 : INT_MIN;
    }
    const int tiles_per_img = p.tiles_w * p.tiles_h;
    const int d_px = t & (W2F_NPX - 1), d_o = t / W2F_NPX;                    // dy element t + 256 it = (o = d_o + (256 / NPX) it, pixel d_px)
    constexpr int OST = 256 / W2F_NPX;
    auto gload = [&](const int kt) {
        // (an opaque copy of the thread index: the index arithmetic below is invariant over the K loop, and hoisted out of it by the
        // compiler it occupied ~100 registers that the nine accumulator tiles need)
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int d_px = tt & (W2F_NPX - 1), d_o = tt / W2F_NPX;
        const int b = kt / tiles_per_img;
        const int rem = kt - b * tiles_per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
        const int h0 = th * W2F_TH, w0 = tw * W2F_TW;
        const float *xb = p.x + (size_t)b * C * H * W;
        const float *dyb = p.dy + (size_t)b * p.O * H * W;
        if (h0 >= 1 && h0 + W2F_TH + 1 <= H && w0 >= 1 && w0 + W2F_TW + 1 <= W) {      // interior tile (wave-uniform)
            const float *dp = dyb + ((size_t)(o0 + d_o) * H + h0 + (d_px >> 5)) * W + w0 + (d_px & 31);
            const size_t ostep = (size_t)OST * H * W;
#pragma unroll
            for (int it = 0; it < DN; ++it) dr[it] = dp[(size_t)it * ostep];
            const float *xp = xb + ((size_t)c_lo * H + h0) * W + w0;
#pragma unroll
            for (int it = 0; it < XN; ++it) xr[it] = xo[it] != INT_MIN ? xp[xo[it]] : 0.f;
            return;
        }
#pragma unroll
        for (int it = 0; it < DN; ++it) {
            const int gh = h0 + (d_px >> 5), gw = w0 + (d_px & 31);
            dr[it] = (gh < H && gw < W) ? dyb[((size_t)(o0 + d_o + OST * it) * H + gh) * W + gw] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = tt + it * 256;
            const int c = idx / (W2F_XH * W2F_XW);
            const int rem2 = idx - c * (W2F_XH * W2F_XW);
            const int r = rem2 / W2F_XW, xx = rem2 - r * W2F_XW;
            const int gh = h0 + r - 1, gw = w0 + xx - 1;
            xr[it] = 0.f;
            if (idx < W2F_CN * W2F_XH * W2F_XW && gh >= 0 && gh < H && gw >= 0 && gw < W)
                xr[it] = xb[((size_t)(c_lo + c) * H + gh) * W + gw];
        }
    };
    auto lstore = [&](const int bf) {
#pragma unroll
        for (int it = 0; it < DN; ++it) Ds[bf][d_o + OST * it][d_px] = dr[it];
#pragma unroll
        for (int it = 0; it < XN; ++it) {
            const int idx = t + it * 256;
            if (idx < W2F_CN * W2F_XH * W2F_XW) (&Xs[bf][0][0][0])[idx] = xr[it];
        }
    };
    if (ksplit < p.n_ktiles) { gload(ksplit); lstore(0); }
    __syncthreads();
    int bf = 0;
    for (int kt = ksplit; kt < p.n_ktiles; kt += p.S, bf ^= 1) {
        const bool has_next = kt + p.S < p.n_ktiles;
        if (has_next) gload(kt + p.S);
        const float *dsp = &Ds[bf][0][0] + doff;
        const float *xs = &Xs[bf][0][0][0];
#pragma unroll
        for (int r = 0; r < W2F_TH; ++r)
#pragma unroll 4
            for (int x2 = 0; x2 < W2F_TW; x2 += 2) {
                const float a = dsp[r * W2F_TW + x2];
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[xoff[q] + r * W2F_XW + x2], acc[q], 0, 0, 0);
            }
        if (has_next) lstore(bf ^ 1);
        __syncthreads();
    }
    float *out = p.partial + (size_t)ksplit * p.O * N;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int n = n0 + q * 32 + l31;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = o0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            out[(size_t)o * N + n] = acc[q][reg];
        }
    }
}

static bool wgrad_v2_ok(int C, int O)
{
    static const bool off = orn_probe_env("ORN_F32_WGRAD_V1") != nullptr;      // tools/probes A/B
    return !off && O % W2F_BO == 0 && C % W2F_CN == 0;
}

static int wgrad_split(int B, int C, int O, int H, int W)
{
    if (wgrad_v2_ok(C, O)) {
        const int n_ktiles = B * orn_cdiv(H, W2F_TH) * orn_cdiv(W, W2F_TW);
        const int tiles = (O / W2F_BO) * (C / W2F_CN);
        static const int s_env = orn_probe_env_int("ORN_F32_WGRAD_WGS", 512);     // tools/probes sweep
        int S = s_env / tiles;                             // one full round of two work-groups per CU
        if (S > n_ktiles) S = n_ktiles;
        return S < 1 ? 1 : S;
    }
    const int n_ktiles = B * orn_cdiv(H, WG_TH) * orn_cdiv(W, WG_TW);
    const int tiles = orn_cdiv(O, WG_BO) * orn_cdiv(C * 9, WG_BN);
    int S = orn_cdiv(1024, tiles);
    if (S > n_ktiles) S = n_ktiles;
    if (S < 1) S = 1;
    return S;
}

static int dbp_rows(int H, int W)
{
    const int a = orn_cdiv((long)H * W, DY_PPB), b = orn_head_bwd_fused_f32_blocks(H, W);
    return a > b ? a : b;
}

extern "C" size_t orn_conv3x3_ps_silu_bwd_ws_bytes(int B, int C, int O, int H, int W)
{
    const size_t HW = (size_t)H * W;
    const int chunks = orn_cdiv((long)HW, DY_PPB);
    size_t f = 0;
    f += orn_align((size_t)B * O * HW * 4) / 4;                         // dy
    f += orn_align((size_t)O * C * 9 * 4) / 4;                          // Wd
    f += orn_align((size_t)wgrad_split(B, C, O, H, W) * O * C * 9 * 4) / 4;   // wgrad partial slabs
    f += orn_align((size_t)B * dbp_rows(H, W) * O * 4) / 4;             // dbias partials (the producer with more blocks: fused head backward)
    f += orn_align((size_t)orn_conv3x3_f32_nsplit(B, O, C, H, W) * B * C * HW * 4) / 4;   // dgrad channel-split slabs
    return f * 4;
}

int orn_launch_conv_bwd_f32(const float *x, const float *wf, const float *z, const float *da, int B, int C, int O,
                            int H, int W, int s, float *dx, float *dwf, float *dbf, float *ws, hipStream_t st,
                            const OrnHeadBwdFuse *head, const float *wd_ready)
{
    const size_t HW = (size_t)H * W;
    const int chunks = orn_cdiv((long)HW, DY_PPB);
    const int S = wgrad_split(B, C, O, H, W);
    float *dy = ws;
    float *wd = dy + orn_align((size_t)B * O * HW * 4) / 4;
    float *slabs = wd + orn_align((size_t)O * C * 9 * 4) / 4;
    float *dbp = slabs + orn_align((size_t)S * O * C * 9 * 4) / 4;
    float *dgs = dbp + orn_align((size_t)B * dbp_rows(H, W) * O * 4) / 4;

    int dbp_n = B * chunks;                           // rows of dbias partials
    if (head) {
        // the last block of the fp32 engine: head backward, SiLU' and the un-shuffle in one pass (dy and the dbias partials land where
        // k_silu_bwd_unshuffle would have put them; it has more, smaller blocks: dbp_rows)
        ORN_REQUIRE(B == 1 && s == 2, "conv_bwd_f32: fused head backward needs B == 1, s == 2");
        dbp_n = orn_head_bwd_fused_f32_blocks(H, W);
        ORN_TRY(orn_launch_head_bwd_fused_f32(z, head->w, head->out, head->dout, O / 4, H, W, head->sigmoid, dy, dbp, head->dw, head->db,
                                              head->hws, st));
    } else {
        if (s == 2) hipLaunchKernelGGL(k_silu_bwd_unshuffle_s2, dim3(chunks, O / 2, B), dim3(256), 0, st, da, z, O, H, W, dy, dbp);
        else hipLaunchKernelGGL(k_silu_bwd_unshuffle, dim3(chunks, O, B), dim3(256), 0, st, da, z, O, H, W, s, dy, dbp);
        ORN_LAUNCH_CHECK("silu_bwd_unshuffle");
    }
    ORN_TRY(orn_launch_reduce_rows(dbp, dbp_n, (size_t)O, (size_t)O, dbf, st));

    WgradP p;
    p.x = x; p.dy = dy; p.partial = slabs;
    p.B = B; p.C = C; p.O = O; p.H = H; p.W = W;
    if (wgrad_v2_ok(C, O)) {
        p.tiles_w = orn_cdiv(W, W2F_TW); p.tiles_h = orn_cdiv(H, W2F_TH);
        p.n_ktiles = B * p.tiles_w * p.tiles_h;
        p.S = S;
        p.n_ntiles = C / W2F_CN; p.n_otiles = O / W2F_BO;
        p.n_items = p.n_otiles * p.n_ntiles * S;
        p.items_per_xcd = orn_cdiv(p.n_items, 8);
        hipLaunchKernelGGL(k_wgrad_f32_v2, dim3(p.items_per_xcd * 8), dim3(256), 0, st, p);
        ORN_LAUNCH_CHECK("wgrad_f32_v2");
    } else {
    p.tiles_w = orn_cdiv(W, WG_TW); p.tiles_h = orn_cdiv(H, WG_TH);
    p.n_ktiles = B * p.tiles_w * p.tiles_h;
    p.S = S;
    p.n_ntiles = orn_cdiv(C * 9, WG_BN);
    p.n_otiles = orn_cdiv(O, WG_BO);
    p.n_items = p.n_otiles * p.n_ntiles * S;
    p.items_per_xcd = orn_cdiv(p.n_items, 8);
    hipLaunchKernelGGL(k_wgrad_f32, dim3(p.items_per_xcd * 8), dim3(256), 0, st, p);
    ORN_LAUNCH_CHECK("wgrad_f32");
    }
    ORN_TRY(orn_launch_reduce_rows(slabs, S, (size_t)O * C * 9, (size_t)O * C * 9, dwf, st));

    if (dx) {
        if (wd_ready) wd = const_cast<float *>(wd_ready);      // (the engine flips every layer's kernel in one launch)
        else hipLaunchKernelGGL(k_flip_transpose_w, dim3(orn_cdiv((long)O * C * 9, 256)), dim3(256), 0, st, wf, O, C, wd);
        ORN_LAUNCH_CHECK("flip_transpose_w");
        ORN_TRY(orn_launch_conv3x3_f32(dy, wd, nullptr, B, O, C, H, W, 1, EPI_PLAIN, nullptr, dx, st, dgs));
    }
    return 0;
}

extern "C" int orn_conv3x3_ps_silu_bwd(const float *x, const float *wf, const float *z, const float *da, int B,
                                       int C, int O, int H, int W, int s, float *dx, float *dwf, float *dbf,
                                       void *ws, size_t ws_bytes, void *stream)
{
    ORN_REQUIRE(x && wf && z && da && dwf && dbf && ws, "conv3x3_ps_silu_bwd: null pointer");
    ORN_REQUIRE(B > 0 && C > 0 && O > 0 && H > 0 && W > 0 && s > 0 && O % (s * s) == 0, "conv3x3_ps_silu_bwd: bad sizes");
    if (ws_bytes < orn_conv3x3_ps_silu_bwd_ws_bytes(B, C, O, H, W)) {
        orn_set_error("conv3x3_ps_silu_bwd: workspace %zu < %zu", ws_bytes, orn_conv3x3_ps_silu_bwd_ws_bytes(B, C, O, H, W));
        return ORN_E_WS;
    }
    return orn_launch_conv_bwd_f32(x, wf, z, da, B, C, O, H, W, s, dx, dwf, dbf, (float *)ws, (hipStream_t)stream);
}
