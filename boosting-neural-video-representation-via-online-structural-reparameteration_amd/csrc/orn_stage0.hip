// First NeRV block of a 16-bit engine, fused (reference: NeRVBlock.forward model.py:518-567 on the fc_h x fc_w stem image).
//
// The block that follows the stem works on a 9 x 16 image (0.04 GF of the 202 GF forward) and stays in fp32.  Run through
// the general fp32 kernels it cost 26 us forward (conv + layout hand-off) and 50 us backward (eight launches: layout
// hand-off, SiLU'/unshuffle, two split-K GEMMs and their reductions, weight flip, dgrad) -- 5 % of the 720p step, all of
// it launch floors and exposed latencies.  Here the whole image, a 16-row slice of the merged kernel and of the gradient
// live in LDS, `v_mfma_f32_16x16x4_f32` for the three GEMMs:
//   forward : one work-group per 16 consecutive conv channels; conv3x3 + bias + PixelShuffle + SiLU, written straight into
//             the next block's 16-bit channels-last input; z kept as [sub-position][pixel][channel] for the backward
//   backward: one work-group per (PixelShuffle sub-position, 16 post-shuffle channels): sum of the next block's fp32 dgrad
//             slabs x SiLU'(z) (both contiguous per pixel in that split), dbias, dW (no split-K: K = H*W pixels) and the
//             group's share of dx, left as per-work-group slabs that the stem's first backward kernel sums (fixed order).
// fp32 products and accumulation throughout (same arithmetic class as the kernels it replaces; summation order differs).
#include "orn_internal.h"
#include "orn_prep_rider.h"
#include "orn_merge_pack.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifdef ORN_CONV_ABLATE
__device__ unsigned long long g_s0_diag[32];
#define S0_STAMP(k_) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_s0_diag[k_] = wall_clock64(); } while (0)
extern "C" int orn_stage0_diag(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s0_diag), sizeof(g_s0_diag)); }
#else
#define S0_STAMP(k_)
#endif

namespace {

struct Stage0P {
    const float *x;      // [C][H][W] block input (stem output)
    const float *wf;     // [O][C][3][3] merged kernel
    const float *bf;     // [O]
    int C, C16, O, H, W, s;         // C16: C rounded up to 16 (LDS images are zero-padded to it)
    unsigned mXP, mXW, mWN, mC16, mW, mSS, mS;   // exact division by multiply-high (s0_div): a runtime integer division is ~40
                                                 // VALU instructions, and the LDS fills did ~100 of them per thread (10 of 15 us)
    float *z;            // [s*s][H*W][O/s^2] pre-activation (layout private to this file); null: not kept (decode)
    void *xpad_next;     // 16-bit [H*s+2][W*s+2][Cp]: interior, channels [0, O/s^2)
    int Cp;
    const float *dxn;    // [nslab][H*s*W*s][Cp] gradient wrt the block output, times 1/inv_gs
    int nslab;
    float inv_gs;
    const OrnScaleState *sc;   // optional: 1/scale from the device-side loss-scale state
    float *dwf, *dbf;    // [O][C][3][3], [O]
    float *dx_slabs;     // [gridDim.x][C][H*W]
    int fwd_blocks;      // forward launch: work-groups of the block itself; the ones behind them are prep riders
    OrnPrepRider rider;  // 16-bit operand copies of the later blocks' merged kernels (orn_prep_rider.h); n == 0: none
    int prep_blocks;     // rider work-groups of `rider`; behind them: the merge backward's T -> Th copies (orn_merge_pack.h)
};

// x / d for 0 <= x < 2^16, 2 <= d < 2^16 with m = ceil(2^32 / d); d == 1 is m = 0
__device__ __forceinline__ int s0_div(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

// LDS fills in two halves -- batch_load issues U unconditional global loads per thread (a pointer that is always safe
// to read; ok = false marks a zero), batch_store writes them -- so that a kernel can put EVERY operand's loads in flight
// before it waits once.  (`for (i = t; i < N; i += nt) lds[i] = global[f(i)]` exposes one memory latency per iteration,
// and these operands were written by the kernels just before: each miss goes past the L2.)
template <int U, typename F>
__device__ __forceinline__ void batch_load(float (&v)[U], bool (&ok)[U], int base, int N, F src)
{
    const int t = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        int i = base + t + k * nt;
        i = i < N ? i : N - 1;
        v[k] = *src(i, ok[k]);
    }
}
template <int U>
__device__ __forceinline__ void batch_store(const float (&v)[U], const bool (&ok)[U], float *dst, int base, int N)
{
    const int t = threadIdx.x, nt = blockDim.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int i = base + t + k * nt;
        if (i < N) dst[i] = ok[k] ? v[k] : 0.f;
    }
}
#define S0_UX 11      // batch sizes that cover the 720p block (C = 26, 9 x 16) in one round; larger blocks loop
#define S0_UW 9

// zero-bordered input [C16][H+2][W+2] (channels >= C are zeros: the K loops run in whole batches of four k-steps)
#define S0_SRC_X                                                                                                \
    [&](int i, bool &ok) {                                                                                      \
        const int c = s0_div(i, p.mXP), r = i - c * XP, hh = s0_div(r, p.mXW), ww = r - hh * XW;                \
        ok = c < p.C && hh >= 1 && hh <= H && ww >= 1 && ww <= W;                                               \
        return p.x + (ok ? (c * H + hh - 1) * W + ww - 1 : 0);                                                  \
    }
// 16 output channels of the kernel as [n][tap][C16] (+4 floats per n: the pad slots are written as zeros)
#define S0_SRC_W                                                                                                \
    [&](int i, bool &ok) {                                                                                      \
        const int n = s0_div(i, p.mWN), r = i - n * WN, tap = s0_div(r, p.mC16), c = r - tap * C16;             \
        ok = tap < 9 && o0 + n < p.O && c < p.C;                                                                \
        return p.wf + (ok ? ((size_t)(o0 + n) * p.C + c) * 9 + tap : 0);                                        \
    }

template <typename H16>
__global__ void __launch_bounds__(1024) k_stage0_fwd(Stage0P p, MhPackAll pack)
{
    ORN_PRIO_HIGH();
    extern __shared__ float sm[];
    if ((int)blockIdx.x >= p.fwd_blocks) {            // riders: nothing in this launch depends on them
        const int rb = (int)blockIdx.x - p.fwd_blocks;
        if (rb < p.prep_blocks) orn_prep_rider_block<H16>(p.rider, rb, sm);
        else {
            int layer, pjob;
            const int blk = mh_pack_decode(pack, MH_TAB_T, rb - p.prep_blocks, layer, pjob);
            mh_pack_block(pack, layer, pjob, blk, reinterpret_cast<float (*)[65]>(sm));
        }
        return;
    }
    const int C16 = p.C16, H = p.H, W = p.W, HW = H * W, XW = W + 2, XP = (H + 2) * XW, WN = 9 * C16 + 4;
    float *xs = sm, *ws_ = sm + C16 * XP;
    const int nt = blockDim.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o0 = blockIdx.x * 16;
    const int m = lane & 15, kq = lane >> 4;
    const int o = o0 + m;
    S0_STAMP(0);
    {
        float vx[S0_UX], vw[S0_UW];
        bool okx[S0_UX], okw[S0_UW];
        batch_load(vx, okx, 0, C16 * XP, S0_SRC_X);
        batch_load(vw, okw, 0, 16 * WN, S0_SRC_W);
        batch_store(vx, okx, xs, 0, C16 * XP);
        batch_store(vw, okw, ws_, 0, 16 * WN);
        for (int base = S0_UX * nt; base < C16 * XP; base += S0_UX * nt) {
            batch_load(vx, okx, base, C16 * XP, S0_SRC_X);
            batch_store(vx, okx, xs, base, C16 * XP);
        }
        for (int base = S0_UW * nt; base < 16 * WN; base += S0_UW * nt) {
            batch_load(vw, okw, base, 16 * WN, S0_SRC_W);
            batch_store(vw, okw, ws_, base, 16 * WN);
        }
    }
    const float b = (p.bf && o < p.O) ? p.bf[o] : 0.f;   // in flight under the GEMM
    S0_STAMP(1);
    __syncthreads();
    S0_STAMP(2);
    // D[pixel][channel]: A = patch values (row = pixel, k = input channel), B = kernel (k, col = output channel)
    int pa = wave * 16 + m;
    if (pa >= HW) pa = HW - 1;                        // clamped rows are never stored
    const int ph = s0_div(pa, p.mW), pw = pa - ph * W;
    const float *xa = xs + kq * XP + ph * XW + pw;
    const float *wb = ws_ + m * WN + kq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const int ti = tap / 3, tj = tap - ti * 3;
        const float *xt = xa + ti * XW + tj, *wt = wb + tap * C16;
        for (int c0 = 0; c0 < C16; c0 += 16) {        // four k-steps per batch: the LDS reads of a batch issue together
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = xt[(c0 + 4 * u) * XP]; b[u] = wt[c0 + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
        }
    }
    const int px0 = wave * 16 + 4 * kq;
    S0_STAMP(3);
    if (o >= p.O || px0 >= HW) return;                // HW % 4 == 0: a lane's four pixels are in or out together
    const float v[4] = {acc[0] + b, acc[1] + b, acc[2] + b, acc[3] + b};
    const int s = p.s, ss = s * s, n = s0_div(o, p.mSS), rem = o - n * ss, si = s0_div(rem, p.mS), sj = rem - si * s;
    if (p.z) {           // grouped by sub-position, channels innermost: what one backward work-group reads is contiguous
        const int Cn = p.O / ss;
#pragma unroll
        for (int r = 0; r < 4; ++r) p.z[((size_t)rem * HW + px0 + r) * Cn + n] = v[r];
    }
    H16 *dst = reinterpret_cast<H16 *>(p.xpad_next);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int px = px0 + r, h = s0_div(px, p.mW), w = px - h * W;
        dst[((size_t)(h * s + si + 1) * (W * s + 2) + (w * s + sj + 1)) * p.Cp + n] = (H16)orn_silu_exact(v[r]);
    }
    S0_STAMP(4);
}

// One work-group = one PixelShuffle sub-position (si, sj) x 16 consecutive post-shuffle channels n: its 16 conv channels
// o = n*s^2 + si*s + sj.  For a fixed pixel those 16 values are contiguous in the next block's channels-last dgrad slabs
// and in z, so the gather costs 64 B per line instead of 4 (with 16 consecutive o per work-group it was 374 K cache-line
// requests from 41 CUs: ~17 us).
__global__ void __launch_bounds__(1024) k_stage0_bwd(Stage0P p)
{
    ORN_PRIO_HIGH();
    extern __shared__ float sm[];
    const int C = p.C, C16 = p.C16, H = p.H, W = p.W, HW = H * W, XW = W + 2, XP = (H + 2) * XW, WN = 9 * C16 + 4;
    float *xs = sm, *ws_ = sm + C16 * XP, *dys = ws_ + 16 * WN;       // dys: [16][H+2][W+2], zero border
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = nt >> 6;
    const int s = p.s, ss = s * s, Cn = p.O / ss;
    const int g = blockIdx.x / ss, sub = blockIdx.x - g * ss, si = sub / s, sj = sub - si * s;
    const int n0 = g * 16;                                            // rows r -> channel n0 + r, conv channel (n0 + r)*ss + sub
    const float inv_gs = p.sc ? p.sc->inv_gs : p.inv_gs;
    S0_STAMP(8);
    {
        float vx[S0_UX], vw[S0_UW];
        bool okx[S0_UX], okw[S0_UW];
        batch_load(vx, okx, 0, C16 * XP, S0_SRC_X);
        batch_load(vw, okw, 0, 16 * WN, [&](int i, bool &ok) {
            const int r16 = s0_div(i, p.mWN), r = i - r16 * WN, tap = s0_div(r, p.mC16), c = r - tap * C16;
            ok = tap < 9 && n0 + r16 < Cn && c < C;
            return p.wf + (ok ? ((size_t)((n0 + r16) * ss + sub) * C + c) * 9 + tap : 0);
        });
        // dy = (sum of the next block's dgrad slabs) / gs * SiLU'(z): element e = pixel * 16 + r
        const int Ws = W * s;
        const size_t slab = (size_t)H * s * Ws * p.Cp;
        constexpr int U = 4;
        for (int base = 0; base < 16 * HW; base += U * nt) {
            float gsum[U], zz[U];
            size_t q[U];
            bool ok[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                int e = base + t + k * nt;
                e = e < 16 * HW ? e : 16 * HW - 1;
                const int px = e >> 4, r16 = e & 15, h = s0_div(px, p.mW), w = px - h * W;
                ok[k] = n0 + r16 < Cn;
                q[k] = ok[k] ? ((size_t)(h * s + si) * Ws + w * s + sj) * p.Cp + n0 + r16 : 0;
                zz[k] = p.z[ok[k] ? ((size_t)sub * HW + px) * Cn + n0 + r16 : 0];
                gsum[k] = 0.f;
            }
            for (int sl = 0; sl < p.nslab; ++sl) {
#pragma unroll
                for (int k = 0; k < U; ++k) gsum[k] += p.dxn[sl * slab + q[k]];
            }
            if (base == 0) {          // the operand images' loads were issued first: their stores cost no extra wait
                batch_store(vx, okx, xs, 0, C16 * XP);
                batch_store(vw, okw, ws_, 0, 16 * WN);
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int e = base + t + k * nt;
                if (e < 16 * HW) {
                    const int px = e >> 4, r16 = e & 15, h = s0_div(px, p.mW), w = px - h * W;
                    dys[r16 * XP + (h + 1) * XW + w + 1] = ok[k] ? gsum[k] * inv_gs * orn_silu_grad_exact(zz[k]) : 0.f;
                }
            }
        }
        for (int i = t; i < 16 * XP; i += nt) {       // zero border of dys
            const int r16 = s0_div(i, p.mXP), r = i - r16 * XP, hh = s0_div(r, p.mXW), ww = r - hh * XW;
            if (hh == 0 || hh == H + 1 || ww == 0 || ww == W + 1) dys[i] = 0.f;
        }
        for (int base = S0_UX * nt; base < C16 * XP; base += S0_UX * nt) {
            batch_load(vx, okx, base, C16 * XP, S0_SRC_X);
            batch_store(vx, okx, xs, base, C16 * XP);
        }
        for (int base = S0_UW * nt; base < 16 * WN; base += S0_UW * nt) {
            batch_load(vw, okw, base, 16 * WN, [&](int i, bool &ok) {
                const int r16 = s0_div(i, p.mWN), r = i - r16 * WN, tap = s0_div(r, p.mC16), c = r - tap * C16;
                ok = tap < 9 && n0 + r16 < Cn && c < C;
                return p.wf + (ok ? ((size_t)((n0 + r16) * ss + sub) * C + c) * 9 + tap : 0);
            });
            batch_store(vw, okw, ws_, base, 16 * WN);
        }
    }
    S0_STAMP(9);
    __syncthreads();
    S0_STAMP(10);
    const int m = lane & 15, kq = lane >> 4;
    // ---- dbias: row sums of dy --------------------------------------------------------------------
    for (int r16 = wave; r16 < 16; r16 += nw) {
        float sum = 0.f;
        for (int px = lane; px < HW; px += 64) {
            const int h = s0_div(px, p.mW);
            sum += dys[r16 * XP + (h + 1) * XW + px - h * W + 1];
        }
        sum = orn_wave_sum(sum);
        if (lane == 0 && n0 + r16 < Cn) p.dbf[(n0 + r16) * ss + sub] = sum;
    }
    S0_STAMP(11);
    // ---- dW[o][c][tap] = sum_px dy[o][px] * x[c][px + off(tap)]: D[row][j], j = c*9 + tap (memory order), K = pixels ----
    const int NJ = C * 9;
    for (int jb = wave; jb * 16 < NJ; jb += nw) {
        const int j = jb * 16 + m, jc = j < NJ ? j : NJ - 1;
        const int c = jc / 9, tap = jc - c * 9, ti = tap / 3, tj = tap - ti * 3;
        const float *xb = xs + c * XP + ti * XW + tj;
        const float *da = dys + m * XP + XW + 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int ph = 0, pw = kq;                          // W >= 4
        for (int ks = 0; ks < HW / 4; ks += 4) {      // HW % 16 == 0
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = ph * XW + pw;
                a[u] = da[off]; b[u] = xb[off];
                pw += 4;
                if (pw >= W) { pw -= W; ++ph; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
        }
        if (j < NJ) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * kq + r;
                if (n < Cn) p.dwf[(size_t)(n * ss + sub) * NJ + j] = acc[r];
            }
        }
    }
    S0_STAMP(12);
    // ---- this slice's share of dx[c][h][w] = sum_{o,i,j} dy[o][h-i+1][w-j+1] * Wf[o][c][i][j]: D[pixel][c], K = (tap, row) ----
    int pa = wave * 16 + m;
    if (pa >= HW) pa = HW - 1;
    const int ph = s0_div(pa, p.mW), pw = pa - ph * W;
    const int px0 = wave * 16 + 4 * kq;
    float *slab = p.dx_slabs + (size_t)blockIdx.x * C * HW;
    for (int cb = 0; cb < C16 / 16; ++cb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ti = tap / 3, tj = tap - ti * 3;
            const float *da = dys + kq * XP + (ph - ti + 2) * XW + (pw - tj + 2);
            const float *wb = ws_ + kq * WN + tap * C16 + cb * 16 + m;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da[r0 * XP], wb[r0 * WN], acc, 0, 0, 0);
        }
        const int c = cb * 16 + m;
        if (c < C && px0 < HW) *reinterpret_cast<float4 *>(slab + (size_t)c * HW + px0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    S0_STAMP(13);
}

size_t smem_bytes(int C16, int H, int W, bool bwd)
{
    const size_t XP = (size_t)(H + 2) * (W + 2);
    return (C16 * XP + 16 * (size_t)(9 * C16 + 4) + (bwd ? 16 * XP : 0)) * sizeof(float);
}

int fill(Stage0P &p, const float *x, const float *wf, const float *bf, int C, int O, int H, int W, int s)
{
    ORN_REQUIRE(orn_stage0_supported(C, O, H, W, s), "stage0: unsupported C=%d O=%d H=%d W=%d s=%d", C, O, H, W, s);
    p = Stage0P{};
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d); };
    p.mXP = magic((H + 2) * (W + 2)); p.mXW = magic(W + 2); p.mWN = magic(9 * ((C + 15) / 16 * 16) + 4);
    p.mC16 = magic((C + 15) / 16 * 16); p.mW = magic(W); p.mSS = magic(s * s); p.mS = magic(s);
    p.x = x; p.wf = wf; p.bf = bf; p.C = C; p.C16 = (C + 15) / 16 * 16; p.O = O; p.H = H; p.W = W; p.s = s;
    return 0;
}

template <typename K>
int set_smem(K kern, size_t bytes)
{
    if (bytes <= 65536) return 0;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { orn_set_error("stage0: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // namespace

// H*W pixels as whole 16-pixel blocks, one wave each (<= 16 waves)
bool orn_stage0_supported(int C, int O, int H, int W, int s)
{
    return C >= 1 && C <= 64 && O >= 1 && s >= 1 && O % (s * s) == 0 && W >= 4 && H * W <= 256 && (H * W) % 16 == 0 &&
           smem_bytes((C + 15) / 16 * 16, H, W, true) <= 160 * 1024;
}

// backward work-groups (= dx partial slabs): sub-positions x groups of 16 post-shuffle channels
int orn_stage0_slabs(int O, int s) { return s * s * orn_cdiv(O / (s * s), 16); }

// precision: 1 bf16, 2 IEEE half (element type of xpad_next)
int orn_launch_stage0_fwd(const float *x, const float *wf, const float *bf, int C, int O, int H, int W, int s, float *z,
                          void *xpad_next, int Cp, int precision, hipStream_t st, int n_prep, const OrnPrepLayer *prep,
                          const void *pack, int pack_t_blocks)
{
    Stage0P p;
    ORN_TRY(fill(p, x, wf, bf, C, O, H, W, s));
    ORN_REQUIRE(n_prep >= 0 && n_prep <= ORN_MAX_LAYERS, "stage0_fwd: bad rider count");
    int cmax = 0;
    p.rider.n = n_prep;
    p.rider.blk_start[0] = 0;
    for (int i = 0; i < n_prep; ++i) {
        auto &l = p.rider.l[i];
        l.wf = prep[i].wf; l.bf = prep[i].bf; l.O = prep[i].O; l.C = prep[i].C; l.Cp = prep[i].Cp > 0 ? prep[i].Cp : prep[i].C;
        l.s2 = prep[i].s * prep[i].s; l.Cn = prep[i].O / l.s2; l.wb = prep[i].wb; l.wd = prep[i].wd; l.biasp = prep[i].biasp;
        p.rider.blk_start[i + 1] = p.rider.blk_start[i] + orn_cdiv(l.O, ORN_PREP_ROWS);
        if (l.C > cmax) cmax = l.C;
    }
    ORN_REQUIRE(xpad_next && O / (s * s) <= Cp && (precision == 1 || precision == 2), "stage0_fwd: bad output arguments");
    p.z = z; p.xpad_next = xpad_next; p.Cp = Cp;
    size_t smem = smem_bytes(p.C16, H, W, false);
    if (n_prep > 0 && orn_prep_rider_lds_bytes(cmax) > smem) smem = orn_prep_rider_lds_bytes(cmax);
    p.fwd_blocks = orn_cdiv(O, 16);
    p.prep_blocks = p.rider.blk_start[n_prep];
    MhPackAll pk = {};
    if (pack && pack_t_blocks > 0) pk = *(const MhPackAll *)pack; else pack_t_blocks = 0;
    const dim3 grid(p.fwd_blocks + p.prep_blocks + pack_t_blocks), block(orn_cdiv(H * W, 16) * 64);
    if (precision == 2) {
        ORN_TRY(set_smem(k_stage0_fwd<_Float16>, smem));
        hipLaunchKernelGGL(k_stage0_fwd<_Float16>, grid, block, smem, st, p, pk);
    } else {
        ORN_TRY(set_smem(k_stage0_fwd<__bf16>, smem));
        hipLaunchKernelGGL(k_stage0_fwd<__bf16>, grid, block, smem, st, p, pk);
    }
    ORN_LAUNCH_CHECK("stage0_fwd");
    return 0;
}

// dxn: the next block's fp32 dgrad slabs [nslab][H*s][W*s][Cp] (times 1/inv_gs); z: what orn_launch_stage0_fwd kept;
// slabs: orn_stage0_slabs(O, s) * C*H*W floats of scratch; dx [C][H][W] (null: leave the slabs to the caller), dwf, dbf
// are overwritten.
int orn_launch_stage0_bwd(const float *x, const float *wf, const float *z, const float *dxn, int nslab, int Cp, float inv_gs, int C,
                          int O, int H, int W, int s, float *slabs, float *dx, float *dwf, float *dbf, hipStream_t st, const OrnScaleState *sc)
{
    Stage0P p;
    ORN_TRY(fill(p, x, wf, nullptr, C, O, H, W, s));
    ORN_REQUIRE(z && dxn && nslab >= 1 && slabs && dwf && dbf, "stage0_bwd: null pointer");
    p.z = const_cast<float *>(z); p.dxn = dxn; p.nslab = nslab; p.Cp = Cp; p.inv_gs = inv_gs; p.sc = sc; p.dwf = dwf; p.dbf = dbf; p.dx_slabs = slabs;
    const size_t smem = smem_bytes(p.C16, H, W, true);
    ORN_TRY(set_smem(k_stage0_bwd, smem));
    const int nwg = orn_stage0_slabs(O, s);
    hipLaunchKernelGGL(k_stage0_bwd, dim3(nwg), dim3(orn_cdiv(H * W, 16) * 64), smem, st, p);
    ORN_LAUNCH_CHECK("stage0_bwd");
    if (!dx) return 0;                 // the consumer (orn_launch_stem_bwd with dh2_nslab) sums the slabs itself
    const size_t n = (size_t)C * H * W;
    return orn_launch_reduce_rows(slabs, nwg, n, n, dx, st);
}
