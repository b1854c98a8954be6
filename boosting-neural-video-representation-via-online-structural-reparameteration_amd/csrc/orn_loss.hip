// A7 + A10  loss_fn (utils.py:139-189: L2, L1, Fusion6 = 0.7*L1 + 0.3*(1-SSIM)) fused with its
// gradient and with psnr_fn (utils.py:191-199).  HBM-bound on 11 MB planes; two tiled passes:
//   k_ssim_stats : 11-tap separable Gaussian (sigma 1.5, valid) of p, t, p^2, t^2, pt in LDS ->
//                  SSIM map value (summed) and dS/dm, dS/dq, dS/dr maps
//   k_loss_grad  : adjoint (full) filter of the three maps + L1/L2 terms -> dL/dpred, |p-t| and
//                  (p-t)^2 sums
// followed by a fixed-order finalize.  SSIM follows pytorch_msssim 0.2.1's published algorithm
// (the package is not in the reference tree: parity unpinned, see oracle/cpu_ref.py).
#include "orn_internal.h"

#define SS_TH 16
#define SS_TW 64
#define SS_PH (SS_TH + 10)
#define SS_PW (SS_TW + 10)
#define SS_PWP 76            // padded row stride (floats): 16-byte aligned rows for float4 LDS reads

__constant__ float c_gauss[11];

struct LossP {
    const float *pred, *target;
    const int *frame_idx;      // optional device index selecting the target frame
    size_t frame_stride;
    int planes, H, W, Hv, Wv;  // planes = B*Ch
    float *part_ssim;          // [n_blocks]
    float *part_l1;            // [n_blocks][2]
    float *dpred;              // may be null
    int loss_type;
    int vec4;                  // W % 4 == 0 and 16-byte aligned planes: the fused Fusion6 kernel loads float4
    // Fusion6, optional: the TARGET's own filtered maps G*t and G*t^2 on the valid map, [frame][2][planes][Hv][Wv], computed
    // once per video by k_fusion6<.., TC_FILL> with the arithmetic of the in-kernel form (same taps, same fmaf order: the
    // step's results do not change by a bit).  The target never changes during a fit, and 288 GB of HBM hold 2.9 GB of
    // them for a 132-frame 720p video: the step then filters three maps (p, p^2, pt) instead of five.
    float *tstats;
    size_t tstats_stride;      // floats per frame
    float g_l1, g_l2, g_ssim;  // gradient scales (already include loss_scale and 1/n)
    int tiles_w, tiles_h;
};

// Fusion6 in ONE pass per image tile (round 3; rounds 1-2 ran k_ssim_stats + k_loss_grad with the three dS maps making a
// round trip through HBM).  A work-group owns a 16x64 tile of dL/dpred.  It needs the dS maps on the tile + a 10-pixel apron
// (adjoint of the valid 11x11 filter), hence p and t on the tile + a 20-pixel apron: the five filtered maps and the three dS
// maps are RECOMPUTED on the apron (2.9x / 1.9x the arithmetic of the tile alone: the separable filters are cheap VALU work)
// and never leave LDS.  HBM traffic: p, t read (aprons from L2), dL/dpred written.  Register-blocked separable filtering:
// every thread produces 4 adjacent outputs from a 14-wide window (3x fewer LDS reads than one output per thread).
//   LDS (80 KB, two work-groups per CU):  X [2][36][88] p, t patch | Hm [5][36][76] row-filtered p, t, p^2, t^2, pt
//                                         D [3][26][76] dS/dm, dS/dq, dS/dr (over X) | Hh [3][26][64] row-adjoint of D (over Hm)
#define F6_TH 16
#define F6_TW 64
#define F6_PH (F6_TH + 20)
#define F6_PWP 88            // 84 columns used
#define F6_DH (F6_TH + 10)
#define F6_DWP 76            // 74 columns used
#define F6_LDS_FLOATS (2 * F6_PH * F6_PWP + 5 * F6_PH * F6_DWP + 16)

enum { TC_NONE = 0, TC_READ = 1, TC_FILL = 2 };
template <bool GRAD, int TC>
__global__ void __launch_bounds__(256) k_fusion6(LossP q)
{
    extern __shared__ __attribute__((aligned(16))) float f6s[];
    float *Xp = f6s, *Xt = f6s + F6_PH * F6_PWP;
    float *Hm = f6s + 2 * F6_PH * F6_PWP;
    float *Dm = f6s;                       // written after the last read of Xp / Xt
    float *Hh = Hm;                        // written after the last read of Hm
    float *sred = f6s + 2 * F6_PH * F6_PWP + 5 * F6_PH * F6_DWP;
    const int t = threadIdx.x;
    const int plane = blockIdx.y;
    const int tw = blockIdx.x % q.tiles_w, th = blockIdx.x / q.tiles_w;
    const int y0 = th * F6_TH, x0 = tw * F6_TW;
    const size_t HW = (size_t)q.H * q.W;
    const size_t fr = TC == TC_FILL ? (size_t)blockIdx.z : (q.frame_idx ? (size_t)(*q.frame_idx) : 0);
    const float *tp = q.target + fr * q.frame_stride + (size_t)plane * HW;
    const float *pp = TC == TC_FILL ? tp : q.pred + (size_t)plane * HW;       // (TC_FILL: only the target's two maps are formed)
    float *ts_mu = nullptr, *ts_tt = nullptr;
    if (TC != TC_NONE) {
        ts_mu = q.tstats + fr * q.tstats_stride + (size_t)plane * q.Hv * q.Wv;
        ts_tt = ts_mu + (size_t)q.planes * q.Hv * q.Wv;
    }
    // the cached target statistics of this thread's nine map positions (column filter mapping below) are requested first: two
    // barriers lie between here and their use.  (Loaded where they are used, they cost more than the two filters they save:
    // with two work-groups per CU nothing covers a round trip to HBM in the middle of a phase.)
    constexpr int RPT = 9;
    const int vrg = t / F6_DWP, vj = t - vrg * F6_DWP, vr4 = vrg == 2 ? 17 : vrg * RPT;
    float c_mu[RPT], c_tt[RPT];
    if (TC == TC_READ && t < 3 * F6_DWP) {
#pragma unroll
        for (int o = 0; o < RPT; ++o) {
            const int vy = min(max(y0 - 10 + vr4 + o, 0), q.Hv - 1), vx = min(max(x0 - 10 + vj, 0), q.Wv - 1);
            c_mu[o] = ts_mu[(size_t)vy * q.Wv + vx];
            c_tt[o] = ts_tt[(size_t)vy * q.Wv + vx];
        }
    }
    // ---- patch rows y0-10 .. y0+25, columns x0-10 .. x0+73 (LDS column c <-> image column x0 - 10 + c); zero outside the image
    if (q.vec4) {           // rows are 16-byte aligned: float4 units from image column x0 - 12
        constexpr int NU = 22, NIT = (F6_PH * NU + 255) / 256;
        float4 ra[NIT], rb[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = t + it * 256;
            const int r = idx / NU, u = idx - r * NU;
            const int gy = y0 - 10 + r, gx = x0 - 12 + 4 * u;
            ra[it] = make_float4(0.f, 0.f, 0.f, 0.f); rb[it] = ra[it];
            if (idx < F6_PH * NU && gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) {
                ra[it] = *reinterpret_cast<const float4 *>(pp + (size_t)gy * q.W + gx);
                rb[it] = *reinterpret_cast<const float4 *>(tp + (size_t)gy * q.W + gx);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = t + it * 256;
            const int r = idx / NU, u = idx - r * NU;
            if (idx < F6_PH * NU) {
                const int c = 4 * u - 2;
                if (u > 0) {
                    *reinterpret_cast<float2 *>(Xp + r * F6_PWP + c) = make_float2(ra[it].x, ra[it].y);
                    *reinterpret_cast<float2 *>(Xt + r * F6_PWP + c) = make_float2(rb[it].x, rb[it].y);
                }
                *reinterpret_cast<float2 *>(Xp + r * F6_PWP + c + 2) = make_float2(ra[it].z, ra[it].w);
                *reinterpret_cast<float2 *>(Xt + r * F6_PWP + c + 2) = make_float2(rb[it].z, rb[it].w);
            }
        }
    } else {
        constexpr int NL = (F6_PH * F6_PWP + 255) / 256;
#pragma unroll 1
        for (int it = 0; it < NL; ++it) {
            const int idx = t + it * 256;
            const int r = idx / F6_PWP, c = idx - r * F6_PWP;
            const int gy = y0 - 10 + r, gx = x0 - 10 + c;
            float a = 0.f, b = 0.f;
            if (idx < F6_PH * F6_PWP && gy >= 0 && gy < q.H && gx >= 0 && gx < q.W) { a = pp[(size_t)gy * q.W + gx]; b = tp[(size_t)gy * q.W + gx]; }
            if (idx < F6_PH * F6_PWP) { Xp[idx] = a; Xt[idx] = b; }
        }
    }
    __syncthreads();
    // ---- row filter: Hm[m][r][j] = sum_k g[k] X[r][j + k], j < 76 (74 used); item = (row r, 4 columns)
    for (int idx = t; idx < F6_PH * (F6_DWP / 4); idx += 256) {
        const int r = idx / (F6_DWP / 4), c4 = (idx - r * (F6_DWP / 4)) * 4;
        float a[16], b[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 va = *reinterpret_cast<const float4 *>(Xp + r * F6_PWP + c4 + 4 * k);
            const float4 vb = *reinterpret_cast<const float4 *>(Xt + r * F6_PWP + c4 + 4 * k);
            a[4 * k] = va.x; a[4 * k + 1] = va.y; a[4 * k + 2] = va.z; a[4 * k + 3] = va.w;
            b[4 * k] = vb.x; b[4 * k + 1] = vb.y; b[4 * k + 2] = vb.z; b[4 * k + 3] = vb.w;
        }
        float sp[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0}, spp[4] = {0, 0, 0, 0}, stt[4] = {0, 0, 0, 0}, spt[4] = {0, 0, 0, 0};
        constexpr bool P_MAPS = TC != TC_FILL, T_MAPS = TC != TC_READ;       // which of the five maps this mode forms
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            const float aa = a[k] * a[k], bb = b[k] * b[k], ab = a[k] * b[k];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int tap = k - o;
                if (tap >= 0 && tap < 11) {
                    const float g = c_gauss[tap];
                    if (P_MAPS) { sp[o] = fmaf(g, a[k], sp[o]); spp[o] = fmaf(g, aa, spp[o]); spt[o] = fmaf(g, ab, spt[o]); }
                    if (T_MAPS) { st[o] = fmaf(g, b[k], st[o]); stt[o] = fmaf(g, bb, stt[o]); }
                }
            }
        }
        float *h = Hm + r * F6_DWP + c4;
        if (P_MAPS) {
            *reinterpret_cast<float4 *>(h) = make_float4(sp[0], sp[1], sp[2], sp[3]);
            *reinterpret_cast<float4 *>(h + 2 * F6_PH * F6_DWP) = make_float4(spp[0], spp[1], spp[2], spp[3]);
            *reinterpret_cast<float4 *>(h + 4 * F6_PH * F6_DWP) = make_float4(spt[0], spt[1], spt[2], spt[3]);
        }
        if (T_MAPS) {
            *reinterpret_cast<float4 *>(h + F6_PH * F6_DWP) = make_float4(st[0], st[1], st[2], st[3]);
            *reinterpret_cast<float4 *>(h + 3 * F6_PH * F6_DWP) = make_float4(stt[0], stt[1], stt[2], stt[3]);
        }
    }
    __syncthreads();
    // ---- column filter + SSIM map + dS maps on valid-map rows y0-10+i (i < 26), columns x0-10+j (j < 74).
    // item = (column j, 9 rows); the third row group restarts at row 17 (rows 17..25: row 17 is computed twice, same value)
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float ssum = 0.f;
    if (t < 3 * F6_DWP) {
        const int rg = vrg, j = vj, r4 = vr4;
        float v[5][RPT];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            if ((TC == TC_READ && (m == 1 || m == 3)) || (TC == TC_FILL && !(m == 1 || m == 3))) continue;
            float col[RPT + 10];
#pragma unroll
            for (int k = 0; k < RPT + 10; ++k) col[k] = Hm[(m * F6_PH + r4 + k) * F6_DWP + j];
#pragma unroll
            for (int o = 0; o < RPT; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc = fmaf(c_gauss[k], col[o + k], acc);
                v[m][o] = acc;
            }
        }
#pragma unroll
        for (int o = 0; o < RPT; ++o) {
            const int i = r4 + o;
            const int vy = y0 - 10 + i, vx = x0 - 10 + j;
            float dm = 0.f, dq = 0.f, dr = 0.f;
            if (TC == TC_FILL) {                        // the owned part of the two target maps, once
                if (vy >= 0 && vy < q.Hv && vx >= 0 && vx < q.Wv && i >= 10 && j >= 10 && j < F6_TW + 10 && (rg < 2 || o > 0)) {
                    ts_mu[(size_t)vy * q.Wv + vx] = v[1][o];
                    ts_tt[(size_t)vy * q.Wv + vx] = v[3][o];
                }
                continue;
            }
            if (vy >= 0 && vy < q.Hv && vx >= 0 && vx < q.Wv && j < F6_TW + 10) {
                const float m = v[0][o], qq = v[2][o], rr = v[4][o];
                const float mu = TC == TC_READ ? c_mu[o] : v[1][o];
                const float tt = TC == TC_READ ? c_tt[o] : v[3][o];
                const float sp = qq - m * m, st = tt - mu * mu, spt = rr - m * mu;
                const float A1 = 2.f * m * mu + C1, A2 = 2.f * spt + C2;
                const float B1 = m * m + mu * mu + C1, B2 = sp + st + C2;
                // two reciprocals (v_rcp_f32 + one Newton step: < 1 ulp) instead of four IEEE divisions (~10 instructions each)
                float i1 = __builtin_amdgcn_rcpf(B1), i2 = __builtin_amdgcn_rcpf(B2);
                i1 = i1 * (2.0f - B1 * i1);
                i2 = i2 * (2.0f - B2 * i2);
                const float inv = i1 * i2;
                const float S = A1 * A2 * inv;
                if (i >= 10 && j >= 10 && (rg < 2 || o > 0)) ssum += S;      // this tile's own part of the map, once
                dm = 2.f * mu * (A2 - A1) * inv - 2.f * m * S * i1 + 2.f * m * S * i2;
                dq = -S * i2;
                dr = 2.f * A1 * inv;
            }
            float *d = Dm + i * F6_DWP + j;
            d[0] = dm; d[F6_DH * F6_DWP] = dq; d[2 * F6_DH * F6_DWP] = dr;      // X is dead since the barrier above
        }
    }
    if (TC == TC_FILL) return;
    // this thread's four output pixels (last phase) are requested before the two adjoint passes
    const int fc = t & (F6_TW - 1), fr4 = (t >> 6) * 4;
    float fp[4], ft[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const size_t oo = (size_t)min(y0 + fr4 + o, q.H - 1) * q.W + min(x0 + fc, q.W - 1);
        fp[o] = pp[oo]; ft[o] = tp[oo];
    }
    __syncthreads();
    float sabs = 0.f, ssq = 0.f;
    // ---- adjoint row filter: Hh[m][i][c] = sum_k g[k] D[m][i][c + 10 - k]; item = (row i, 4 columns, map m)
    if (GRAD) {
        for (int idx = t; idx < F6_DH * (F6_TW / 4); idx += 256) {
            const int r = idx / (F6_TW / 4), c4 = (idx - r * (F6_TW / 4)) * 4;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float w[16];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 vv = *reinterpret_cast<const float4 *>(Dm + (m * F6_DH + r) * F6_DWP + c4 + 4 * k);
                    w[4 * k] = vv.x; w[4 * k + 1] = vv.y; w[4 * k + 2] = vv.z; w[4 * k + 3] = vv.w;
                }
                float hv[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(c_gauss[k], w[o + 10 - k], acc);
                    hv[o] = acc;
                }
                *reinterpret_cast<float4 *>(Hh + (m * F6_DH + r) * F6_TW + c4) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            }
        }
        __syncthreads();
    }
    // ---- adjoint column filter + L1 term: thread = (column c, 4 rows)
    {
        const int c = fc, r4 = fr4;
        float am[4] = {0, 0, 0, 0}, aq[4] = {0, 0, 0, 0}, ar[4] = {0, 0, 0, 0};
        if (GRAD) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float col[14];
#pragma unroll
                for (int k = 0; k < 14; ++k) col[k] = Hh[(m * F6_DH + r4 + k) * F6_TW + c];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(c_gauss[k], col[o + 10 - k], acc);
                    if (m == 0) am[o] = acc; else if (m == 1) aq[o] = acc; else ar[o] = acc;
                }
            }
        }
        const int gx = x0 + c;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int gy = y0 + r4 + o;
            if (gy >= q.H || gx >= q.W) continue;
            const size_t oo = (size_t)gy * q.W + gx;
            const float p = fp[o], tg = ft[o], d = p - tg;
            sabs += fabsf(d);
            ssq = fmaf(d, d, ssq);
            if (GRAD) {
                float g = q.g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
                g -= q.g_ssim * (am[o] + 2.f * p * aq[o] + tg * ar[o]);
                q.dpred[(size_t)plane * HW + oo] = g;
            }
        }
    }
    const float ts = orn_block_sum(ssum, sred);
    const float ta = orn_block_sum(sabs, sred);
    const float tq = orn_block_sum(ssq, sred);
    if (t == 0) {
        const size_t bi = (size_t)plane * gridDim.x + blockIdx.x;
        q.part_ssim[bi] = ts;
        q.part_l1[2 * bi] = ta;
        q.part_l1[2 * bi + 1] = tq;
    }
}

// L1 / L2 losses (no SSIM term).  LT / GRAD are compile-time: the per-pixel loop carries no loss-type or null-pointer branches
template <int LT, bool GRAD>
__global__ void __launch_bounds__(256) k_loss_grad(LossP q)
{
    __shared__ float sred[16];
    const int t = threadIdx.x;
    const int plane = blockIdx.y;
    const int tw = blockIdx.x % q.tiles_w, th = blockIdx.x / q.tiles_w;
    const int y0 = th * SS_TH, x0 = tw * SS_TW;
    const size_t HW = (size_t)q.H * q.W;
    const float *pp = q.pred + (size_t)plane * HW;
    const float *tp = q.target + (q.frame_idx ? (size_t)(*q.frame_idx) * q.frame_stride : 0) + (size_t)plane * HW;
    float sabs = 0.f, ssq = 0.f;
    for (int idx = t; idx < SS_TH * SS_TW; idx += 256) {
        const int r = idx / SS_TW, c = idx - r * SS_TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= q.H || gx >= q.W) continue;
        const size_t o = (size_t)gy * q.W + gx;
        const float p = pp[o], tg = tp[o], d = p - tg;
        sabs += fabsf(d);
        ssq = fmaf(d, d, ssq);
        if (GRAD) {
            float g;
            if (LT == ORN_LOSS_L2) g = q.g_l2 * d;
            else g = q.g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
            q.dpred[(size_t)plane * HW + o] = g;
        }
    }
    const float ta = orn_block_sum(sabs, sred);
    const float tq = orn_block_sum(ssq, sred);
    if (t == 0) {
        const size_t bi = (size_t)plane * gridDim.x + blockIdx.x;
        q.part_l1[2 * bi] = ta;
        q.part_l1[2 * bi + 1] = tq;
    }
}

// One block; fixed-order strided partial sums in double, then a fixed tree (orn_common.h: orn_loss_finalize_block).
__global__ void __launch_bounds__(256) k_loss_finalize(OrnLossFinalJob j)
{
    __shared__ double sd[3 * 256];
    orn_loss_finalize_block(j, sd);
}

static bool g_gauss_ready = false;

static int ensure_gauss()
{
    if (g_gauss_ready) return 0;
    // pytorch_msssim _fspecial_gauss_1d(11, 1.5): fp32 exp, fp32 normalise
    float g[11], s = 0.f;
    for (int i = 0; i < 11; ++i) {
        const float c = (float)(i - 5);
        g[i] = expf(-(c * c) / (2.0f * 1.5f * 1.5f));
        s += g[i];
    }
    for (int i = 0; i < 11; ++i) g[i] /= s;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g));
    if (e != hipSuccess) {
        orn_set_error("loss: hipMemcpyToSymbol failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    g_gauss_ready = true;
    return 0;
}

struct LossGeom { int planes, Hv, Wv, tw, th; size_t nmap, off_pl, total; };

// one tiling (16 x 64 pixels of the image) for every loss type; per-tile partial sums: SSIM [n], {|d|, d^2} [n][2]
static LossGeom loss_geom(int B, int Ch, int H, int W)
{
    LossGeom g;
    g.planes = B * Ch;
    g.Hv = H > 10 ? H - 10 : 0;
    g.Wv = W > 10 ? W - 10 : 0;
    g.tw = orn_cdiv(W, SS_TW); g.th = orn_cdiv(H, SS_TH);
    g.nmap = (size_t)g.planes * g.Hv * g.Wv;
    g.off_pl = orn_align((size_t)g.planes * g.tw * g.th * 4 + 4) / 4;
    g.total = g.off_pl + orn_align((size_t)g.planes * g.tw * g.th * 8 + 8) / 4;
    return g;
}

extern "C" size_t orn_loss_ws_bytes(int B, int Ch, int H, int W) { return loss_geom(B, Ch, H, W).total * 4; }

static int ensure_fusion6_lds()
{
    static bool done = false;
    if (done) return 0;
    hipError_t e = hipSuccess;
    const void *kerns[] = {(const void *)k_fusion6<true, TC_NONE>, (const void *)k_fusion6<false, TC_NONE>, (const void *)k_fusion6<true, TC_READ>,
                           (const void *)k_fusion6<false, TC_READ>, (const void *)k_fusion6<false, TC_FILL>};
    for (const void *k : kerns)
        if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, F6_LDS_FLOATS * 4);
    if (e != hipSuccess) { orn_set_error("loss: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    done = true;
    return 0;
}

// Must be called once outside any graph capture (hipMemcpyToSymbol is synchronous).
int orn_loss_init() { ORN_TRY(ensure_gauss()); return ensure_fusion6_lds(); }

int orn_launch_loss(const float *pred, const float *target, const int *frame_idx, size_t frame_stride, int B, int Ch,
                    int H, int W, int loss_type, float loss_scale, float *stats, float *dpred, float *ws,
                    hipStream_t st, const OrnStepCur *cur, float *ring, OrnScaleState *sc, const float *tstats, OrnLossFinalJob *defer)
{
    ORN_REQUIRE(loss_type == ORN_LOSS_L2 || loss_type == ORN_LOSS_L1 || loss_type == ORN_LOSS_FUSION6,
                "loss: unsupported loss_type %d", loss_type);
    ORN_TRY(orn_loss_init());
    static_assert(SS_TH == F6_TH && SS_TW == F6_TW, "one tiling for all loss kernels");
    const LossGeom g = loss_geom(B, Ch, H, W);
    if (loss_type == ORN_LOSS_FUSION6) ORN_REQUIRE(g.Hv > 0 && g.Wv > 0, "loss: SSIM needs H,W > 10 (got %dx%d)", H, W);
    LossP q;
    q.pred = pred; q.target = target; q.frame_idx = frame_idx; q.frame_stride = frame_stride;
    q.planes = g.planes; q.H = H; q.W = W; q.Hv = g.Hv; q.Wv = g.Wv;
    q.part_ssim = ws; q.part_l1 = ws + g.off_pl;
    q.dpred = dpred; q.loss_type = loss_type;
    q.tstats = const_cast<float *>(tstats); q.tstats_stride = 2 * g.nmap;
    q.vec4 = (W % 4 == 0 && ((uintptr_t)pred | (uintptr_t)target) % 16 == 0 && (frame_stride % 4 == 0 || !frame_idx)) ? 1 : 0;
    const double n = (double)g.planes * H * W;
    q.g_l1 = (float)((loss_type == ORN_LOSS_FUSION6 ? 0.7 : 1.0) * loss_scale / n);
    q.g_l2 = (float)(2.0 * loss_scale / n);
    q.g_ssim = g.nmap ? (float)(0.3 * loss_scale / (double)g.nmap) : 0.f;
    q.tiles_w = g.tw; q.tiles_h = g.th;
    const int n_tiles = g.planes * g.tw * g.th;
    {
        const dim3 gr(g.tw * g.th, g.planes), bl(256);
        const bool gd = q.dpred != nullptr;
        if (loss_type == ORN_LOSS_FUSION6) {
            if (tstats) { if (gd) hipLaunchKernelGGL((k_fusion6<true, TC_READ>), gr, bl, F6_LDS_FLOATS * 4, st, q); else hipLaunchKernelGGL((k_fusion6<false, TC_READ>), gr, bl, F6_LDS_FLOATS * 4, st, q); }
            else if (gd) hipLaunchKernelGGL((k_fusion6<true, TC_NONE>), gr, bl, F6_LDS_FLOATS * 4, st, q); else hipLaunchKernelGGL((k_fusion6<false, TC_NONE>), gr, bl, F6_LDS_FLOATS * 4, st, q);
        }
        else if (loss_type == ORN_LOSS_L2) { if (gd) hipLaunchKernelGGL((k_loss_grad<ORN_LOSS_L2, true>), gr, bl, 0, st, q); else hipLaunchKernelGGL((k_loss_grad<ORN_LOSS_L2, false>), gr, bl, 0, st, q); }
        else { if (gd) hipLaunchKernelGGL((k_loss_grad<ORN_LOSS_L1, true>), gr, bl, 0, st, q); else hipLaunchKernelGGL((k_loss_grad<ORN_LOSS_L1, false>), gr, bl, 0, st, q); }
    }
    ORN_LAUNCH_CHECK("loss");
    const OrnLossFinalJob fj = {q.part_ssim, loss_type == ORN_LOSS_FUSION6 ? n_tiles : 0, q.part_l1, n_tiles, n, (double)g.nmap, loss_type,
                                loss_scale, stats, cur, ring, sc};
    if (defer) { *defer = fj; return 0; }              // the caller runs it as a rider of a later launch
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, st, fj);
    ORN_LAUNCH_CHECK("loss_finalize");
    return 0;
}

// The target side of Fusion6's SSIM statistics for `n` frames (LossP::tstats): out [n][2][Ch][H-10][W-10].
extern "C" size_t orn_loss_target_stats_bytes(int n, int Ch, int H, int W)
{
    if (n <= 0 || Ch <= 0 || H <= 10 || W <= 10) return 0;
    return (size_t)n * 2 * Ch * (H - 10) * (W - 10) * sizeof(float);
}

extern "C" int orn_loss_target_stats(const float *frames, int n, int Ch, int H, int W, float *out, void *stream)
{
    ORN_REQUIRE(frames && out && n > 0 && Ch > 0 && H > 10 && W > 10 && n <= 65535, "loss_target_stats: bad arguments");
    ORN_TRY(orn_loss_init());
    const LossGeom g = loss_geom(1, Ch, H, W);
    LossP q = {};
    q.pred = frames; q.target = frames; q.frame_idx = nullptr; q.frame_stride = (size_t)Ch * H * W;
    q.planes = g.planes; q.H = H; q.W = W; q.Hv = g.Hv; q.Wv = g.Wv;
    q.tstats = out; q.tstats_stride = 2 * g.nmap;
    q.loss_type = ORN_LOSS_FUSION6;
    q.vec4 = (W % 4 == 0 && (uintptr_t)frames % 16 == 0) ? 1 : 0;
    q.tiles_w = g.tw; q.tiles_h = g.th;
    hipLaunchKernelGGL((k_fusion6<false, TC_FILL>), dim3(g.tw * g.th, g.planes, n), dim3(256), F6_LDS_FLOATS * 4, (hipStream_t)stream, q);
    ORN_LAUNCH_CHECK("loss_target_stats");
    return 0;
}

extern "C" int orn_loss_fwd_bwd(const float *pred, const float *target, int B, int Ch, int H, int W, int loss_type,
                                float loss_scale, float *stats, float *dpred, void *ws, size_t ws_bytes,
                                void *stream)
{
    ORN_REQUIRE(pred && target && stats && ws, "loss_fwd_bwd: null pointer");
    ORN_REQUIRE(B > 0 && Ch > 0 && H > 0 && W > 0, "loss_fwd_bwd: bad sizes");
    if (ws_bytes < orn_loss_ws_bytes(B, Ch, H, W)) {
        orn_set_error("loss_fwd_bwd: workspace %zu < %zu", ws_bytes, orn_loss_ws_bytes(B, Ch, H, W));
        return ORN_E_WS;
    }
    return orn_launch_loss(pred, target, nullptr, 0, B, Ch, H, W, loss_type, loss_scale, stats, dpred, (float *)ws,
                           (hipStream_t)stream, nullptr, nullptr);
}

// ================================================================================================
// N3  msssim_fn (utils.py:201-211): pytorch_msssim.ms_ssim, 5 scales, weights (0.0448, 0.2856, 0.3001, 0.2363,
// 0.1333), avg_pool2d(2) between scales (parity unpinned like SSIM: the package is not in the reference tree).
// Logging metric only -- kept off the timed training path.
// ================================================================================================
// per-block partial sums of ssim_map and cs_map over a 16x64 tile of one plane
__global__ void __launch_bounds__(256)
k_ssim_cs_partial(const float *__restrict__ x, const float *__restrict__ y, int H, int W, int tiles_w, float *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float Ps[SS_PH][SS_PWP];
    __shared__ __attribute__((aligned(16))) float Ts[SS_PH][SS_PWP];
    __shared__ float Hs[5][SS_PH][SS_TW];
    __shared__ float sred[16];
    const int t = threadIdx.x, plane = blockIdx.y;
    const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
    const int y0 = th * SS_TH, x0 = tw * SS_TW;
    const int Hv = H - 10, Wv = W - 10;
    const float *pp = x + (size_t)plane * H * W, *tp = y + (size_t)plane * H * W;
    for (int idx = t; idx < SS_PH * SS_PWP; idx += 256) {
        const int r = idx / SS_PWP, c = idx - r * SS_PWP;
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = c < SS_PW && gy < H && gx < W;
        Ps[r][c] = ok ? pp[(size_t)gy * W + gx] : 0.f;
        Ts[r][c] = ok ? tp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int idx = t; idx < SS_PH * SS_TW; idx += 256) {
        const int r = idx / SS_TW, c = idx - r * SS_TW;
        float sp = 0.f, st = 0.f, spp = 0.f, stt = 0.f, spt = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = c_gauss[k], a = Ps[r][c + k], b = Ts[r][c + k];
            sp = fmaf(g, a, sp); st = fmaf(g, b, st);
            spp = fmaf(g, a * a, spp); stt = fmaf(g, b * b, stt); spt = fmaf(g, a * b, spt);
        }
        Hs[0][r][c] = sp; Hs[1][r][c] = st; Hs[2][r][c] = spp; Hs[3][r][c] = stt; Hs[4][r][c] = spt;
    }
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float ss = 0.f, cs = 0.f;
    for (int idx = t; idx < SS_TH * SS_TW; idx += 256) {
        const int r = idx / SS_TW, c = idx - r * SS_TW;
        if (y0 + r >= Hv || x0 + c >= Wv) continue;
        float m = 0.f, mu = 0.f, qq = 0.f, tt = 0.f, rr = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = c_gauss[k];
            m = fmaf(g, Hs[0][r + k][c], m); mu = fmaf(g, Hs[1][r + k][c], mu);
            qq = fmaf(g, Hs[2][r + k][c], qq); tt = fmaf(g, Hs[3][r + k][c], tt); rr = fmaf(g, Hs[4][r + k][c], rr);
        }
        const float csm = (2.f * (rr - m * mu) + C2) / ((qq - m * m) + (tt - mu * mu) + C2);
        cs += csm;
        ss += ((2.f * m * mu + C1) / (m * m + mu * mu + C1)) * csm;
    }
    const float a = orn_block_sum(ss, sred);
    const float b = orn_block_sum(cs, sred);
    if (t == 0) {
        const size_t bi = (size_t)plane * gridDim.x + blockIdx.x;
        part[2 * bi] = a;
        part[2 * bi + 1] = b;
    }
}

// avg_pool2d(kernel 2, padding = size % 2, count_include_pad = True) of every plane
__global__ void k_avgpool2(const float *__restrict__ in, int planes, int H, int W, int Ho, int Wo, int ph, int pw, float *__restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)planes * Ho * Wo) return;
    const int xo = (int)(idx % Wo), yo = (int)((idx / Wo) % Ho), pl = (int)(idx / ((size_t)Wo * Ho));
    const float *p = in + (size_t)pl * H * W;
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int yy = yo * 2 + dy - ph, xx = xo * 2 + dx - pw;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += p[(size_t)yy * W + xx];
        }
    out[idx] = s * 0.25f;
}

// level means -> relu -> weighted product per plane -> mean over planes
__global__ void k_msssim_finalize(const float *__restrict__ part, const int *__restrict__ nblk, const float *__restrict__ nmap,
                                  int planes, float *__restrict__ out)
{
    __shared__ double acc[64];
    const float wts[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    const int pl = threadIdx.x;
    double val = 0.0;
    if (pl < planes) {
        size_t off = 0;
        double prod = 1.0;
        for (int lv = 0; lv < 5; ++lv) {
            double ss = 0.0, cs = 0.0;
            for (int b = 0; b < nblk[lv]; ++b) {
                ss += (double)part[off + 2 * ((size_t)pl * nblk[lv] + b)];
                cs += (double)part[off + 2 * ((size_t)pl * nblk[lv] + b) + 1];
            }
            off += (size_t)2 * planes * nblk[lv];
            const double v = (lv < 4 ? cs : ss) / (double)nmap[lv];
            prod *= pow(v > 0.0 ? v : 0.0, (double)wts[lv]);
        }
        val = prod;
    }
    acc[pl] = val;
    __syncthreads();
    if (pl == 0) {
        double s = 0.0;
        for (int i = 0; i < planes; ++i) s += acc[i];
        out[0] = (float)(s / planes);
    }
}

static void msssim_geom(int H, int W, int Hs[5], int Ws[5])
{
    Hs[0] = H; Ws[0] = W;
    for (int l = 1; l < 5; ++l) { Hs[l] = (Hs[l - 1] + 2 * (Hs[l - 1] % 2)) / 2; Ws[l] = (Ws[l - 1] + 2 * (Ws[l - 1] % 2)) / 2; }
}

extern "C" size_t orn_msssim_ws_bytes(int B, int Ch, int H, int W)
{
    int Hs[5], Ws[5];
    msssim_geom(H, W, Hs, Ws);
    const size_t planes = (size_t)B * Ch;
    size_t f = 64;                                       // level tables
    for (int l = 1; l < 5; ++l) f += 2 * orn_align(planes * Hs[l] * Ws[l] * 4) / 4;
    for (int l = 0; l < 5; ++l) f += orn_align(2 * planes * orn_cdiv(Ws[l] - 10, SS_TW) * orn_cdiv(Hs[l] - 10, SS_TH) * 4 + 8) / 4;
    return f * 4;
}

// out[0] = ms_ssim(pred, target, data_range=1, size_average=True).  Needs min(H, W) > 160 (pytorch_msssim's own limit).
extern "C" int orn_msssim(const float *pred, const float *target, int B, int Ch, int H, int W, float *out, void *ws,
                          size_t ws_bytes, void *stream)
{
    ORN_REQUIRE(pred && target && out && ws, "msssim: null pointer");
    ORN_REQUIRE(B > 0 && Ch > 0 && B * Ch <= 64, "msssim: bad plane count");
    ORN_REQUIRE((H < W ? H : W) > 160, "msssim: image side must exceed 160 (got %dx%d)", H, W);
    if (ws_bytes < orn_msssim_ws_bytes(B, Ch, H, W)) { orn_set_error("msssim: workspace too small"); return ORN_E_WS; }
    ORN_TRY(ensure_gauss());
    hipStream_t st = (hipStream_t)stream;
    int Hs[5], Ws[5];
    msssim_geom(H, W, Hs, Ws);
    const int planes = B * Ch;
    float *base = (float *)ws;
    int *d_nblk = (int *)base;
    float *d_nmap = base + 8;
    float *cur = base + 64;
    const float *x = pred, *y = target;
    int h_nblk[5];
    float h_nmap[5];
    float *pooled[5][2] = {};
    for (int l = 1; l < 5; ++l)
        for (int k = 0; k < 2; ++k) { pooled[l][k] = cur; cur += orn_align((size_t)planes * Hs[l] * Ws[l] * 4) / 4; }
    float *part = cur;
    size_t poff = 0;
    for (int l = 0; l < 5; ++l) {
        const int tw = orn_cdiv(Ws[l] - 10, SS_TW), th = orn_cdiv(Hs[l] - 10, SS_TH);
        h_nblk[l] = tw * th;
        h_nmap[l] = (float)(Hs[l] - 10) * (float)(Ws[l] - 10);
        hipLaunchKernelGGL(k_ssim_cs_partial, dim3(tw * th, planes), dim3(256), 0, st, x, y, Hs[l], Ws[l], tw, part + poff);
        ORN_LAUNCH_CHECK("ssim_cs_partial");
        poff += (size_t)2 * planes * h_nblk[l];
        if (l < 4) {
            const size_t n = (size_t)planes * Hs[l + 1] * Ws[l + 1];
            hipLaunchKernelGGL(k_avgpool2, dim3(orn_cdiv((long)n, 256)), dim3(256), 0, st, x, planes, Hs[l], Ws[l], Hs[l + 1], Ws[l + 1],
                               Hs[l] % 2, Ws[l] % 2, pooled[l + 1][0]);
            hipLaunchKernelGGL(k_avgpool2, dim3(orn_cdiv((long)n, 256)), dim3(256), 0, st, y, planes, Hs[l], Ws[l], Hs[l + 1], Ws[l + 1],
                               Hs[l] % 2, Ws[l] % 2, pooled[l + 1][1]);
            ORN_LAUNCH_CHECK("avgpool2");
            x = pooled[l + 1][0];
            y = pooled[l + 1][1];
        }
    }
    hipError_t rc = hipMemcpyAsync(d_nblk, h_nblk, sizeof(h_nblk), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess) rc = hipMemcpyAsync(d_nmap, h_nmap, sizeof(h_nmap), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);        // tables live on the host stack (logging path, not graph-captured)
    if (rc != hipSuccess) { orn_set_error("msssim: table upload failed: %s", hipGetErrorString(rc)); return (int)rc; }
    hipLaunchKernelGGL(k_msssim_finalize, dim3(1), dim3(64), 0, st, part, d_nblk, d_nmap, planes, out);
    ORN_LAUNCH_CHECK("msssim_finalize");
    return 0;
}
