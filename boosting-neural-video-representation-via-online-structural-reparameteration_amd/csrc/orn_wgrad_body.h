// wgrad work-group body of the 16-bit fast path (dW[tap][o'][c] = sum_p dy[p][o'] * x[p + off(tap)][c]), shared by
// orn_conv_bf16.hip (its own launches) and orn_conv2_bf16.hip (wgrad work-groups riding behind a block's dgrad tiles).
// Include INSIDE namespace HNS, after h16 / h16x8 / s16x4 / f32x16 / MFMA_H16 (v_mfma_f32_32x32x16) / PDBG are defined.
#pragma once
#define WB_TH 2                  // K tile = 2 rows x 32 pixels
#define WB_TW 32
#define WB_NPX (WB_TH * WB_TW)
#define WB_BO 128
#define WB_DYB 256               // LDS bytes per dy pixel row: unpadded, 16-byte chunks XOR-swizzled by (pixel & 3) << 2
#define WB_XB 192                // LDS bytes per x pixel (96 ch, unpadded: conflict-free tr reads as is)
#define WB_XW (WB_TW + 2)
#define WB_DY_BYTES (WB_NPX * WB_DYB)                        /* 16 KiB = 16 DMA wave-instructions */
#define WB_X_INSTR ((WB_TH * WB_XW * 12 + 63) / 64)          /* 13 DMA wave-instructions          */
#define WB_X_BYTES (WB_X_INSTR * 1024)
#define WB_BUF_BYTES (WB_DY_BYTES + WB_X_BYTES)

struct WgradBP {
    const h16 *xpad;    // [H+2][W+2][96]
    const h16 *dypad;   // [H+2][W+2][O]
    float *slabs;       // [S][9][O][96]
    float *bias_slabs;  // [S][O]  (column sums of dy, from the ti == 1 work-groups)
    int H, W, O;
    int tiles_w, n_ktiles, S, n_otiles;
    int dbg;            // timing-only ablation flags (tools/probes)
};

__device__ __forceinline__ h16x8 tr_frag(const unsigned char *base0, const unsigned char *base1)
{
    // two transposed 4x16 block reads -> the 8 K-consecutive elements of this lane's row/column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(h16x8, v);
}

// Both LDS images are filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 2-deep ring:
// tile t+1 streams in while tile t feeds the matrix core; one barrier per tile; two work-groups per CU.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // the LDS-DMA asm names m0 as clobbered (nothing else in this function uses it)
__device__ __forceinline__ void wgrad_body(const WgradBP &p, const int id)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, hh = lane >> 5;
    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    // XCD-aware decode: the 9 work-groups that share one pixel range sit on one XCD (speed only)
    const int xcd = id & 7, qx = id >> 3;
    const int sub = qx % (3 * p.n_otiles), sidx = (qx / (3 * p.n_otiles)) * 8 + xcd;
    const int ti = sub % 3, ot = sub / 3;
    const int o0 = ot * WB_BO;
    const int H = p.H, W = p.W, O = p.O;

    f32x16 acc[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][c][r] = 0.f;
    // dbias rides along: the A fragment holds 8 pixels of one dy channel per lane, so its column sum is four packed
    // dot products with ones (v_dot2c_f32, fp32 accumulate) that issue under the MFMAs.  (A ones-MFMA in the ti == 1
    // work-groups made those 11 % longer than their neighbours: -17 us on the last block's launch.)  Every work-group
    // computes it -- no branch in the K loop -- and the ti == 1 ones store it.
    float bsum = 0.f;
    typedef __attribute__((ext_vector_type(2))) h16 h16v2;
    const h16v2 ones2 = {(h16)1.0f, (h16)1.0f};
    const bool do_bias = (ti == 1);

    // per-lane transposed-read offsets (pixel part is added per K slice).  dy: this lane's pixels all have
    // (pixel & 3) == lq, so the XOR swizzle is a per-lane constant.
    const int a_chunk = uwave * 4 + 2 * (g & 1) + (lp >> 1);
    const int a_off = (8 * (g >> 1) + lq) * WB_DYB + ((a_chunk ^ (lq << 2)) * 16) + (lp & 1) * 8;
    const int b_off = (8 * (g >> 1) + lq) * WB_XB + (16 * (g & 1) + 4 * lp) * 2;

    // DMA plan of this wave: 4 dy instructions (64 pixels x 16 chunks / 4 waves) + up to 4 x instructions
    constexpr int DY_PW = (WB_DY_BYTES / 1024) / 4;          // 4
    constexpr int X_PW = (WB_X_INSTR + 3) / 4;               // 4 (13 instructions over 4 waves)
    int dy_px[DY_PW], dy_c[DY_PW], x_px[X_PW], x_c[X_PW];
#pragma unroll
    for (int k = 0; k < DY_PW; ++k) {
        const int L = (uwave + 4 * k) * 64 + lane;           // linear 16-byte slot
        dy_px[k] = L >> 4;
        dy_c[k] = (L & 15) ^ ((dy_px[k] & 3) << 2);          // logical chunk stored at this slot
    }
#pragma unroll
    for (int k = 0; k < X_PW; ++k) {
        const int L = (uwave + 4 * k) * 64 + lane;
        x_px[k] = L / 12;
        x_c[k] = L - x_px[k] * 12;
    }
    // The LDS-DMA is issued through inline asm: behind the BUILTIN the compiler (which sees an LDS store it cannot tell apart
    // from the buffer being read) puts `s_waitcnt vmcnt(0)` in front of the first fragment read of the CURRENT tile -- the wave
    // then waits for the tile it has just requested before it starts the one it holds, and the 2-deep ring prefetches nothing.
    // The one wait this ring needs is the explicit one at the top of the loop.
    const unsigned wlds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
#define WDMA16(gptr_, ldsoff_)                                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"   /* (one wait state between the M0 write and its use) */ \
                 :: "s"(wlds0 + (unsigned)(ldsoff_)), "v"((const void *)(gptr_)) : "memory", "m0")
    // same, source = wave-uniform base (SGPR pair) + per-lane 32-bit byte offset
#define WDMA16S(sbase_, voff_, ldsoff_)                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                               \
                 :: "s"(wlds0 + (unsigned)(ldsoff_)), "v"(voff_), "s"((const void *)(sbase_)) : "memory", "m0")
    // Interior tiles (every pixel of the tile inside the image: all but a ragged last column / row of tiles) take their
    // addresses from per-lane offsets computed ONCE relative to the tile origin; the general form below redoes the index
    // arithmetic and the bounds tests per piece -- ~25 vector instructions x 8 pieces per tile against 36 MFMAs.
    unsigned dy_loff[DY_PW], x_loff[X_PW];
#pragma unroll
    for (int k = 0; k < DY_PW; ++k)
        dy_loff[k] = (unsigned)(((dy_px[k] / WB_TW + 1) * (W + 2) + (dy_px[k] & (WB_TW - 1)) + 1) * O + o0 + dy_c[k] * 8) * 2u;
#pragma unroll
    for (int k = 0; k < X_PW; ++k) {
        const int r = x_px[k] / WB_XW, c = x_px[k] - r * WB_XW;
        x_loff[k] = x_px[k] < WB_TH * WB_XW ? (unsigned)(((r + ti) * (W + 2) + c) * 96 + x_c[k] * 8) * 2u : 0u;   // slots behind the patch: never read
    }
#define WDMA_TILE(kt_, buf_)                                                                                    \
    {                                                                                                           \
        const int th_ = (kt_) / p.tiles_w, tw_ = (kt_) - th_ * p.tiles_w;                                       \
        const int h0_ = th_ * WB_TH, w0_ = tw_ * WB_TW;                                                         \
        if (h0_ + WB_TH <= H && w0_ + WB_TW <= W && !(PDBG(p) & 1)) {                                           \
            const h16 *dyb_ = p.dypad + ((size_t)h0_ * (W + 2) + w0_) * O;                                      \
            const h16 *xb_ = p.xpad + ((size_t)h0_ * (W + 2) + w0_) * 96;                                       \
            _Pragma("unroll") for (int k = 0; k < DY_PW; ++k)                                                   \
                WDMA16S(dyb_, dy_loff[k], (buf_) * WB_BUF_BYTES + (uwave + 4 * k) * 1024);                      \
            _Pragma("unroll") for (int k = 0; k < X_PW; ++k)                                                    \
                if (uwave + 4 * k < WB_X_INSTR)                                                                 \
                    WDMA16S(xb_, x_loff[k], (buf_) * WB_BUF_BYTES + WB_DY_BYTES + (uwave + 4 * k) * 1024);      \
        } else {                                                                                                \
        _Pragma("unroll") for (int k = 0; k < DY_PW; ++k) {                                                     \
            const int gh = h0_ + dy_px[k] / WB_TW, gw = w0_ + (dy_px[k] & (WB_TW - 1));                         \
            const bool ok = gh < H && gw < W && !(PDBG(p) & 1);                                                   \
            const h16 *src = ok ? p.dypad + ((size_t)(gh + 1) * (W + 2) + (gw + 1)) * O + o0 + dy_c[k] * 8      \
                                : p.dypad + dy_c[k] * 8; /* border pixel (0,0): zeros */                        \
            WDMA16(src, (buf_) * WB_BUF_BYTES + (uwave + 4 * k) * 1024);                                        \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < X_PW; ++k) {                                                      \
            if (uwave + 4 * k < WB_X_INSTR) {                                                                   \
                const int r = x_px[k] / WB_XW, c = x_px[k] - r * WB_XW;                                         \
                const int gh = h0_ + r + ti, gw = w0_ + c;                                                      \
                const bool ok = x_px[k] < WB_TH * WB_XW && gh < H + 2 && gw < W + 2 && !(PDBG(p) & 1);            \
                const h16 *src = ok ? p.xpad + ((size_t)gh * (W + 2) + gw) * 96 + x_c[k] * 8 : p.xpad + x_c[k] * 8; \
                WDMA16(src, (buf_) * WB_BUF_BYTES + WB_DY_BYTES + (uwave + 4 * k) * 1024);                      \
            }                                                                                                   \
        }                                                                                                       \
        }                                                                                                       \
    }

    int buf = 0;
    if (sidx < p.n_ktiles) WDMA_TILE(sidx, 0)
    for (int kt = sidx; kt < p.n_ktiles; kt += p.S) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                         // ... everyone's have; everyone is done with tile kt - S
        if (kt + p.S < p.n_ktiles) WDMA_TILE(kt + p.S, buf ^ 1)
        const unsigned char *dys = smem + buf * WB_BUF_BYTES;
        const unsigned char *xs = dys + WB_DY_BYTES;
#pragma unroll
        for (int r = 0; r < WB_TH; ++r)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned char *ap = dys + (r * WB_TW + 16 * half) * WB_DYB + a_off;
                const h16x8 a = tr_frag(ap, ap + 4 * WB_DYB);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16v2 a2 = {a[2 * e], a[2 * e + 1]};
#ifdef ORN_FP16
                    bsum = __builtin_amdgcn_fdot2(a2, ones2, bsum, false);
#else
                    bsum = __builtin_amdgcn_fdot2_f32_bf16(a2, ones2, bsum, false);
#endif
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const unsigned char *bp = xs + (r * WB_XW + 16 * half + j) * WB_XB + b_off;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const h16x8 b = tr_frag(bp + c * 64, bp + c * 64 + 4 * WB_XB);
                        acc[j][c] = MFMA_H16(a, b, acc[j][c]);
                    }
                }
            }
        buf ^= 1;
    }
#undef WDMA16
#undef WDMA16S
#undef WDMA_TILE
    bsum += __shfl_xor(bsum, 32);                             // the two K halves of the row
    if (o0 + wave * 32 >= O) return;                          // ragged last tile (O % 128 != 0): this wave's 32 channels do not exist
    if (do_bias && hh == 0) p.bias_slabs[(size_t)sidx * O + o0 + wave * 32 + l31] = bsum;
    float *out = p.slabs + (size_t)sidx * 9 * O * 96;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = o0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                out[((size_t)(ti * 3 + j) * O + o) * 96 + c * 32 + l31] = acc[j][c][reg];
            }
}

#pragma clang diagnostic pop
