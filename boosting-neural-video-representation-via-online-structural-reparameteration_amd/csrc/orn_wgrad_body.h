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
#define WB_LDS_BYTES (4 * 16384)                         /* dynamic LDS of a wgrad work-group: ring of 4 half-tile buffers */

struct WgradBP {
    const h16 *xpad;    // [H+2][W+2][96]
    const h16 *dypad;   // [H+2][W+2][O]
    float *slabs;       // [S][9][O][96]
    float *bias_slabs;  // [S][O]  (column sums of dy, from the ti == 1 work-groups)
    int H, W, O;
    int tiles_w, n_ktiles, S, n_otiles;
    int dbg;            // timing-only ablation flags (tools/probes)
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // the LDS-DMA asm names m0 as clobbered (nothing else in this function uses it)
// ------------------------------------------------------------------------------------------------------------------------
// The K loop is a software pipeline (round 3; the first form -- a 2-deep ring of 2-row tiles, `vmcnt(0)` + barrier per tile, fragment
// reads left to the compiler, which issued each ds_read_b64_tr_b16 pair one or two MFMAs ahead of its use -- is in the git history:
// L4 alone 189 -> 178 us, the batched launch of the 720p step 194 -> 188.5 us).
//   K unit = HALF-tile = 1 row x 32 pixels: dy 8 KiB + x 34 px x 192 B (8 KiB reserved) = 16 KiB; ring of 4 = 64 KiB per
//   work-group, still two work-groups per CU.  A half-tile is 18 MFMAs per wave (2 k-steps x 9 tiles) = 18 "positions".
//   Position i: read B fragment i+5 (2 transposed reads; fragments live in a 9-slot register ring, fragment f in slot f % 9;
//   from position 13 on they come from the NEXT half-tile's buffer) | at i % 9 == 2 the next k-step's A fragment |
//   counted lgkmcnt wait for fragment i (10 or 12 younger reads stay in flight; at most 14 outstanding: the counter has 4 bits) |
//   MFMA i.  One rendezvous per half-tile, before position 9: `vmcnt(4)` (this wave's 4 pieces of half-tile m+1 have landed,
//   those of m+2 may fly) + s_barrier; behind it every wave has consumed its fragments of half-tile m-1, so its buffer takes
//   the 4 DMA pieces of half-tile m+3, one per position 9..12.  No lgkmcnt drain anywhere in the loop.
//   Every wave issues exactly 4 pieces per half-tile (2 dy, 2 x; the 16th piece of a half-tile lies behind the x patch and
//   loads the zero pixel) and half-tiles past the end are "loaded" from the zero pixel too: the counts in the waits are
//   compile-time constants and a position is one basic block.
#define W2_HB 16384
#ifndef W2_ABL
#define W2_ABL 0                 /* tools/probes timing-only ablations (tagged builds): 1 no DMA pieces in the loop, 2 no DMA plan, 4 no B reads, 8 no rendezvous */
#endif
#define W2_XOFF 8192
#define W2_LEAD 5
struct WFrag { s16x4 lo, hi; };
__device__ __forceinline__ h16x8 w2_join(const WFrag &f)
{
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h16x8, v);
}
template <int F>
__device__ __forceinline__ void w2_read_b(WFrag &f, unsigned baddr)
{
    constexpr int half = F / 9, e = F % 9, j = e / 3, c = e % 3;
    constexpr int off = (16 * half + j) * WB_XB + c * 64;
    if constexpr ((W2_ABL & 4) != 0) return;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(baddr), "n"(off) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(baddr), "n"(off + 4 * WB_XB) : "memory");
}
template <int HALF>
__device__ __forceinline__ void w2_read_a(WFrag &f, unsigned aaddr)
{
    constexpr int off = 16 * HALF * WB_DYB;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(aaddr), "n"(off) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(aaddr), "n"(off + 4 * WB_DYB) : "memory");
}
// the wait names the registers it retires as read-write: no MFMA that consumes them can be scheduled above it
template <int N> __device__ __forceinline__ void w2_wait(WFrag &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b.lo), "+v"(b.hi) : "n"(N)); }
template <int N> __device__ __forceinline__ void w2_wait2(WFrag &b, WFrag &a)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b.lo), "+v"(b.hi), "+v"(a.lo), "+v"(a.hi) : "n"(N));
}

#define W2DMA(sbase_, voff_, ldsaddr_)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                               \
                 :: "s"(ldsaddr_), "v"(voff_), "s"((const void *)(sbase_)) : "memory", "m0")

struct W2Dma {                 // the DMA front: wave-uniform state of the next half-tile to request
    int h, tw, left;           // row, tile column, half-tiles still to request
    int dh, dw;                // S / tiles_w, S % tiles_w
};

template <int I>
__device__ __forceinline__ void w2_pos(f32x16 (&acc)[3][3], WFrag (&fb)[9], WFrag (&fa)[2], const unsigned a_cur, const unsigned b_cur,
                                       const unsigned a_nxt, const unsigned b_nxt, float &bsum)
{
    constexpr int FN = I + W2_LEAD;
    if constexpr (FN < 18) w2_read_b<FN>(fb[FN % 9], b_cur);
    else w2_read_b<FN - 18>(fb[FN % 9], b_nxt);
    if constexpr (I == 2) w2_read_a<1>(fa[1], a_cur);
    if constexpr (I == 11) w2_read_a<0>(fa[0], a_nxt);
    constexpr int N = 2 * W2_LEAD + ((I % 9 >= 2 && I % 9 <= 7) ? 2 : 0);
    if constexpr (I % 9 == 0) w2_wait2<N>(fb[I % 9], fa[I / 9]);
    else w2_wait<N>(fb[I % 9]);
    const h16x8 a = w2_join(fa[I / 9]);
    if constexpr (I % 9 == 0) {
        typedef __attribute__((ext_vector_type(2))) h16 h16v2;
        const h16v2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const h16v2 a2 = {a[2 * e], a[2 * e + 1]};
#ifdef ORN_FP16
            bsum = __builtin_amdgcn_fdot2(a2, ones2, bsum, false);
#else
            bsum = __builtin_amdgcn_fdot2_f32_bf16(a2, ones2, bsum, false);
#endif
        }
    }
    constexpr int e9 = I % 9, j = e9 / 3, c = e9 % 3;
    acc[j][c] = MFMA_H16(a, w2_join(fb[I % 9]), acc[j][c]);
    __builtin_amdgcn_sched_barrier(0);
}

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
    float bsum = 0.f;              // dbias = column sums of the dy fragment (v_dot2c under the MFMAs); the ti == 1 work-groups store it
    const bool do_bias = (ti == 1);

    const unsigned wlds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // per-lane transposed-read addresses inside a half-tile buffer.  dy: this lane's pixels all have (pixel & 3) == lq, so the
    // XOR swizzle is a per-lane constant.
    const int a_chunk = uwave * 4 + 2 * (g & 1) + (lp >> 1);
    const unsigned a_lds = wlds0 + (8 * (g >> 1) + lq) * WB_DYB + ((a_chunk ^ (lq << 2)) * 16) + (lp & 1) * 8;
    const unsigned b_lds = wlds0 + W2_XOFF + (8 * (g >> 1) + lq) * WB_XB + (16 * (g & 1) + 4 * lp) * 2;

    // DMA plan: pieces q = uwave, uwave + 4 of the 8 dy instructions (16-byte slot L = 64 q + lane: pixel L >> 4, chunk L & 15)
    // and of the 8 x instructions (pixel L / 12, chunk L % 12; pixels >= 34 lie behind the patch).  Interior half-tiles take
    // a wave-uniform base + these per-lane byte offsets; ragged ones (last tile column) and the filler loads recompute them.
    unsigned loff[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int L = (uwave + 4 * k) * 64 + lane, px = L >> 4, cc = (L & 15) ^ ((px & 3) << 2);
        loff[k] = (unsigned)((px + 1) * O + o0 + cc * 8) * 2u;
        const int Lx = (uwave + 4 * k) * 64 + lane, pxx = Lx / 12, cx = Lx - pxx * 12;
        loff[2 + k] = pxx < WB_XW ? (unsigned)(pxx * 96 + cx * 8) * 2u : 0u;
    }
    const int tiles_w = p.tiles_w;
    const int NH = H * tiles_w;                                   // half-tiles of the layer; this work-group takes sidx, sidx + S, ..
    const int n_my = sidx < NH ? (NH - sidx + p.S - 1) / p.S : 0;
    W2Dma d;
    d.h = sidx / tiles_w; d.tw = sidx - d.h * tiles_w; d.left = n_my;
    d.dh = p.S / tiles_w; d.dw = p.S - d.dh * tiles_w;
    // sources advance by wave-uniform byte strides (64-bit adds on the scalar unit; no multiplies in the loop)
    const unsigned char *pdy = (const unsigned char *)(p.dypad + ((size_t)(d.h + 1) * (W + 2) + d.tw * WB_TW) * O);
    const unsigned char *pxs = (const unsigned char *)(p.xpad + ((size_t)(d.h + ti) * (W + 2) + d.tw * WB_TW) * 96);
    const long pix_step = (long)d.dh * (W + 2) + d.dw * WB_TW;          // pixels from one half-tile of this work-group to its next
    const long pix_wrap = (long)(W + 2) - (long)tiles_w * WB_TW;          // .. extra when the tile column wraps into the next row
    const long dy_step = pix_step * O * 2, dy_wrap = pix_wrap * O * 2, x_step = pix_step * 192, x_wrap = pix_wrap * 192;

    auto dma_plan = [&](unsigned (&vo)[4], const unsigned char *&sdy, const unsigned char *&sx) {
        // source of the next half-tile (or the zero pixel when none is left)
        const int w0 = d.tw * WB_TW;
        const bool has = d.left > 0;
        const bool interior = has && (w0 + WB_TW <= W);
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = loff[k];
        sdy = pdy; sx = pxs;
        if (!interior) {
            sdy = (const unsigned char *)p.dypad; sx = (const unsigned char *)p.xpad;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int L = (uwave + 4 * k) * 64 + lane, px = L >> 4, cc = (L & 15) ^ ((px & 3) << 2);
                const bool ok = has && (w0 + px < W);
                vo[k] = ok ? (unsigned)(((d.h + 1) * (W + 2) + w0 + px + 1) * O + o0 + cc * 8) * 2u : (unsigned)(cc * 8) * 2u;   // else the border pixel (0,0): zeros
                const int pxx = L / 12, cx = L - pxx * 12;
                const bool okx = has && pxx < WB_XW && (w0 + pxx < W + 2);
                vo[2 + k] = okx ? (unsigned)(((d.h + ti) * (W + 2) + w0 + pxx) * 96 + cx * 8) * 2u : (unsigned)(cx * 8) * 2u;
            }
        }
        d.left -= 1;
        d.tw += d.dw; d.h += d.dh;
        pdy += dy_step; pxs += x_step;
        if (d.tw >= tiles_w) { d.tw -= tiles_w; d.h += 1; pdy += dy_wrap; pxs += x_wrap; }
    };
    auto dma_piece = [&](const int k, const unsigned (&vo)[4], const unsigned char *sdy, const unsigned char *sx, const unsigned slot_lds) {
        if (k < 2) W2DMA(sdy, vo[k], slot_lds + (uwave + 4 * k) * 1024);
        else W2DMA(sx, vo[k], slot_lds + W2_XOFF + (uwave + 4 * (k - 2)) * 1024);
    };

    // The 16 pieces of a half-tile are dealt one per position over the 16 positions that follow the rendezvous (9..17 of this
    // half-tile, 0..6 of the next): linear slot n = 4 k + w is piece k of wave w, so the four waves never issue in the same
    // MFMA gap (all four issuing behind the barrier cost 23 us of the L4 launch: tools/probes/wgrad_variants.sh).  The plan of
    // the pieces that fall behind the loop's back edge (k = 2 of waves 1..3, k = 3) is carried.
    auto dma_at = [&](const int lin, const unsigned (&vo)[4], const unsigned char *sdy, const unsigned char *sx, const unsigned slot_lds) {
        if ((W2_ABL & 1) != 0) return;
#ifdef W2_SPREAD            /* one piece per position and work-group: measured slower (L4 alone 184.5 vs 180 us), see DESIGN 4.5 */
        if (lin < 16 && uwave == (lin & 3)) dma_piece(lin >> 2, vo, sdy, sx, slot_lds);
#else                       /* every wave's piece k at linear slot k: four pieces per gap right behind the rendezvous */
        if (lin < 4) dma_piece(lin, vo, sdy, sx, slot_lds);
#endif
    };
    // prologue: half-tiles 0, 1 into slots 0, 1 and the first half (linear slots 0..8) of half-tile 2
    unsigned vo_c[4]; const unsigned char *sdy_c, *sx_c; unsigned dst_c = wlds0 + 2 * W2_HB;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        unsigned vo[4]; const unsigned char *sdy, *sx;
        dma_plan(vo, sdy, sx);
#pragma unroll
        for (int k = 0; k < 4; ++k) dma_piece(k, vo, sdy, sx, wlds0 + m * W2_HB);
    }
    dma_plan(vo_c, sdy_c, sx_c);
#pragma unroll
    for (int lin = 0; lin < 9; ++lin) dma_at(lin, vo_c, sdy_c, sx_c, dst_c);
    WFrag fb[9], fa[2];
#ifdef W2_SPREAD
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // half-tile 0 (this wave's first 4 pieces) has landed
#else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    unsigned a_cur = a_lds, b_cur = b_lds;
    w2_read_a<0>(fa[0], a_cur);
    w2_read_b<0>(fb[0], b_cur); w2_read_b<1>(fb[1], b_cur); w2_read_b<2>(fb[2], b_cur); w2_read_b<3>(fb[3], b_cur); w2_read_b<4>(fb[4], b_cur);
    for (int m = 0; m < n_my; ++m) {
        const unsigned nxt = (unsigned)((m + 1) & 3) * W2_HB;
        const unsigned a_nxt = a_lds + nxt, b_nxt = b_lds + nxt;
#define W2_P(i_)                                                                                                \
        w2_pos<i_>(acc, fb, fa, a_cur, b_cur, a_nxt, b_nxt, bsum);                                              \
        dma_at((i_) >= 9 ? (i_) - 9 : (i_) + 9, vo_c, sdy_c, sx_c, dst_c);
        W2_P(0) W2_P(1) W2_P(2) W2_P(3) W2_P(4) W2_P(5) W2_P(6) W2_P(7) W2_P(8)
        // rendezvous: half-tile m+1 is in LDS for everyone; everyone is done with half-tile m-1, whose slot takes m+3
        if constexpr ((W2_ABL & 2) == 0) dma_plan(vo_c, sdy_c, sx_c);
        dst_c = wlds0 + (unsigned)((m + 3) & 3) * W2_HB;
        if constexpr ((W2_ABL & 8) == 0) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        W2_P(9) W2_P(10) W2_P(11) W2_P(12) W2_P(13) W2_P(14) W2_P(15) W2_P(16) W2_P(17)
#undef W2_P
        a_cur = a_nxt; b_cur = b_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the filler loads and the run-ahead reads: drain before the LDS goes away
    // keep the run-ahead fragments "used" so their reads are not dropped as dead (they are asm volatile: nothing to do)

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
#undef W2DMA

#pragma clang diagnostic pop
