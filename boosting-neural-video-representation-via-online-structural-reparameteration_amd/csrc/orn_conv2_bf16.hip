// A4 fast path, large images: conv3x3 forward (+ bias + PixelShuffle, + SiLU copy for the next block; model.py:539,567) and its
// dgrad (x SiLU'(z_prev), scattered into the previous block's padded gradient) as ONE kernel family built for TWO work-groups
// per CU, on v_mfma_f32_16x16x32 (fp32 accumulate).
//
// The first forms (orn_conv_fwd_bf16.hip, orn_conv_bf16.hip) keep a 64 KB patch of 96 input channels + a weight ring in LDS: one
// work-group per CU, and every bubble of that work-group -- the patch load at its start, the exposed patch switch between channel
// chunks, the rendezvous of every tap, the epilogue -- idles the matrix pipe (phase stamps: 41 % MFMA busy).  Here the K chunk is
// 32 channels: a patch is 21 KB, so TWO patches (the next chunk streams in while this one is consumed: no exposed switch) + a
// 4-deep ring of [96][32] weight tiles are 68 KB, and two 4-wave work-groups share a CU.  Each SIMD then hosts one wave of each
// work-group: whenever one of them waits -- rendezvous, prologue, epilogue -- the other owns the matrix pipe.
//   work-group = 8 x 32 output pixels x 96 output channels; wave w: rows 2w, 2w+1 (wave tile 64 px x 96 ch, 24 accumulator
//   tiles); K stream = chunks x 9 taps, one k-step (24 MFMAs per wave) per (chunk, tap), epilogue at the end.
//     dgrad:   Cin / 32 chunks (Cin = the layer's conv output channels);
//     forward: one 96-channel N tile = 3 chunks (K = 9 x 96) per work-group; the O / 96 work-groups of a pixel tile follow each
//              other on one XCD.  With 96 channels per output pixel (Cn == 96) an N tile is ALL channels of one output
//              sub-position, so the last block's epilogue CAN also run the A5 head (1x1 conv 96 -> 3 + activation) on them
//              (ORN_HEAD_FUSED=1; measured slower than the separate HBM-bound head kernel, see orn_launch_fwd2).
//   Pipeline per step u: fragment reads of u+1 (other register set) | counted lgkmcnt wait for u | counted vmcnt wait + rendezvous
//   | 24 MFMAs | DMA of weight tile u+4 into the slot of tile u | (taps 0..5) one DMA piece of the NEXT chunk's patch.
//   What is known to have landed after rendezvous v: weight tiles <= v+2 and the patch pieces issued up to step v-2, because the
//   wait before rendezvous v leaves exactly the operations issued after rendezvous v-1 in flight (vmcnt counts in issue order).
//   The body is compiled once per wave (its DMA role: which pieces of a step it issues), so those waits are compile-time
//   constants and a tap is one basic block (a runtime branch inside it cost 50 us per launch: the scheduler no longer
//   interleaves the next reads with the MFMAs).
// Compiled twice like its siblings: as is (bf16, namespace orn_bf16) and with -DORN_FP16 (IEEE half, namespace orn_f16).
#include "orn_internal.h"
#include <type_traits>
#ifdef ORN_FP16
#define HNS orn_f16
typedef _Float16 h16;
#define MFMA16_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
#define HNS orn_bf16
typedef __bf16 h16;
#define MFMA16_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define PDBG(p_) 0
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void c2_sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        c2_sfor<I + 1, N>(f);
    }
}

namespace HNS {

#define C2_TH 8
#define C2_TW 32
#define C2_PH (C2_TH + 2)
#define C2_PW (C2_TW + 2)
#define C2_CK 32
#define C2_ROWB 64                                    // LDS bytes per patch pixel / weight row (32 halfs)
#define C2_PATCH_INSTR ((C2_PH * C2_PW * C2_ROWB + 1023) / 1024)      // 22 DMA wave-instructions per patch
#define C2_PATCH_LDS (C2_PATCH_INSTR * 1024)
#define C2_TILE_BYTES (96 * C2_ROWB)                  // 6 DMA wave-instructions per weight tile
#ifndef C2_NSLOT
#define C2_NSLOT 4                                     // 5 and 6 (more DMA in flight at every rendezvous) time the same: issue-bound, not latency-bound
#endif
#define C2_LDS (2 * C2_PATCH_LDS + C2_NSLOT * C2_TILE_BYTES)

enum { C2_DGRAD = 0, C2_FWD = 1, C2_FWD_LAST = 2 };

struct Conv2P {
    const h16 *xpad;     // [H+2][W+2][Cx]   (forward: the block's padded input; dgrad: the padded gradient of the conv output)
    const h16 *w;        // forward: [9][Nout][96]; dgrad: [9][96][Cx]
    int H, W, Cx;
    int wrow, wtap;      // elements between two weight rows / BYTES between two taps
    int qseg;            // chunks of 32 channels per work-group
    int tiles_w, tiles_h, ptiles, nsplit;   // forward: `nsplit` = O / 96 work-groups share a pixel tile, one N tile each
    // dgrad epilogue
    const h16 *zprev;    // [H][W][96]
    h16 *dyprev;         // [H/sp+2][W/sp+2][96*sp*sp]
    int sp;
    unsigned mSp;
    // forward epilogue
    const float *bias;   // [Nout] (o' order) or null
    h16 *z;              // [H*s][W*s][Cn]
    h16 *apad;           // [H*s+2][W*s+2][Cn] or null
    int s, Cn, Nout;
    unsigned z_bytes, apad_bytes;   // sizes of the two buffers (raw-buffer bounds)
    unsigned mCn, mS;
    // A5 head in the last block's epilogue (C2_FWD_LAST, Cn == 96: an N tile is all channels of one output sub-position)
    const float *head_w, *head_b;   // [3][96], [3]
    float *head_out;                // fp32 [3][H*s][W*s], or null: no head
    int head_sigmoid;
};

__device__ __forceinline__ int c2_div(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }
// m with x / d == umulhi(x, m) for every 0 <= x < 2^16 and 2 <= d < 2^16; d == 1 is encoded as m = 0
static unsigned c2_magic(int d)
{
    return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d);
}
// v_permlane16_swap: odd 16-lane rows of `a` <-> even rows of `b`.  Afterwards rows 0 / 2 hold (own a, the next row's a) and
// rows 1 / 3 hold (the previous row's b, own b) -- checked on hardware with tools/probes (row = lane >> 4).
// (keep the operands named lvalues: with bit_cast temporaries as arguments hipcc 7.2 returned a wrong second half)
__device__ __forceinline__ void c2_swap_rows(unsigned &a, unsigned &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void c2_swap_rows_f(float &a, float &b)
{
    unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    c2_swap_rows(ua, ub);
    a = __builtin_bit_cast(float, ua);
    b = __builtin_bit_cast(float, ub);
}
__device__ __forceinline__ unsigned c2_pack(float lo, float hi)
{
    h16x2 v;
    v[0] = (h16)lo;
    v[1] = (h16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// Fragment reads of step (tap TAP) into set SET: 4 pixel sub-blocks (rows 2w + {0,1} + ti, two 16-pixel halves) and 6 channel
// sub-blocks.  LDS rows are 64 B = 4 chunks of 16 B; logical chunk c of row R sits at position c ^ ((R >> 1) & 3) (conflict-free
// for the lane groups of ds_read_b128).  The swizzle term of a patch read, 16 * (g4 ^ ((pix >> 1) & 3)) with pix = 68 w + l15 +
// 34 r + 16 h + tj (w wave, r patch row relative to the wave's first, h half, tj tap column), only depends on
// (2 w + r + ((l15 + tj) >> 1)) & 3: eight per-lane values sw[r & 3][x], x = 0 for tj = 0, 1 for tj = 1, and tj = 2 is
// (r + 1, x = 0).  They are computed once; a read's address is one add (patch buffer base + term), the pixel offset is in the
// instruction's immediate.
template <int SET, int TAP, int I>
__device__ __forceinline__ void c2_read_a(h16x8 (&fa)[2][4], unsigned a_base, const unsigned (&sw)[4][2])
{
    constexpr int ti = TAP / 3, tj = TAP - ti * 3;
    constexpr int r = (I >> 1) + ti;
    constexpr unsigned pixoff = r * C2_PW + 16 * (I & 1) + tj;
    const unsigned addr = a_base + sw[(r + (tj == 2 ? 1 : 0)) & 3][tj == 1 ? 1 : 0];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[SET][I]) : "v"(addr), "n"(pixoff * C2_ROWB) : "memory");
}
template <int SET, int TAP>
__device__ __forceinline__ void c2_read_step(h16x8 (&fa)[2][4], h16x8 (&fb)[2][6], unsigned a_base, const unsigned (&sw)[4][2], unsigned w_base)
{
    c2_read_a<SET, TAP, 0>(fa, a_base, sw);
    c2_read_a<SET, TAP, 1>(fa, a_base, sw);
    c2_read_a<SET, TAP, 2>(fa, a_base, sw);
    c2_read_a<SET, TAP, 3>(fa, a_base, sw);
    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[SET][0]) : "v"(w_base) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][1]) : "v"(w_base), "n"(16 * C2_ROWB) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][2]) : "v"(w_base), "n"(32 * C2_ROWB) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][3]) : "v"(w_base), "n"(48 * C2_ROWB) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][4]) : "v"(w_base), "n"(64 * C2_ROWB) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][5]) : "v"(w_base), "n"(80 * C2_ROWB) : "memory");
}
// The wait that retires register set SET (its reads were issued before the PEND newest ones) names every register of the set
// as read-write, so no MFMA that consumes them can be scheduled above it.
template <int SET, int PEND>
__device__ __forceinline__ void c2_wait_set(h16x8 (&fa)[2][4], h16x8 (&fb)[2][6])
{
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(fa[SET][0]), "+v"(fa[SET][1]), "+v"(fa[SET][2]), "+v"(fa[SET][3]), "+v"(fb[SET][0]), "+v"(fb[SET][1]),
                 "+v"(fb[SET][2]), "+v"(fb[SET][3]), "+v"(fb[SET][4]), "+v"(fb[SET][5]) : "n"(PEND));
}
template <int SET>
__device__ __forceinline__ void c2_mfma_step(h16x8 (&fa)[2][4], h16x8 (&fb)[2][6], f32x4 (&acc)[4][6])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = MFMA16_H16(fb[SET][j], fa[SET][i], acc[i][j]);
}
template <int N> __device__ __forceinline__ void c2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// DMA roles.  A step's pieces -- the 6 of weight tile u+4, then (taps 0..5) 4, 4, 4, 4, 3, 3 of the next patch's 22 -- are dealt
// round-robin to the four waves: wave R issues the pieces at list positions R, R+4, R+8.  The per-wave count of a step differs
// between waves, and the counted waits need it as an immediate, so the whole body is compiled once per wave (template
// argument ROLE, selected once at the top of the kernel: no branch inside the loop): 76 DMA instructions per chunk and
// work-group instead of the 108 of a role-free stream that pads every wave to the same count with pieces loaded twice.
constexpr int c2_wcnt(int role) { return role < 2 ? 2 : 1; }                              // weight pieces j = role, role + 4 (< 6)
constexpr int c2_pn(int t) { return t < 4 ? 4 : (t < 6 ? 3 : 0); }                        // patch pieces dealt at tap t
constexpr int c2_pbase(int t) { int b = 0; for (int i = 0; i < t; ++i) b += c2_pn(i); return b; }
constexpr int c2_pidx(int role, int t) { const int n = (role + 2) & 3; return n < c2_pn(t) ? c2_pbase(t) + n : -1; }   // list position 6 + n has (6 + n) % 4 == role
constexpr int c2_cnt(int role, int t) { return c2_wcnt(role) + (c2_pidx(role, t) >= 0 ? 1 : 0); }
static_assert(c2_pbase(6) == C2_PATCH_INSTR && C2_NSLOT == 4, "conv2: the piece deal assumes 22 patch pieces and the 4-deep ring (LAG 1)");

template <int EPI, int ROLE>
__device__ __forceinline__ void c2_body(const Conv2P &p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    constexpr int uwave = ROLE;
    const int l15 = lane & 15, g4 = lane >> 4;
    // forward: the `nsplit` work-groups of a pixel tile follow each other on one XCD (work-groups go round the 8 XCDs), so the
    // patch the first of them pulls from HBM is an L2 hit for the others
    int tile = blockIdx.x, seg0 = 0;
    if (EPI != C2_DGRAD) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        const int ns = r % p.nsplit;
        tile = (r / p.nsplit) * 8 + xcd;
        if (tile >= p.ptiles) return;
        seg0 = ns;
    }
    const int tw = tile % p.tiles_w, th = tile / p.tiles_w;
    const int h0 = th * C2_TH, w0 = tw * C2_TW;
    const int H = p.H, W = p.W, Cx = p.Cx;
    const int QS = p.qseg;

    // DMA plans: per-lane SOURCE byte offsets of this wave's pieces; destination = lane-linear 1 KiB per instruction.
    unsigned p_goff[6], b_goff[2];                       // patch piece of tap k (0..5, if this role has one); weight pieces j = ROLE, ROLE + 4
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int m = c2_pidx(ROLE, k) >= 0 ? c2_pidx(ROLE, k) : 0;
        const int L = m * 64 + lane, pix = L >> 2, pos = L & 3;
        const int c = pos ^ ((pix >> 1) & 3);
        const int pr = pix / C2_PW, pc = pix - pr * C2_PW;
        const int gh = h0 + pr, gw = w0 + pc;
        const bool ok = (pix < C2_PH * C2_PW) && gh < H + 2 && gw < W + 2;       // others read the all-zero border pixel (0,0)
        p_goff[k] = (unsigned)((ok ? (gh * (W + 2) + gw) * Cx : 0) + c * 8) * 2u;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int m = ROLE + 4 * k;                      // (k = 1 only exists for roles 0 and 1)
        const int L = m * 64 + lane, R = L >> 2, pos = L & 3;
        b_goff[k] = (unsigned)(R * p.wrow + (pos ^ ((R >> 1) & 3)) * 8) * 2u;
    }
#define C2_DMA16(gptr_, ldsoff_)                                                                                \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr_),                   \
                                     (__attribute__((address_space(3))) void *)(smem + (ldsoff_)), 16, 0, 0)
    // this wave's patch piece of tap k_ (if it has one) of source chunk qsrc_ into patch buffer buf_
#define C2_DMA_PATCH_PIECE(qsrc_, buf_, k_)                                                                     \
    if constexpr (c2_pidx(ROLE, (k_)) >= 0)                                                                     \
        C2_DMA16((const char *)p.xpad + (size_t)(qsrc_) * (C2_CK * 2) + p_goff[k_], (buf_) * C2_PATCH_LDS + c2_pidx(ROLE, (k_)) * 1024);
    // this wave's pieces of the weight tile (rows and chunk of wq_, tap tap_) into ring slot slot_
#define C2_DMA_TILE_AT(wq_, tap_, slot_)                                                                        \
    {                                                                                                           \
        const char *wb__ = (const char *)(wq_) + (size_t)(tap_) * p.wtap;                                       \
        _Pragma("unroll") for (int k = 0; k < c2_wcnt(ROLE); ++k)                                               \
            C2_DMA16(wb__ + b_goff[k], 2 * C2_PATCH_LDS + (slot_) * C2_TILE_BYTES + (ROLE + 4 * k) * 1024);     \
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned pix_lane = (2 * uwave) * C2_PW + l15;                                  // patch pixel of (row 2w, column l15)
    const unsigned a_lane = lds0 + pix_lane * C2_ROWB;
    const unsigned b_lane = lds0 + 2 * C2_PATCH_LDS + l15 * C2_ROWB + 16 * (g4 ^ ((l15 >> 1) & 3));
    unsigned sw[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            sw[r][x] = 16 * (g4 ^ ((2 * uwave + r + ((l15 + x) >> 1)) & 3));
            asm volatile("" : "+v"(sw[r][x]));
        }
    // forward: this work-group's biases behind the ring
    float *sbias = reinterpret_cast<float *>(smem + C2_LDS);
    if (EPI != C2_DGRAD)
        for (int i = t; i < 96; i += 256) sbias[i] = p.bias ? p.bias[seg0 * 96 + i] : 0.f;
    float *shead = sbias + 96;                    // [3][96] head weights + [3] biases (+ pad)
    const bool head = (EPI == C2_FWD_LAST) && p.head_out != nullptr;
    if (EPI == C2_FWD_LAST && head)
        for (int i = t; i < 3 * 96 + 3; i += 256) shead[i] = i < 3 * 96 ? p.head_w[i] : p.head_b[i - 3 * 96];
    // z / apad as raw buffers: a byte offset of 0x80000000 (out of range for any buffer the launcher admits) drops the lane's
    // store, so every lane issues every store and the number of vector-memory operations of an epilogue is a constant
    const auto z_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.z, 0, EPI != C2_DGRAD ? p.z_bytes : 0, 0x00020000);
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.apad, 0, EPI == C2_FWD ? p.apad_bytes : 0, 0x00020000);
    const int c8_lane = 16 * (g4 & 1) + 8 * (g4 >> 1);    // first of the 8 channels (of a 32-channel block) this lane stores

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 fa[2][4], fb[2][6];

    // prologue: patch of chunk 0, weight tiles 0..R-2, rendezvous; tile R-1 stays in flight behind it
    const h16 *w_cur = p.w + (size_t)seg0 * 96 * p.wrow;
    c2_sfor<0, 6>([&](auto k_c) __attribute__((always_inline)) { C2_DMA_PATCH_PIECE(0, 0, decltype(k_c)::value) });
#pragma unroll
    for (int u0 = 0; u0 < C2_NSLOT - 1; ++u0) C2_DMA_TILE_AT(w_cur, u0, u0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    C2_DMA_TILE_AT(w_cur, C2_NSLOT - 1, C2_NSLOT - 1)
    c2_read_step<0, 0>(fa, fb, a_lane, sw, b_lane);

    int slot_c = 0;                                      // (9 c) mod R: ring slot of the chunk's first tile
    for (int c = 0; c < QS; ++c) {
        const int qn = (c + 1 < QS) ? c + 1 : 0;         // the chunk after this one (past the end: wraps to the first, never consumed)
        const h16 *w_nxt = p.w + (size_t)seg0 * 96 * p.wrow + qn * C2_CK;
        const unsigned a_cur = a_lane + (c & 1) * C2_PATCH_LDS;
        const unsigned a_nxt = a_lane + ((c + 1) & 1) * C2_PATCH_LDS;
        // nine taps: an odd count, so the set that tap 8 prefetched into (set 1) is handed over to set 0 between chunks --
        // 40 register moves per chunk against 216 MFMAs, and no second copy of the loop body
        if (c > 0) {
            c2_wait_set<1, 0>(fa, fb);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = fa[1][i];
#pragma unroll
            for (int j = 0; j < 6; ++j) fb[0][j] = fb[1][j];
        }
        c2_sfor<0, 9>([&](auto tap_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int cur = tap & 1, nxt = cur ^ 1;
            // fragment reads of step u + 1 (its tile is known to have landed since rendezvous u - 1); the reads issued by the
            // very last step go to wrapped-around data and are never consumed
            const unsigned wslot = b_lane + ((slot_c + tap + 1) % C2_NSLOT) * C2_TILE_BYTES;
            if constexpr (tap < 8) c2_read_step<nxt, tap + 1>(fa, fb, a_cur, sw, wslot);
            else c2_read_step<nxt, 0>(fa, fb, a_nxt, sw, wslot);
            c2_wait_set<cur, 10>(fa, fb);
            // rendezvous u in front of the step's MFMAs.  What has to be known as landed here: weight tile u + 2 (issued at step
            // u + 2 - R) and, at tap 7, the next chunk's patch: everything issued up to step u - (R - 2).  So the operations of
            // the LAG = R - 3 steps before this one stay in flight: per step two weight pieces + that step's patch pieces.
            // (A forward epilogue's stores are older than all of these: the first rendezvous after it waits for them.)
            c2_wait_vm<c2_cnt(ROLE, (tap + 8) % 9)>();
            __builtin_amdgcn_s_barrier();
            c2_mfma_step<cur>(fa, fb, acc);
#ifndef C2_ABL_NO_WDMA            // compile-time timing ablations (tools/probes/abl_conv2.sh)
            C2_DMA_TILE_AT((tap + C2_NSLOT < 9) ? w_cur : w_nxt, (tap + C2_NSLOT) % 9, (slot_c + tap) % C2_NSLOT)   // into the slot of tile u: everyone is past its reads
#endif
#ifndef C2_ABL_NO_PDMA
            C2_DMA_PATCH_PIECE(qn, (c + 1) & 1, tap)                                       // next chunk's patch
#endif
        });
        slot_c = (slot_c + 9) % C2_NSLOT;

        w_cur = w_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           // drain the wrapped-around loads before the LDS goes away
#undef C2_DMA16
#undef C2_DMA_PATCH_PIECE
#undef C2_DMA_TILE_AT

    if (EPI != C2_DGRAD) {
        // ---- forward epilogue: + bias, PixelShuffle scatter of z (and of a = SiLU(z) into the next block's
        // padded input), 8 channels = 16 B per lane and store -------------------------------------------------------------
        const int n0 = seg0 * 96;
        float hu[4][3];                               // head: this lane's partial W . SiLU(z) of its 4 pixels
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) hu[pi][0] = hu[pi][1] = hu[pi][2] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c8 = n0 + j * 32 + c8_lane;
            const float4 ba = *reinterpret_cast<const float4 *>(sbias + j * 32 + c8_lane);
            const float4 bb = *reinterpret_cast<const float4 *>(sbias + j * 32 + c8_lane + 4);
            const int ij = c2_div(c8, p.mCn), n = c8 - ij * p.Cn;
            const int si = c2_div(ij, p.mS), sj = ij - si * p.s;
            float hw[3][8];
            if (EPI == C2_FWD_LAST && head) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(shead + o * 96 + j * 32 + c8_lane);
                    const float4 w1 = *reinterpret_cast<const float4 *>(shead + o * 96 + j * 32 + c8_lane + 4);
                    hw[o][0] = w0.x; hw[o][1] = w0.y; hw[o][2] = w0.z; hw[o][3] = w0.w;
                    hw[o][4] = w1.x; hw[o][5] = w1.y; hw[o][6] = w1.z; hw[o][7] = w1.w;
                }
            }
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
                const int gh = h0 + 2 * uwave + (pi >> 1), gw = w0 + 16 * (pi & 1) + l15;
                const bool ok = (gh < H) && (gw < W);
                const f32x4 ta = acc[pi][2 * j], tb = acc[pi][2 * j + 1];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = ta[e], x1 = tb[e];
                    c2_swap_rows_f(x0, x1);
                    v[e] = x0; v[4 + e] = x1;
                }
                v[0] += ba.x; v[1] += ba.y; v[2] += ba.z; v[3] += ba.w;
                v[4] += bb.x; v[5] += bb.y; v[6] += bb.z; v[7] += bb.w;
                const int Ws = W * p.s, oh = gh * p.s + si, ow = gw * p.s + sj;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{c2_pack(v[0], v[1]), c2_pack(v[2], v[3]), c2_pack(v[4], v[5]), c2_pack(v[6], v[7])}, z_rsrc,
                                                       ok ? ((oh * Ws + ow) * p.Cn + n) * 2 : (int)0x80000000, 0, 0);   // < 2^31 bytes: launcher
                if (EPI == C2_FWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = orn_silu(v[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{c2_pack(v[0], v[1]), c2_pack(v[2], v[3]), c2_pack(v[4], v[5]), c2_pack(v[6], v[7])}, a_rsrc,
                                                           ok ? (((oh + 1) * (Ws + 2) + (ow + 1)) * p.Cn + n) * 2 : (int)0x80000000, 0, 0);
                }
                if (EPI == C2_FWD_LAST && head) {
                    // on the 16-bit z that was just stored: what the separate head kernel reads back, and what the backward uses
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = orn_silu((float)(h16)v[e]);
                        hu[pi][0] = fmaf(hw[0][e], a, hu[pi][0]);
                        hu[pi][1] = fmaf(hw[1][e], a, hu[pi][1]);
                        hu[pi][2] = fmaf(hw[2][e], a, hu[pi][2]);
                    }
                }
            }
        }
        if (EPI == C2_FWD_LAST && head) {
            // the four lanes l15 + 16 g of a pixel hold 24 channels each: sum them, lane g < 3 stores output channel g
            const int ij = c2_div(n0, p.mCn), si = c2_div(ij, p.mS), sj = ij - si * p.s;
            const size_t HWs = (size_t)(H * p.s) * (W * p.s);
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
                const int gh = h0 + 2 * uwave + (pi >> 1), gw = w0 + 16 * (pi & 1) + l15;
                float u = 0.f;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    float x = hu[pi][o];
                    x += __shfl_xor(x, 16, 64);
                    x += __shfl_xor(x, 32, 64);
                    if (g4 == o) u = x;
                }
                if (g4 < 3 && gh < H && gw < W) {
                    u += shead[3 * 96 + g4];
                    const int oh = gh * p.s + si, ow = gw * p.s + sj;
                    p.head_out[(size_t)g4 * HWs + (size_t)oh * (W * p.s) + ow] = p.head_sigmoid ? 1.0f / (1.0f + __expf(-u)) : (tanhf(u) + 1.0f) * 0.5f;
                }
            }
        }
    }
    if (EPI == C2_DGRAD) {
        // ---- dgrad epilogue: x SiLU'(z_prev), scatter into the previous block's padded gradient -----------------------------
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int gh = h0 + 2 * uwave + (pi >> 1), gw = w0 + 16 * (pi & 1) + l15;
            const bool ok = (gh < H) && (gw < W);
            const int sp = p.sp, ph = c2_div(gh, p.mSp), pw = c2_div(gw, p.mSp);
            const int sub = (gh - ph * sp) * sp + (gw - pw * sp);
            h16x8 zz[3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ok) zz[j] = *reinterpret_cast<const h16x8 *>(p.zprev + ((size_t)gh * W + gw) * 96 + j * 32 + c8_lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c8 = j * 32 + c8_lane;
                const f32x4 ta = acc[pi][2 * j], tb = acc[pi][2 * j + 1];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = ta[e], x1 = tb[e];
                    c2_swap_rows_f(x0, x1);
                    v[e] = x0; v[4 + e] = x1;
                }
                if (ok) {
                    h16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (h16)(v[e] * orn_silu_grad((float)zz[j][e]));
                    *reinterpret_cast<h16x8 *>(p.dyprev + ((size_t)(ph + 1) * (W / sp + 2) + (pw + 1)) * (96 * sp * sp) + sub * 96 + c8) = o8;
                }
            }
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(256, 2) k_conv2_nhwc(Conv2P p)
{
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {       // one copy of the body per DMA role (wave-uniform)
    case 0: c2_body<EPI, 0>(p); break;
    case 1: c2_body<EPI, 1>(p); break;
    case 2: c2_body<EPI, 2>(p); break;
    default: c2_body<EPI, 3>(p); break;
    }
}

template <int EPI>
static int c2_launch(const Conv2P &p, int blocks, size_t lds, hipStream_t st, const char *what)
{
    // (one process drives one device: the attribute is set once per process, see include/orn.h)
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)k_conv2_nhwc<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C2_LDS + 2048));
        if (e != hipSuccess) { orn_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    hipLaunchKernelGGL(k_conv2_nhwc<EPI>, dim3(blocks), dim3(256), lds, st, p);
    ORN_LAUNCH_CHECK(what);
    return 0;
}

// dgrad of a block whose input image has >= 128 pixel tiles: dx = conv_transpose(dy) x SiLU'(z_prev) into the previous block's
// dypad.  dypad [H+2][W+2][O], wd [9][96][O] (+ slack: see orn_conv_bf16_wd_elems), O % 32 == 0.
int orn_launch_dgrad2(const h16 *dypad, const h16 *wd, int H, int W, int O, const h16 *zprev, h16 *dyprev, int sp, hipStream_t st)
{
    ORN_REQUIRE(O % C2_CK == 0 && zprev && dyprev && sp >= 1 && sp < 65536 && H % sp == 0 && W % sp == 0 && H < 65536 && W < 65536,
                "conv_bf16_dgrad: unsupported O=%d sp=%d", O, sp);
    Conv2P p = {};
    p.xpad = dypad; p.w = wd; p.H = H; p.W = W; p.Cx = O;
    p.wrow = O; p.wtap = 96 * O * 2;                      // (wtap in BYTES)
    p.qseg = O / C2_CK;
    p.tiles_w = orn_cdiv(W, C2_TW); p.tiles_h = orn_cdiv(H, C2_TH); p.ptiles = p.tiles_w * p.tiles_h; p.nsplit = 1;
    p.zprev = zprev; p.dyprev = dyprev; p.sp = sp; p.mSp = c2_magic(sp);
    return c2_launch<C2_DGRAD>(p, p.ptiles, C2_LDS, st, "dgrad2_nhwc");
}

// forward of a block with 96 input channels and O % 96 == 0 output channels.  Returns -1 without launching when the shape is
// not this form's (the caller falls back to the first form).
int orn_launch_fwd2(const h16 *xpad, const h16 *wb, const float *bias_p, int H, int W, int O, int s, h16 *z, h16 *apad, hipStream_t st, OrnHeadFuse *head)
{
    if (O % 96 != 0 || O > 2048) return -1;
    Conv2P p = {};
    p.xpad = xpad; p.w = wb; p.H = H; p.W = W; p.Cx = 96;
    p.wrow = 96; p.wtap = O * 96 * 2;                     // (wtap in BYTES)
    p.qseg = 3;
    p.tiles_w = orn_cdiv(W, C2_TW); p.tiles_h = orn_cdiv(H, C2_TH); p.ptiles = p.tiles_w * p.tiles_h;
    const int NT = O / 96;
    // One work-group per (pixel tile, N tile).  (Several N tiles per work-group, the epilogue inside the loop and the streams
    // running on across it, measured 160 / 162 us at 1 / 2 work-groups per tile against 154 us: an epilogue's stores sit in
    // front of the next rendezvous' counted wait, and its registers spill into the tap loop.)
    p.nsplit = NT;
    p.bias = bias_p; p.z = z; p.apad = apad; p.s = s; p.Cn = O / (s * s); p.Nout = O;
    p.z_bytes = (unsigned)((size_t)(H * s) * (W * s) * p.Cn * 2);
    p.apad_bytes = apad ? (unsigned)((size_t)(H * s + 2) * (W * s + 2) * p.Cn * 2) : 0;
    p.mCn = c2_magic(p.Cn); p.mS = c2_magic(s);
    // The head in this epilogue is correct (tests run it with ORN_HEAD_FUSED=1) but does not pay: 96 SiLUs + 288 FMAs per lane on
    // the vector pipe cost the last block 145 -> 190 us, the 42 us HBM-bound head kernel it replaces included -- and the denser
    // launch drags the clock of its neighbours down (720p step 1.168 -> 1.188 ms on one box).  Off unless asked for.
    static const bool fuse_head = orn_probe_env("ORN_HEAD_FUSED") != nullptr;
    if (head && !apad && p.Cn == 96 && fuse_head) {
        p.head_w = head->w; p.head_b = head->b; p.head_out = head->out; p.head_sigmoid = head->sigmoid;
        head->fused = 1;
    }
    const int blocks = orn_cdiv(p.ptiles, 8) * 8 * p.nsplit;
    const size_t lds = C2_LDS + 96 * 4 + (p.head_out ? (3 * 96 + 4) * 4 : 0);
    return apad ? c2_launch<C2_FWD>(p, blocks, lds, st, "fwd2_nhwc") : c2_launch<C2_FWD_LAST>(p, blocks, lds, st, "fwd2_nhwc_last");
}

}  // namespace HNS
