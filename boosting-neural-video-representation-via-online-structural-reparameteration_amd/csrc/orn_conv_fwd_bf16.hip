// A4 fast path, FORWARD: conv3x3 + bias + PixelShuffle (+ SiLU copy for the next block) on 16-bit MFMA
// (v_mfma_f32_32x32x16, fp32 accumulate), model.py:539,567.  Buffer layouts, the LDS-DMA staging, the weight-tile ring that
// runs on across N tiles, the hand-placed fragment reads with counted waits and the counted raw-buffer stores are described in
// orn_conv_bf16.hip, which holds the same machinery for the dgrad on the OTHER MFMA shape: the dgrad gains 4 % from
// v_mfma_f32_16x16x32 (higher clock under load), this kernel loses 12 % with it (measured both ways on the 720p shapes), so
// the two directions keep separate kernels -- different lane maps, LDS swizzles and epilogues -- in separate files.
// Compiled twice like its sibling: as is (bf16, namespace orn_bf16) and with -DORN_FP16 (IEEE half, namespace orn_f16).
#include "orn_internal.h"
#include <type_traits>
#ifdef ORN_FP16
#define HNS orn_f16
typedef _Float16 h16;
#define MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
#define HNS orn_bf16
typedef __bf16 h16;
#define MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void orn_sfor_f(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        orn_sfor_f<I + 1, N>(f);
    }
}

namespace HNS {

// The timing-ablation flags cost registers and branches in the hot loops: they are compiled in only with
// -DORN_CONV_ABLATE (tools/probes builds); product builds see a constant 0.
#ifdef ORN_CONV_ABLATE
#define PDBG(p_) ((p_).dbg)
#else
#define PDBG(p_) 0
#endif
static int g_convf_dbg = 0;   // timing experiments only (tools/probes), see orn_debug_set
// Phase stamps (diagnostic build -DORN_CONV_STAMP; the product build compiles none of it): wave 0 of every work-group writes
// s_memtime at the N-tile phase boundaries into a buffer no other code reads.
#ifdef ORN_CONV_STAMP
static unsigned long long *g_convf_stamps = nullptr;
// stamps collect in 512 B of LDS behind the kernel's own images (a global store per stamp would sit in every vmcnt wait)
#define STAMP_LDS ((unsigned long long *)(smem + PATCH_LDS + NBUF * BS_BYTES + (EPI_IS_FWD(EPI) ? ((((p.Nout + BN - 1) / BN * BN) * 4 + 255) & ~255) : 0)))
#define STAMP(i_) { if (p.stamps && t == 0) STAMP_LDS[i_] = __builtin_amdgcn_s_memtime(); }
#define STAMP_RT(i_) { if (p.stamps && t == 0) STAMP_LDS[i_] = __builtin_amdgcn_s_memrealtime(); }
#define STAMP_FLUSH() { if (p.stamps && t < 128) p.stamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 128 + t] = STAMP_LDS[t]; }
// per-tap stamps of wave 0 (slots 16..) and of the wave that shares its SIMD (slots 64..): up to 4 N tiles / chunks x 9 taps
#define STAMP_TAP(seg_, tap_) { if (p.stamps && (seg_) < 4 && lane == 0 && (wave == 0 || wave == NWAVES / 2)) STAMP_LDS[(wave == 0 ? 16 : 64) + (seg_) * 9 + (tap_)] = __builtin_amdgcn_s_memtime(); }
// rendezvous of taps 3..5 of segment 0: arrival (k 0), after the vmcnt wait (1), after the barrier (2); wave 0 -> slots 100.., partner -> 112..
#define STAMP_BAR(seg_, tap_, k_) { if (p.stamps && (seg_) == 0 && (tap_) >= 3 && (tap_) <= 5 && lane == 0 && (wave == 0 || wave == NWAVES / 2)) STAMP_LDS[(wave == 0 ? 100 : 112) + ((tap_) - 3) * 3 + (k_)] = __builtin_amdgcn_s_memtime(); }
#else
#define STAMP(i_)
#define STAMP_RT(i_)
#define STAMP_FLUSH()
#define STAMP_TAP(seg_, tap_)
#define STAMP_BAR(seg_, tap_, k_)
#endif

#define CB_TH 8
#define CB_TW 32
#define CB_PH (CB_TH + 2)
#define CB_PW (CB_TW + 2)
#define CB_CK 96                 // channels per K chunk
#define CB_PIXB 208              // LDS bytes per patch pixel (192 data + 16 pad: conflict-free b128 reads)
#define CB_PATCH_BYTES (CB_PH * CB_PW * CB_PIXB)
#define CB_ROWB 208              // LDS bytes per weight-tile row

// EPI_B_FWD_LAST: the forward of the last block (no activation copy for a next layer): its own instantiation, so the
// largest launch of the step carries neither the second set of deferred-store registers nor the SiLU code
enum { EPI_B_FWD = 0, EPI_B_DGRAD = 1, EPI_B_DGRAD_F32 = 2, EPI_B_FWD_LAST = 3 };
#define EPI_IS_FWD(e_) ((e_) == EPI_B_FWD || (e_) == EPI_B_FWD_LAST)

typedef __attribute__((ext_vector_type(2))) h16 h16x2;
__device__ __forceinline__ unsigned packf_h16x2(float lo, float hi)
{
    h16x2 v;
    v[0] = (h16)lo;
    v[1] = (h16)hi;
    return __builtin_bit_cast(unsigned, v);
}
// v_permlane32_swap: lanes 32-63 of `a` <-> lanes 0-31 of `b` (guide T21).  After the call lanes < 32
// hold (own a, upper half's a) and lanes >= 32 hold (lower half's b, own b).
__device__ __forceinline__ void swapf_halves(unsigned &a, unsigned &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swapf_halves_f(float &a, float &b)
{
    unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    swapf_halves(ua, ub);
    a = __builtin_bit_cast(float, ua);
    b = __builtin_bit_cast(float, ub);
}

__device__ __forceinline__ int convf_div(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

struct ConvFP {
    const h16 *xpad;     // [H+2][W+2][Cin]
    const h16 *w;        // [9][Nout][Cin]
    const float *bias;   // [Nout] (o' order) or null
    int H, W, Cin, Nout;
    int tiles_w, tiles_h, n_tiles_per_wg;
    int n_full;          // work-groups [0, n_full) own whole pixel tiles; the rest own one N tile each
    int qsplit;          // EPI_B_DGRAD_F32 on small images: blockIdx.y = input chunk, one fp32 partial slab per chunk
    // EPI_B_FWD
    h16 *z;              // [H*s][W*s][Cn]
    h16 *apad;           // [H*s+2][W*s+2][Cn] or null
    int s, Cn;
    unsigned z_bytes, apad_bytes;   // sizes of the two buffers (raw-buffer bounds)
    // EPI_B_DGRAD: out = dx * silu'(zprev) scattered into the previous layer's dypad
    const h16 *zprev;    // [H][W][Nout]
    h16 *dyprev;         // [H/sp+2][W/sp+2][Nout*sp*sp]
    int sp;
    // EPI_B_DGRAD_F32
    float *dx_f32;       // [H][W][Nout]
    // exact division by multiply-high for the epilogues' index math (a runtime integer division costs ~30 instructions,
    // and 16 of them per N tile per lane were a measurable part of the forward kernel): convf_div / convf_magic
    unsigned mCn, mS, mSp;
    int dbg;             // timing-only ablation flags (tools/probes): 1 no weight restage, 2 no patch stage, 4 no stores
    unsigned long long *stamps;   // -DORN_CONV_STAMP diagnostic builds only: 64 time stamps per work-group (tools/probes/conv_stamps.py)
};

// Fragment register sets: reads run CONV_NSET - 1 k-steps ahead of the MFMAs that consume them
#ifndef CONVF_NSET
#define CONVF_NSET 2
#endif
#ifndef CONVF_NSET_UNUSED
#define CONVF_NSET_UNUSED 2
#endif
// Fragment reads of k-step (TAP, KS_) into register set SET: MB patch rows (the MFMA's B operand: pixels) and NB weight
// blocks (A operand: output channels).  a_lane = LDS byte address of this lane's patch pixel for (row wm*MB, tap 0),
// pix_lane = that pixel's index (for the swizzle), b_par0 = this lane's weight-row address for k-step parity 0.
template <int NSET, int MB, int NB, int ROWB, int BS_BYTES, bool ALLTAPS, int SET, int TAP, int KS_>
__device__ __forceinline__ void convf_read_step(h16x8 (&fa)[NSET][MB], h16x8 (&fb)[NSET][NB], unsigned a_lane, unsigned pix_lane, unsigned b_par0, int hh)
{
    constexpr int ti = TAP / 3, tj = TAP - ti * 3, par = KS_ & 1;
    constexpr int buf = ALLTAPS ? TAP : TAP % 3;
    constexpr int kimm = 64 * (KS_ >> 1);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const unsigned pixoff = (i + ti) * CB_PW + tj;
        const unsigned pix = pix_lane + pixoff;
        const unsigned addr = a_lane + pixoff * ROWB + 16 * ((2 * par + hh) ^ ((pix >> 2) & 3));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[SET][i]) : "v"(addr), "n"(kimm) : "memory");
    }
    const unsigned baddr = (b_par0 ^ (32 * par)) + (buf >= 4 ? 4 * BS_BYTES : 0);
    constexpr int bimm = (buf >= 4 ? buf - 4 : buf) * BS_BYTES + kimm;
    static_assert(bimm + (NB - 1) * 32 * ROWB < 65536, "ds_read offset field");
    static_assert(NB <= 3, "convf_read_step: add the fourth weight block");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][0]) : "v"(baddr), "n"(bimm) : "memory");
    if constexpr (NB > 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][NB > 1 ? 1 : 0]) : "v"(baddr), "n"(bimm + 32 * ROWB) : "memory");
    if constexpr (NB > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][NB > 2 ? 2 : 0]) : "v"(baddr), "n"(bimm + 64 * ROWB) : "memory");
}

// The wait that retires register set SET (its reads were issued before the PEND newest ones) names every register of the
// set as read-write, so no MFMA that consumes them can be scheduled above it.
template <int NSET, int MB, int NB, int SET, int PEND>
__device__ __forceinline__ void convf_wait_set(h16x8 (&fa)[NSET][MB], h16x8 (&fb)[NSET][NB])
{
    if constexpr (MB == 2 && NB == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[SET][0]), "+v"(fa[SET][MB > 1 ? 1 : 0]), "+v"(fb[SET][0]), "+v"(fb[SET][NB > 1 ? 1 : 0]) : "n"(PEND));
    else if constexpr (MB == 1 && NB == 3)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[SET][0]), "+v"(fb[SET][0]), "+v"(fb[SET][NB > 1 ? 1 : 0]), "+v"(fb[SET][NB > 2 ? 2 : 0]) : "n"(PEND));
    else if constexpr (MB == 1 && NB == 1)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[SET][0]), "+v"(fb[SET][0]) : "n"(PEND));
    else
        static_assert(MB == 2 && NB == 2, "convf_wait_set: add this wave tile");
}

// CK = input channels per K chunk: 96 (the general form above), or 32 for a layer whose input has <= 32 real channels
// (the zero-padded narrow layer, forward only): its whole K = 9 x 32 fits LDS -- patch 22 KB + all nine [BN][32] weight
// tiles 72 KB -- so an N tile is ONE rendezvous and 72 back-to-back MFMAs per wave instead of nine rounds of barrier +
// counted wait + 24 MFMAs of which two thirds multiply zeros.  Rows are 64 B: 4 chunks, XOR swizzle (chunk ^ ((row >> 2) & 3)).
// ALLTAPS: all nine weight tiles of the (single) K chunk resident, one rendezvous per N tile -- the narrow form, and the
// chunk-split dgrad of a layer with <= 32 real OUTPUT channels (N tile 32: 9 x 6 KB next to the 64 KB patch).
template <int WAVES_M, int WAVES_N, int MB, int NB, int EPI, int CK = CB_CK, bool ALLTAPS = (CK != CB_CK)>
__global__ void __launch_bounds__(WAVES_M *WAVES_N * 64) k_conv_fwd_nhwc_bf16(ConvFP p)
{
    ORN_PRIO_HIGH();
    static_assert(EPI_IS_FWD(EPI), "this file holds the forward kernel only (dgrad: orn_conv_bf16.hip)");
    constexpr bool NARROW = (CK != CB_CK);
    static_assert(CK == CB_CK || (CK == 32 && EPI_IS_FWD(EPI)), "narrow form: 32 channels, forward only");
    constexpr int NCH = CK / 8;                        // 16-byte chunks per LDS row
    static_assert(!NARROW || ALLTAPS, "the narrow form keeps all taps resident");
    constexpr int NBUF = ALLTAPS ? 9 : 3;              // weight tiles resident at once
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BN = WAVES_N * NB * 32;
    static_assert(WAVES_M * MB == CB_TH, "M tile must be 8 rows of 32 pixels");
    // LDS images: UNPADDED 192-byte rows (12 x 16-byte chunks) filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
    // wave-instruction, lane-linear destination, no VGPRs, no ds_write).  Conflict-free ds_read_b128 comes from a
    // rotation swizzle -- logical chunk c of row R sits at position (c + ((R >> 2) & 3)) % 12 -- applied on the DMA's
    // per-lane SOURCE address and on the fragment reads (both sides or neither: guide rule 21).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int ROWB = CK * 2;
    constexpr int PATCH_INSTR = (CB_PH * CB_PW * ROWB + 1023) / 1024;   // 340 pixels x 192 B = 65,280 -> 64 wave-instructions
    constexpr int PATCH_LDS = PATCH_INSTR * 1024;
    constexpr int BS_BYTES = BN * ROWB;
    constexpr int B_INSTR = BS_BYTES / 1024;           // wave-instructions per weight tile
    // Weight tiles are fetched by the FIRST HALF of the waves only (one per SIMD: waves w and w + NWAVES/2 share one): an
    // LDS-DMA instruction parks its wave for ~100 cycles, and when both waves of a SIMD issue theirs right after the
    // rendezvous the matrix pipe idles for all of them (~300 cycles per tap, measured with phase stamps); with one loader
    // per SIMD its partner's MFMAs run meanwhile, and the loader catches up while the partner waits at the next rendezvous.
#ifdef ORN_DGRAD_ALL_LOAD
    constexpr int NLOAD = (ALLTAPS || NWAVES < 8 || !EPI_IS_FWD(EPI)) ? NWAVES : NWAVES / 2;
#else
    constexpr int NLOAD = (ALLTAPS || NWAVES < 8) ? NWAVES : NWAVES / 2;
#endif
    constexpr int B_PER_WAVE = (B_INSTR + NLOAD - 1) / NLOAD;
    constexpr int P_PER_WAVE = (PATCH_INSTR + NWAVES - 1) / NWAVES;
    static_assert((NARROW || PATCH_INSTR % NWAVES == 0) && BS_BYTES % 1024 == 0, "tile geometry");
    unsigned char *patch = smem;
    unsigned char *bs0 = smem + PATCH_LDS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // work list: the first n_full work-groups take a pixel tile with all its N tiles; the pixel tiles of the
    // last partial round are cut into single-N-tile work-groups so the tail spreads over every CU
    int tile = blockIdx.x, nt0 = blockIdx.y * p.n_tiles_per_wg, nt_cnt = p.n_tiles_per_wg;
    if ((int)blockIdx.x >= p.n_full) {
        const int r = blockIdx.x - p.n_full;
        tile = p.n_full + r / p.n_tiles_per_wg;
        nt0 = r - (r / p.n_tiles_per_wg) * p.n_tiles_per_wg;
        nt_cnt = 1;
    }
    const int tw = tile % p.tiles_w, th = tile / p.tiles_w;
    const int h0 = th * CB_TH, w0 = tw * CB_TW;
    const int H = p.H, W = p.W, Cin = p.Cin;
    const int q_base = (EPI == EPI_B_DGRAD_F32 && p.qsplit) ? (int)blockIdx.y : 0;   // chunk split: this WG's chunk
    const int Q = (ALLTAPS || (EPI == EPI_B_DGRAD_F32 && p.qsplit)) ? 1 : Cin / CB_CK;   // ALLTAPS dgrad: launched chunk-split
    if (EPI == EPI_B_DGRAD_F32 && p.qsplit) nt0 = 0;
    const int n_tiles = Q * 9;                         // weight tiles per N tile

    const int uwave = __builtin_amdgcn_readfirstlane(wave);        // provably wave-uniform (LDS-DMA base -> M0)
    // per-lane SOURCE offsets (elements) of this wave's DMA instructions; rot() un-swizzles position -> logical chunk
    int b_goff[B_PER_WAVE], p_goff[P_PER_WAVE];
    bool p_ok[P_PER_WAVE];
#pragma unroll
    for (int k = 0; k < B_PER_WAVE; ++k) {
        const int m = (uwave + NLOAD * k) % B_INSTR;               // surplus instructions re-load a tile piece (harmless)
        const int L = m * 64 + lane, R = L / NCH, pos = L - R * NCH;
        const int c = pos ^ ((R >> 2) & 3);
        b_goff[k] = R * Cin + c * 8;
    }
#pragma unroll
    for (int k = 0; k < P_PER_WAVE; ++k) {
        const int m = uwave + NWAVES * k;
        const int L = m * 64 + lane, pix = L / NCH, pos = L - pix * NCH;
        const int c = pos ^ ((pix >> 2) & 3);
        const int pr = pix / CB_PW, pc = pix - pr * CB_PW;
        const int gh = h0 + pr, gw_ = w0 + pc;
        p_ok[k] = (pix < CB_PH * CB_PW) && gh < H + 2 && gw_ < W + 2;
        // out-of-image pixels read the (0,0) border pixel, which is all zeros
        p_goff[k] = p_ok[k] ? ((gh * (W + 2) + gw_) * Cin + c * 8) : c * 8;
    }
#define DMA16(gptr_, ldsoff_)                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr_),                   \
                                     (__attribute__((address_space(3))) void *)(smem + (ldsoff_)), 16, 0, 0)
#define DMA_B(buf_, nt_, q_, tap_)                                                                              \
    {                                                                                                           \
        if (NLOAD == NWAVES || uwave < NLOAD) {                                                                 \
            const h16 *wbase = p.w + ((size_t)((tap_) * p.Nout + (nt_) * BN) * Cin + ((q_) + q_base) * CK);     \
            _Pragma("unroll") for (int k = 0; k < B_PER_WAVE; ++k)                                              \
                DMA16(wbase + b_goff[k], PATCH_LDS + (buf_) * BS_BYTES + ((uwave + NLOAD * k) % B_INSTR) * 1024); \
        }                                                                                                       \
    }
#define DMA_PATCH(q_)                                                                                           \
    {                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < P_PER_WAVE; ++k)                                                  \
            if (!NARROW || uwave + NWAVES * k < PATCH_INSTR)                                                    \
                DMA16(p.xpad + p_goff[k] + (p_ok[k] ? ((q_) + q_base) * CK : 0), (uwave + NWAVES * k) * 1024);  \
    }
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define WAIT_VMC(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define BARRIER() __builtin_amdgcn_s_barrier()
    // Fragment reads are hand-placed (inline asm: hipcc sinks every builtin LDS read next to its consumer and waits
    // lgkmcnt(0) right behind it, which exposed one LDS round trip per k-step).  16 bytes at logical chunk c = 2*ks + hh of
    // row R sit at position c ^ ((R >> 2) & 3): byte offset 64*(ks >> 1) [an immediate] + 16*((2*(ks & 1) + hh) ^ rot) [two
    // per-lane values, one per k-step parity, 32 apart by XOR].  All addresses are LDS byte offsets in a VGPR.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned b_rot = (l31 >> 2) & 3;
    const unsigned b_par0 = lds0 + PATCH_LDS + (wn * NB * 32 + l31) * ROWB + 16 * (hh ^ b_rot);   // weight rows, parity 0
    const unsigned a_lane = lds0 + (wm * MB * CB_PW + l31) * ROWB;                               // patch pixel of (row wm*MB, tap 0)

    // EPI_B_FWD*: the packed outputs of an N tile are stored by its epilogue as RAW BUFFER stores that every lane issues
    // (out-of-image lanes carry an out-of-range offset and are dropped by the bounds check): the number of vector-memory
    // operations behind the last DMA is then known, and the next rendezvous' wait steps over them (vmcnt(n) = all but the n
    // newest) instead of stalling on HBM write latency.
    constexpr bool APAD = (EPI == EPI_B_FWD);          // also writes a = SiLU(z) into the next layer's padded input
    constexpr int NST = EPI_IS_FWD(EPI) ? MB * NB * 2 * (APAD ? 2 : 1) : 0;   // stores per wave and N tile
    bool pending = false;                              // epilogue stores were issued after this wave's last DMA wait
    // z / apad as raw buffers: byte offsets; 0x80000000 (out of range for any buffer the launcher admits) drops the lane's store
    const auto z_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.z, 0, EPI_IS_FWD(EPI) ? p.z_bytes : 0, 0x00020000);
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.apad, 0, APAD ? p.apad_bytes : 0, 0x00020000);
    float *sbias = reinterpret_cast<float *>(smem + PATCH_LDS + NBUF * BS_BYTES);     // [Nout] after the weight ring (EPI_B_FWD)
    if (EPI_IS_FWD(EPI))
        for (int i = t; i < (p.Nout + BN - 1) / BN * BN; i += NT)                    // visible after the first N tile's barriers
            sbias[i] = (p.bias && i < p.Nout) ? p.bias[i] : 0.f;                    // (zeros behind Nout: a ragged last N tile)
#ifdef ORN_CONV_PRIO
    if (NLOAD != NWAVES && uwave >= NLOAD) __builtin_amdgcn_s_setprio(1);   // experiment: static priority for the non-loader half
#endif
    STAMP_RT(0)
    constexpr int NSET = EPI_IS_FWD(EPI) ? CONVF_NSET : CONVF_NSET_UNUSED, LEAD = NSET - 1;   // reads run LEAD k-steps ahead of their MFMAs
    h16x8 fa[NSET][MB], fb[NSET][NB];                   // fragment register sets (carried across N tiles by the pipeline)
    for (int nti = 0; nti < nt_cnt; ++nti) {
        const int nt = nt0 + nti;
        STAMP(2 + nti * 4)
        // acc[i][j]: D rows = 32 output channels (A operand = weights), D cols = 32 pixels of one row
        // (B operand = input patch): each lane owns ONE pixel and 16 channels in groups of 4 consecutive.
        f32x16 acc[MB][NB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // prologue (every N tile of the all-taps-resident forms; otherwise once per work-group: the weight-tile ring then
        // runs on ACROSS N tiles -- the last three taps of an N tile fetch the first three tiles of the next one, so an N
        // tile boundary costs a rendezvous, not a drained pipeline): the patch (chunk 0) and weight tiles 0, 1; tile 2 stays
        // in flight behind the first rendezvous.
        const bool has_next_nt = (nti + 1 < nt_cnt);
        if (ALLTAPS || nti == 0) {
            BARRIER();
            if ((nti == 0 || Q > 1) && !(PDBG(p) & 2)) DMA_PATCH(0)
            DMA_B(0, nt, 0, 0)
            if (n_tiles > 1) DMA_B(1, nt, 0, 1)
            if (ALLTAPS) {                             // the whole K of this N tile: taps 2..8 too, then the only rendezvous
#pragma unroll
                for (int tp = 2; tp < 9; ++tp) DMA_B(tp, nt, 0, tp)
            }
            WAIT_VM(0);
            BARRIER();
            if (!ALLTAPS && n_tiles > 2) DMA_B(2, nt, 0, 2)
        }
        // One software pipeline over all k-steps (chunks of 96 input channels outside; nine taps x CK/16 k-slices inside,
        // unrolled at compile time so tap, kernel row / column, ring slot, register set and LDS offsets are constants): step s
        // issues the fragment reads of step s+LEAD into another register set, waits with a COUNTED lgkmcnt for its own
        // (issued LEAD steps earlier), then runs its MFMAs -- also across a tap boundary, so the rendezvous at the end of a tap
        // sits between MFMAs whose operands are already in registers or in flight.  (LEAD = 2 measured the same as 1 on the
        // 720p shapes: CONV_NSET_* keep the knob.)  Reading tile t+1 before rendezvous t is legal because every wave waits for
        // ALL its outstanding DMA pieces (tile t+2 included) before rendezvous t: tile t+1 was complete, and known to be, at
        // rendezvous t-1.  Ring: after rendezvous t the DMA of tile t+3 overwrites tile t.
        constexpr int KS = CK / 16, NR = MB + NB, NSTEP = 9 * KS;
        static_assert(NSTEP % NSET == 0 && KS >= LEAD, "the register-set rotation must repeat per chunk");
#define READ_STEP(set_, tap_, ks_) convf_read_step<NSET, MB, NB, ROWB, BS_BYTES, ALLTAPS, set_, tap_, ks_>(fa, fb, a_lane, wm * MB * CB_PW + l31, b_par0, hh)
        STAMP(3 + nti * 4)
        if (ALLTAPS || nti == 0) { READ_STEP(0, 0, 0); if constexpr (LEAD > 1) READ_STEP(1 % NSET, 0, 1); }   // later N tiles: issued by the previous N tile's last steps
        for (int q = 0; q < Q; ++q) {
            const bool last_chunk = (q + 1 >= Q);
            const bool more_segs = !last_chunk || has_next_nt;             // another (N tile, chunk) segment follows in the stream
            const int qn = last_chunk ? 0 : q + 1, ntn = last_chunk ? nt + 1 : nt;
            const bool carry = !ALLTAPS && (Q == 1) && has_next_nt;        // same patch next: the pipeline runs on into the next N tile
            orn_sfor_f<0, 9>([&](auto tap_c) __attribute__((always_inline)) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int buf = ALLTAPS ? tap : tap % 3;
                orn_sfor_f<0, KS>([&](auto ks_c) __attribute__((always_inline)) {
                    constexpr int ks = decltype(ks_c)::value;
                    constexpr int g = tap * KS + ks, cur = g % NSET, nxt = (g + LEAD) % NSET;
                    constexpr int g2 = g + LEAD;                           // the step whose reads are issued now
                    if constexpr (g2 < NSTEP) {
                        READ_STEP(nxt, g2 / KS, g2 % KS);
                        convf_wait_set<NSET, MB, NB, cur, LEAD * NR>(fa, fb);
                    } else if (carry) {                                    // first steps of the next N tile
                        READ_STEP(nxt, 0, g2 - NSTEP);
                        convf_wait_set<NSET, MB, NB, cur, LEAD * NR>(fa, fb);
                    } else
                        convf_wait_set<NSET, MB, NB, cur, (NSTEP - 1 - g) * NR>(fa, fb);
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) acc[i][j] = MFMA_H16(fb[cur][j], fa[cur][i], acc[i][j]);
                });
                STAMP_TAP(nti * Q + q, tap)
                if (!ALLTAPS && ((tap < 8) || more_segs)) {
                    // stream per wave: .. DMA (tap 8) [epilogue: NST stores] | tap 0: wait for that DMA only, DMA | tap 1: wait all ..
                    if constexpr (EPI_IS_FWD(EPI) && tap == 0) {
                        if (pending) WAIT_VMC(NST); else WAIT_VM(0);
                        pending = false;
                    } else {
                        STAMP_BAR(nti * Q + q, tap, 0)
                        WAIT_VM(0);                 // this wave's pieces of every tile in flight (tile tt+2) have landed
                        STAMP_BAR(nti * Q + q, tap, 1)
                    }
                    if (!(PDBG(p) & 8)) BARRIER();
                    STAMP_BAR(nti * Q + q, tap, 2)
                    if constexpr (tap == 8) {
                        if (Q > 1) {                // next chunk: everyone is done with the old chunk's patch
                            if (!(PDBG(p) & 2)) DMA_PATCH(qn)
                            WAIT_VM(0);
                            BARRIER();
                        }
                    }
                    if (!(PDBG(p) & 1)) {           // tile tt + 3 into the buffer of tile tt (free now)
                        if constexpr (tap < 6) DMA_B(buf, nt, q, tap + 3)
                        else if (more_segs) DMA_B(buf, ntn, qn, tap - 6)
                    }
                    if constexpr (tap == 8) {
                        if (Q > 1) { READ_STEP(0, 0, 0); if constexpr (LEAD > 1) READ_STEP(1 % NSET, 0, 1); }
                    }
                }
            });
        }
        // a carried-over prefetch lands before the epilogue's code runs (the compiler may move those registers there)
        if (!ALLTAPS && (Q == 1) && has_next_nt) { convf_wait_set<NSET, MB, NB, 0, (LEAD - 1) * NR>(fa, fb); if constexpr (LEAD > 1) convf_wait_set<NSET, MB, NB, 1 % NSET, 0>(fa, fb); }
#undef READ_STEP
        STAMP(4 + nti * 4)

        // ---- epilogue --------------------------------------------------------------------------
        // Lane (pixel l31, half hh) holds channels 8g + 4hh + e (g = reg>>2, e = reg&3) of each 32-ch block.
        // v_permlane32_swap pairs the two half-waves so that every lane ends up with 8 CONSECUTIVE
        // channels of its pixel (lanes <32: group pair's first 8, lanes >=32: the next 8): 16-byte stores.
        const int gw = w0 + l31;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int gh = h0 + wm * MB + i;
            const bool ok = (gh < H) && (gw < W) && !(PDBG(p) & 4);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int cb = nt * BN + (wn * NB + j) * 32;            // first output channel of the block
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const int c8 = cb + 8 * (k + hh);                   // the 8 channels this lane stores
                    if (EPI_IS_FWD(EPI)) {
                        float va[4], vb[4];
                        // bias from its LDS copy (a global load here would expose its latency once per N tile)
                        const float4 ba = *reinterpret_cast<const float4 *>(sbias + cb + 8 * k + 4 * hh);
                        const float4 bb = *reinterpret_cast<const float4 *>(sbias + cb + 8 * (k + 1) + 4 * hh);
                        va[0] = acc[i][j][4 * k + 0] + ba.x; va[1] = acc[i][j][4 * k + 1] + ba.y;
                        va[2] = acc[i][j][4 * k + 2] + ba.z; va[3] = acc[i][j][4 * k + 3] + ba.w;
                        vb[0] = acc[i][j][4 * k + 4] + bb.x; vb[1] = acc[i][j][4 * k + 5] + bb.y;
                        vb[2] = acc[i][j][4 * k + 6] + bb.z; vb[3] = acc[i][j][4 * k + 7] + bb.w;
                        unsigned za0 = packf_h16x2(va[0], va[1]), za1 = packf_h16x2(va[2], va[3]);
                        unsigned zb0 = packf_h16x2(vb[0], vb[1]), zb1 = packf_h16x2(vb[2], vb[3]);
                        swapf_halves(za0, zb0); swapf_halves(za1, zb1);
                        const int ij = convf_div(c8, p.mCn), n = c8 - ij * p.Cn;
                        const int si = convf_div(ij, p.mS), sj = ij - si * p.s;
                        const int Ws = W * p.s, oh = gh * p.s + si, ow = gw * p.s + sj;
                        // (a ragged last N tile -- Nout not a multiple of the N tile -- computes its missing 32-channel blocks
                        // on whatever weight rows follow in memory and drops them here)
                        const bool okc = ok && c8 < p.Nout;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{za0, za1, zb0, zb1}, z_rsrc,
                                                               okc ? ((oh * Ws + ow) * p.Cn + n) * 2 : (int)0x80000000, 0, 0);   // < 2^31 bytes: launcher
                        if (APAD) {
                            unsigned aa0 = packf_h16x2(orn_silu(va[0]), orn_silu(va[1])), aa1 = packf_h16x2(orn_silu(va[2]), orn_silu(va[3]));
                            unsigned ab0 = packf_h16x2(orn_silu(vb[0]), orn_silu(vb[1])), ab1 = packf_h16x2(orn_silu(vb[2]), orn_silu(vb[3]));
                            swapf_halves(aa0, ab0); swapf_halves(aa1, ab1);
                            // the activation copy leaves right away (the next vmcnt wait is a whole tap of the next N tile away)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{aa0, aa1, ab0, ab1}, a_rsrc,
                                                                   okc ? (((oh + 1) * (Ws + 2) + (ow + 1)) * p.Cn + n) * 2 : (int)0x80000000, 0, 0);
                        }
                    }
                }
            }
        }
        if (EPI_IS_FWD(EPI)) pending = true;
        STAMP(5 + nti * 4)
    }
    STAMP(2 + nt_cnt * 4)
    STAMP_RT(1)
    STAMP_FLUSH()
}

#undef DMA16
#undef DMA_B
#undef DMA_PATCH
#undef WAIT_VM
#undef BARRIER

template <int WAVES_M, int WAVES_N, int MB, int NB, int EPI, int CK = CB_CK, bool ALLTAPS = (CK != CB_CK)>
static int launch_convf_cfg(const ConvFP &p, int n_tiles_total, hipStream_t st)
{
    constexpr int BN = WAVES_N * NB * 32;
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr size_t LDS_IMG = (size_t)(CB_PH * CB_PW * CK * 2 + 1023) / 1024 * 1024 + (ALLTAPS ? 9 : 3) * (size_t)BN * CK * 2;
    size_t smem = LDS_IMG + (EPI_IS_FWD(EPI) ? orn_align((size_t)orn_cdiv(p.Nout, BN) * BN * 4) : 0);   // + bias copy (whole N tiles)
#ifdef ORN_CONV_STAMP
    smem += 1024;
#endif
    auto kern = k_conv_fwd_nhwc_bf16<WAVES_M, WAVES_N, MB, NB, EPI, CK, ALLTAPS>;
    static bool attr_done = false;
    if (!attr_done) {
        // opt in once for the largest request (bias copy up to 2048 channels)
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(LDS_IMG + 8192));
        if (e != hipSuccess) { orn_set_error("conv_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    const int ptiles = p.tiles_w * p.tiles_h;
    ConvFP q = p;
    dim3 grid(ptiles, n_tiles_total / p.n_tiles_per_wg);
    if (p.qsplit) grid.y = p.Cin / CB_CK;
    q.n_full = ptiles;
    if (p.n_tiles_per_wg > 1 && p.n_tiles_per_wg == n_tiles_total && ptiles > 256) {
        // 256 CUs, one work-group each: whole rounds keep full tiles, the last partial round is cut up
        q.n_full = ptiles / 256 * 256;
        grid = dim3(q.n_full + (ptiles - q.n_full) * p.n_tiles_per_wg, 1);
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), smem, st, q);
    ORN_LAUNCH_CHECK("conv_nhwc_bf16");
    return 0;
}

void set_debug_fwd(int flags) { g_convf_dbg = flags; }

// m with x / d == umulhi(x, m) for every 0 <= x < 2^16 and 2 <= d < 2^16 (m = ceil(2^32 / d): the error term
// x * (m*d - 2^32) < 2^16 * 2^16); d == 1 is encoded as m = 0 (convf_div returns x)
static unsigned convf_magic(int d)
{
    return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned long long)d - 1) / (unsigned long long)d);
}

int orn_launch_fwd2(const h16 *xpad, const h16 *wb, const float *bias_p, int H, int W, int O, int s, h16 *z, h16 *apad, hipStream_t st, OrnHeadFuse *head);   // orn_conv2_bf16.hip

// fwd: N tile 128 (waves 4x2, wave tile 64 px x 64 ch); dgrad: N = 96 in one tile (waves 8x1, 32 px x 96 ch)
// c_real: input channels that are not zero padding (<= Cin); <= 32 of them take the narrow form (forward of a non-last block)
int orn_launch_conv_bf16_fwd(const h16 *xpad, const h16 *wb, const float *bias_p, int H, int W, int Cin, int O, int s,
                             h16 *z, h16 *apad, hipStream_t st, int c_real, OrnHeadFuse *head)
{
    // O % 32: whole MFMA blocks; an O that is not a multiple of the 128-channel N tile gets a ragged last tile whose weight
    // DMA reads up to 96 rows past row O of each tap: `wb` must be readable for 96 * Cin elements behind its last row
    // (orn_conv_bf16_wb_elems; the values are never used)
    ORN_REQUIRE(Cin % CB_CK == 0 && O % 32 == 0 && O % (s * s) == 0, "conv_bf16_fwd: unsupported Cin=%d O=%d s=%d", Cin, O, s);
    ConvFP p = {};
    p.dbg = g_convf_dbg;
#ifdef ORN_CONV_STAMP
    p.stamps = g_convf_stamps;
#endif
    p.xpad = xpad; p.w = wb; p.bias = bias_p; p.H = H; p.W = W; p.Cin = Cin; p.Nout = O;
    p.tiles_w = orn_cdiv(W, CB_TW); p.tiles_h = orn_cdiv(H, CB_TH);
    p.z = z; p.apad = apad; p.s = s; p.Cn = O / (s * s);
    ORN_REQUIRE(O <= 2048 && s < 65536 && (long)(H * s + 2) * (W * s + 2) * p.Cn < 1073741824L, "conv_bf16_fwd: sizes exceed the 32-bit index math");
    p.z_bytes = (unsigned)((size_t)(H * s) * (W * s) * p.Cn * 2);
    p.apad_bytes = apad ? (unsigned)((size_t)(H * s + 2) * (W * s + 2) * p.Cn * 2) : 0;
    p.mCn = convf_magic(p.Cn); p.mS = convf_magic(s);
    const int nt_total = orn_cdiv(O, 128);
    // One work-group per CU (LDS): keep a pixel tile's N tiles together (patch staged once) unless cutting them
    // apart fills the chip better.  Cost model in units of one N tile: rounds x (work + ~0.3 for the patch).
    const int ptiles = p.tiles_w * p.tiles_h;
    const float cost_whole = (float)orn_cdiv(ptiles, 256) * nt_total;
    const float cost_split = (float)orn_cdiv(ptiles * nt_total, 256) * 1.3f;
    p.n_tiles_per_wg = (ptiles >= 512 || cost_whole <= cost_split) ? nt_total : 1;
    if (apad && c_real > 0 && c_real <= 32) return launch_convf_cfg<4, 2, 2, 2, EPI_B_FWD, 32>(p, nt_total, st);
    // large images with whole 96-channel N tiles: the two-work-groups-per-CU form (orn_conv2_bf16.hip).  Measured in the step:
    // the last block (z only) 150 -> 140 us at 720p; blocks that also write the activation copy are neutral at 230 pixel tiles
    // (720p, 180 x 320: 74.2 vs 73.9 us for the two such launches) and gain from ~500 tiles on (1080p: 181 -> 168 us for its
    // three), so those take it from 400 tiles (ORN_FWD2_APAD: always).
    static const bool form1 = orn_probe_env("ORN_FWD_FORM1") != nullptr;         // tools/probes: A/B against this file's kernel
    static const bool form2_apad = orn_probe_env("ORN_FWD2_APAD") != nullptr;
    static const int min_tiles = orn_probe_env_int("ORN_FWD2_MINTILES", 128);
    if (!form1 && Cin == 96 && O % 96 == 0 && ptiles >= min_tiles && (!apad || ptiles >= 400 || form2_apad)) {
        const int rc = orn_launch_fwd2(xpad, wb, bias_p, H, W, O, s, z, apad, st, head);
        if (rc != -1) return rc;
    }
    return apad ? launch_convf_cfg<4, 2, 2, 2, EPI_B_FWD>(p, nt_total, st) : launch_convf_cfg<4, 2, 2, 2, EPI_B_FWD_LAST>(p, nt_total, st);
}


#ifdef ORN_CONV_STAMP
void set_stamps_fwd(void *buf) { g_convf_stamps = (unsigned long long *)buf; }
#endif

}  // namespace HNS
