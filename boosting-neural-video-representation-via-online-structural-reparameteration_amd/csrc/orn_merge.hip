// A3  ERB online re-parameterisation (model.py:450-516) and its backward.
//
// Both the forward and the backward are chains of small fp32 contractions over weight-shaped
// operands (a few MB, L2/MALL resident).  They run on one generic strided/batched fp32 GEMM whose
// every output element is a SINGLE k-ordered fmaf chain starting from +0.0f -- the summation order
// oracle/merge_ref.c specifies -- so the forward is bit-exact against the CPU oracle.
#include "orn_internal.h"
#include "orn_merge_pack.h"
#include <cstdlib>
#include <type_traits>

struct GemmP {
    const float *A, *B;
    float *C;
    int M, N, K;
    long sam, sak;   // A(m,k) = A[m*sam + k*sak]
    long sbk, sbn;   // B(k,n) = B[k*sbk + n*sbn]
    long scm, scn;   // C(m,n) = C[m*scm + n*scn]
    long ba, bb, bc; // per-batch element offsets (blockIdx.z)
    int a_kfast, b_nfast;
    int a_vec, b_vec;    // gemm_body2: 16-byte global loads of A along k / of B along n are possible (strides, sizes and bases aligned)
    // epilogue 1 (merge combine): C = (w3x3 + (P(w1x3) + P(w3x1))) + acc, n = c*9 + ij
    int epi;
    const float *w3x3, *w1x3, *w3x1;
    int Cch;
    // epilogue 1, optional (engine): the first column of tiles also merges the bias, bf = b3x3 + (b1x3 + b3x1)
    const float *b3x3, *b1x3, *b3x1;
    float *bf;
    // 16-bit MFMA variant (merge backward in the 16-bit engine modes): operands are multiplied by sa / sb when they
    // are rounded to IEEE half (gradient operands ~1e-6 would be subnormal), the result by so = 1/(sa*sb)
    float sa, sb, so;
    // epilogue 1, optional (16-bit engine modes): the merged kernel also leaves in the two 16-bit operand layouts of the conv
    // kernels (what k_prep_weights_bf16_all would otherwise re-read Wf for): wb [9][O'][Cp], wd [9][Cp][O'] with flipped
    // taps, o' = (o % s2) * Cn + o / s2; the bias column writes biasp [O'].  half_kind: 1 = bf16, 2 = IEEE half.
    void *wb, *wd;
    float *biasp;
    int half_kind, s2, Cn, Cp;
};

typedef __attribute__((ext_vector_type(16))) float f32x16;

// 64x64 output tile per work-group, 4 waves (2x2), each wave one 32x32 v_mfma_f32_32x32x2_f32 block.
// K is consumed in ascending order, 2 per instruction (lane half 0 = even k first, then odd k), the
// 64-deep chunks in ascending order, always into the same accumulator: one k-ordered fmaf chain per
// output element, bit-identical to oracle/merge_ref.c.  Out-of-range K is zero-filled
// (fma(0, 0, acc) == acc).  Next chunk's global loads are issued before the MFMAs of the current one.
#define GT 64
#define GK 64
#define GLDA (GK + 1)
#define GLDB (GT + 4)

__device__ __forceinline__ void gemm_body(const GemmP &p, int bx, int by, int bz)
{
    __shared__ float As[GT][GLDA];
    __shared__ float Bs[GK][GLDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * GT, n0 = bx * GT;
    const float *A = p.A + (long)bz * p.ba;
    const float *B = p.B + (long)bz * p.bb;
    float *C = p.C + (long)bz * p.bc;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // Per-thread element coordinates inside a 64x64 tile (16 A + 16 B elements): one coordinate is the thread's own
    // (t & 63, along the operand's contiguous dimension), the other steps by 4 with i.  Loads are UNCONDITIONAL
    // (exec-mask branches around 32 loads per chunk cost more than the chunk's MFMAs): edge rows / columns are clamped
    // onto the last valid one (loaded for nothing, never stored), the K tail is clamped for the address and replaced
    // by +0.0f (fma(0, 0, acc) == acc keeps the chain exact).  24-bit multiplies: operands are weight-sized.
    constexpr int NE = GT * GK / 256;
    float ra[NE], rb[NE];
    const int tf = t & 63, tv = t >> 6;
    const int M = p.M, N = p.N, K = p.K;
    const int sam = (int)p.sam, sak = (int)p.sak, sbk = (int)p.sbk, sbn = (int)p.sbn;
    const bool akf = p.a_kfast, bnf = p.b_nfast;
    auto gload1 = [&](int k0, int i) {
        {
            const int v = tv + 4 * i;
            const int am = akf ? v : tf, ak = k0 + (akf ? tf : v);
            const int bn = bnf ? tf : v, bk = k0 + (bnf ? v : tf);
            const float va = A[__mul24(min(m0 + am, M - 1), sam) + __mul24(min(ak, K - 1), sak)];
            const float vb = B[__mul24(min(bk, K - 1), sbk) + __mul24(min(n0 + bn, N - 1), sbn)];
            ra[i] = va;                         // the K-tail zeroing happens at the LDS store: consuming the value here
            rb[i] = vb;                         // would put the load wait in front of this chunk's MFMAs
        }
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) gload1(k0, i);
    };
    gload(0);
    for (int k0 = 0; k0 < p.K; k0 += GK) {
        __syncthreads();                       // everyone is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int v = tv + 4 * i;
            As[akf ? v : tf][akf ? tf : v] = (k0 + (akf ? tf : v) < K) ? ra[i] : 0.f;
            Bs[bnf ? v : tf][bnf ? tf : v] = (k0 + (bnf ? v : tf) < K) ? rb[i] : 0.f;
        }
        __syncthreads();
        if (k0 + GK < p.K) gload(k0 + GK);     // next chunk's loads fly under this chunk's 32 MFMAs
        const float *ap = &As[wm * 32 + l31][hh];
        const float *bp = &Bs[hh][wn * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk * GLDB], acc, 0, 0, 0);
    }
    const int gn = n0 + wn * 32 + l31;
    if (p.epi == 1 && p.bf && bx == 0 && wn == 0 && l31 == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (gm < p.M) {
                const float b = p.b3x3[gm] + (p.b1x3[gm] + p.b3x1[gm]);     // model.py:476,496
                p.bf[gm] = b;
                if (p.half_kind) p.biasp[(gm % p.s2) * p.Cn + gm / p.s2] = b;
            }
        }
    }
    if (gn >= p.N) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (gm >= p.M) continue;
        float r = acc[reg];
        if (p.epi == 1) {
            const int c = gn / 9, ij = gn % 9, ii = ij / 3, jj = ij % 3;
            const long oc = (long)gm * p.Cch + c;
            const float p13 = (ii == 1) ? p.w1x3[oc * 3 + jj] : 0.f;
            const float p31 = (jj == 1) ? p.w3x1[oc * 3 + ii] : 0.f;
            r = (p.w3x3[(long)gm * p.N + gn] + (p13 + p31)) + r;   // association of model.py:475,495
            if (p.half_kind) {
                const int op = (gm % p.s2) * p.Cn + gm / p.s2;
                const size_t ib = ((size_t)ij * p.M + op) * p.Cp + c, id = ((size_t)(8 - ij) * p.Cp + c) * p.M + op;
                if (p.half_kind == 1) {
                    const __bf16 h = (__bf16)r;
                    reinterpret_cast<__bf16 *>(p.wb)[ib] = h;
                    reinterpret_cast<__bf16 *>(p.wd)[id] = h;
                } else {
                    const _Float16 h = (_Float16)r;
                    reinterpret_cast<_Float16 *>(p.wb)[ib] = h;
                    reinterpret_cast<_Float16 *>(p.wd)[id] = h;
                }
            }
        }
        C[(long)gm * p.scm + (long)gn * p.scn] = r;
    }
}


// ------------------------------------------------------------------------------------------------
// Round 3 forward-merge GEMM (B(k,n) contiguous in n; A any strides): same arithmetic as gemm_body -- one k-ordered fmaf
// chain per output on v_mfma_f32_32x32x2_f32, bit-identical to oracle/merge_ref.c -- rebuilt around its latency chain:
//   * K chunks of 32, two LDS stages, ONE barrier per chunk; the global loads of chunk i+2 are issued before the MFMAs of
//     chunk i and stored to LDS right after the next barrier (a whole chunk of MFMAs to land in);
//   * 16-byte global loads wherever rows are 16-byte aligned (W3, T and W1 of the C = 96 layers), clamped scalar loads
//     otherwise (W2 is read with its stride of 9 floats);
//   * A is kept [row][k] in LDS (16-byte chunks XOR-swizzled with (row >> 1) & 7: conflict-free ds_read_b128); a lane reads
//     the 4 k of two consecutive MFMA steps at once and picks its half's element (k = 2 s + (lane >> 5)) with one select;
//   * work-group tile = (32 WM) x (32 WN): 2x2 waves for the S products, 4x1 for the T products (N = C = 96 or 26 columns).
#define G2_KC 32
#ifndef G2_ABL
#define G2_ABL 0          // timing ablations of the K loop (diagnostic builds only; results WRONG): 1 no global loads, 2 no LDS stores, 4 no fragment reads, 8 no barrier
#endif
template <int WM, int WN>
struct G2Regs { float4 a[WM], b[WN]; };

#ifdef ORN_MERGE_STAMP
// Diagnostic build only (tools/probes/merge_stamps.py): s_memtime phase sums of work-group (0,0,0), wave 0.
// per problem shape (slot (K / 32) % 8, 16 words): [0] whole kernel  [1] prologue  [2] K loop  [3] epilogue  [4] sum of the loop
// iterations  [5] chunks  [6..8] M, N, K
__device__ unsigned long long g_mst[128];
extern "C" __attribute__((visibility("default"))) int orn_debug_merge_stamps(unsigned long long *out128)
{
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_mst), sizeof(g_mst));
}
#define MST_NOW(t_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MST_NOW(t_) do { } while (0)
#endif

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void g2_sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        g2_sfor<I + 1, N>(f);
    }
}

template <int WM, int WN, bool AVEC, bool BVEC>
__device__ __forceinline__ void gemm_body2v(const GemmP &p, int bx, int by, int bz, float *lds)
{
    constexpr int BM = 32 * WM, BN = 32 * WN;
    // one LDS stage: A as two blocks [k parity][BM rows][16 floats] (a lane half of the 32x32x2 MFMA consumes one parity:
    // k = 2 s + (lane >> 5), so its 4 floats of a 16-byte read are the operands of 4 consecutive MFMA steps -- no per-step
    // select), 16-byte chunks XOR-swizzled with (row >> 2) & 3 (conflict-free ds_read_b128); then B [32 k][BN]
    constexpr int STAGE = BM * G2_KC + G2_KC * BN;
    constexpr int BCH = BN / 4;                               // 16-byte chunks per B row
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = by * BM, n0 = bx * BN;
    const float *A = p.A + (long)bz * p.ba;
    const float *B = p.B + (long)bz * p.bb;
    float *C = p.C + (long)bz * p.bc;
    const int M = p.M, N = p.N, K = p.K;
    const int sam = (int)p.sam, sak = (int)p.sak, sbk = (int)p.sbk;
    constexpr bool avec = AVEC;                    // A is contiguous in k (compile-time: the load forms differ)
    static_assert(BVEC, "B is always loaded 16 bytes at a time");
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging units: A unit u = t + 256 i -> (row u >> 3, 4 k from 4 (u & 7)); B unit u -> (k row u / BCH, 4 n from 4 (u % BCH))
    // Loads go through raw buffer descriptors: a read beyond the end of a matrix returns 0 and never faults, so nothing is
    // clamped and nothing is masked (round 3 first clamped every address and zeroed the K tail with 16 selects per unit:
    // ~150 VALU instructions per chunk on the scalar path, three times what hides under the MFMA chain).  The K tail: B's
    // rows k >= K lie behind its last element and read as 0; A's columns k >= K read the next row (finite weights) or 0,
    // and fma(a, 0, acc) == acc keeps the chain exact.  Rows / columns beyond M / N only feed outputs that are never stored.
    const unsigned a_bytes = ((unsigned)(M - 1) * (unsigned)sam + (unsigned)(K - 1) * (unsigned)sak + 1u) * 4u;
    const unsigned b_bytes = ((unsigned)(K - 1) * (unsigned)sbk + (unsigned)N) * 4u;
    const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, a_bytes, 0x00020000);
    const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, b_bytes, 0x00020000);
    int a_dst[WM], b_dst[WN];                              // LDS offsets inside a stage (floats)
    unsigned a_vo[WM][avec ? 1 : 4], b_vo[WN];             // byte offsets of the unit's elements at k0 = 0
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int u = t + 256 * i, row = u >> 3, ch = u & 7;
#pragma unroll
        for (int j = 0; j < (avec ? 1 : 4); ++j) a_vo[i][j] = ((unsigned)(m0 + row) * (unsigned)sam + (unsigned)(4 * ch + j) * (unsigned)sak) * 4u;
        a_dst[i] = row * 16 + 4 * ((ch >> 1) ^ ((row >> 2) & 3)) + 2 * (ch & 1);
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int u = t + 256 * i, kr = u / BCH, ch = u - kr * BCH;
        b_vo[i] = ((unsigned)kr * (unsigned)sbk + (unsigned)(n0 + 4 * ch)) * 4u;
        b_dst[i] = BM * G2_KC + kr * BN + 4 * ch;
    }
    typedef G2Regs<WM, WN> Regs;
    // (bit_cast of the WHOLE result: indexing the builtin's vector_size result element by element, v[1], v[2] .., made hipcc 7.2
    // load one dword and use it for every element)
#define G2_LOAD16(rsrc_, vo_, so_) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_, vo_, so_, 0))
    auto gload_a = [&](Regs &R, int i, int k0) {
        const unsigned so = (unsigned)k0 * (unsigned)sak * 4u;      // wave-uniform: the instruction's scalar offset
        if constexpr (avec) {                                  // k contiguous: one 16-byte load (any 4-byte alignment)
            R.a[i] = G2_LOAD16(rA, a_vo[i][0], so);
        } else {
            R.a[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_vo[i][0], so, 0));
            R.a[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_vo[i][1], so, 0));
            R.a[i].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_vo[i][2], so, 0));
            R.a[i].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_vo[i][3], so, 0));
        }
    };
    auto gload_b = [&](Regs &R, int i, int k0) {
        R.b[i] = G2_LOAD16(rB, b_vo[i], (unsigned)k0 * (unsigned)sbk * 4u);
    };
    auto lstore_a = [&](const Regs &R, int i, int, float *st) {
        const float4 v = R.a[i];
        float *d = st + a_dst[i];
        *reinterpret_cast<float2 *>(d) = make_float2(v.x, v.z);                  // even k
        *reinterpret_cast<float2 *>(d + BM * 16) = make_float2(v.y, v.w);        // odd k
    };
    auto lstore_b = [&](const Regs &R, int i, int, float *st) { *reinterpret_cast<float4 *>(st + b_dst[i]) = R.b[i]; };
    auto gload = [&](Regs &R, int k0) {
#pragma unroll
        for (int i = 0; i < WM; ++i) gload_a(R, i, k0);
#pragma unroll
        for (int i = 0; i < WN; ++i) gload_b(R, i, k0);
    };
    auto lstore = [&](const Regs &R, int k0, float *st) {
#pragma unroll
        for (int i = 0; i < WM; ++i) lstore_a(R, i, k0, st);
#pragma unroll
        for (int i = 0; i < WN; ++i) lstore_b(R, i, k0, st);
    };
    // Software pipeline over the K chunks.  Everything rotates modulo 3 -- chunk c is loaded into staging register set c % 3,
    // stored into LDS stage c % 3, and its MFMA fragments are read into fragment set c % 3 -- and the loop is unrolled by
    // three, so every name is static:
    //   iteration i:  16 MFMAs of chunk i, and BETWEEN them (a wave is in-order: what is written behind the chain starts
    //                 when the chain has issued) the fragment reads of chunk i+1, the global loads of chunk i+4, then
    //                 chunk i+2 -> LDS | barrier
    // A chunk's loads have two iterations (~2500 cycles) to land before they are stored: inside a training step these
    // operands come from the MALL or HBM (the launch follows 200 MB of Adam traffic), 600-2000 cycles away; with one
    // iteration the chain of the first block (K = 650, 21 chunks) waited on them every chunk.  The LDS round trip of chunk
    // i+1 and all staging traffic sit under the MFMA chain of chunk i (a wave has no partner on its SIMD in these launches:
    // ~1.2 waves per SIMD chip-wide).  What the barrier at the end of iteration i orders: chunk i+2 visible to the reads of
    // iteration i+1; the reads of chunk i (finished before its MFMAs) before chunk i+3 overwrites their stage in iteration
    // i+1.  The chunk count is rounded up to a multiple of three (a chunk behind K is all zeros: fma(0, 0, acc) == acc).
    struct Frag { float4 a4[G2_KC / 8]; float b[G2_KC / 2]; };
    const int arow = wm * 32 + l31;
    const int a_rd = hh * (BM * 16) + arow * 16, a_sw = (arow >> 2) & 3;
    const int b_rd = BM * G2_KC + hh * BN + wn * 32 + l31;
    // six pieces of two LDS instructions each: 0, 1 = the A fragments (two ds_read_b128), 2..5 = four B values (two ds_read2st64_b32)
    auto fread_piece = [&](Frag &F, const float *st, int pc) {
        if (pc < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) F.a4[2 * pc + j] = *reinterpret_cast<const float4 *>(st + a_rd + 4 * ((2 * pc + j) ^ a_sw));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) F.b[4 * (pc - 2) + j] = st[b_rd + (2 * (4 * (pc - 2) + j)) * BN];
        }
    };
    auto fread = [&](Frag &F, const float *st) {
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) fread_piece(F, st, pc);
    };
    const int nreal = (K + G2_KC - 1) / G2_KC, nch = (nreal + 2) / 3 * 3;
    Frag F0, F1, F2;
    Regs R0, R1, R2;
    // the combine epilogue's branch terms (3 loads per output row, cold inside a training step) are requested before the K
    // loop: 16 registers held across it instead of a ~2 us round trip behind it
    const int gn = n0 + wn * 32 + l31, gnc = min(gn, N - 1);
    const int ec = gnc / 9, eij = gnc - ec * 9, eii = eij / 3, ejj = eij - eii * 3;
    float base[16];
    if (p.epi == 1) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gm = min(m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, p.M - 1);
            const long oc = (long)gm * p.Cch + ec;
            const float p13 = (eii == 1) ? p.w1x3[oc * 3 + ejj] : 0.f;
            const float p31 = (ejj == 1) ? p.w3x1[oc * 3 + eii] : 0.f;
            base[reg] = p.w3x3[(long)gm * p.N + gnc] + (p13 + p31);         // association of model.py:475,495
        }
    }
    unsigned long long t0 = 0, t1 = 0, t2 = 0, ta = 0, tb = 0, s6 = 0;
    (void)t0; (void)t1; (void)t2; (void)ta; (void)tb; (void)s6;
    MST_NOW(t0);
    gload(R0, 0);
    gload(R1, G2_KC);
    gload(R2, 2 * G2_KC);
    __syncthreads();                                       // the LDS block may still be in use by a previous role of this array
    lstore(R0, 0, lds);
    lstore(R1, G2_KC, lds + STAGE);                        // chunk 1 (zeros if it lies behind K)
    gload(R0, 3 * G2_KC);
    __syncthreads();
    fread(F0, lds);                                        // chunk 0
    MST_NOW(t1);
    constexpr int NPC = WM + WN;                           // staging pieces per chunk (one 16-byte unit each)
    static_assert(2 * NPC + 6 <= G2_KC / 2, "more staging pieces than MFMA slots");
    // Fcur: chunk i; Fnext <- chunk i+1 from stage sr; Rnew <- chunk i+4; Rold (chunk i+2) -> stage sw
    auto step = [&](Frag &Fcur, Frag &Fnext, Regs &Rold, Regs &Rnew, const float *sr, float *sw, int i) {
        MST_NOW(ta);
        // No conditions on the chunk index in here: the step is ONE basic block (behind per-piece branches hipcc's wait
        // insertion put vmcnt(0) in front of every load, i.e. one L2 round trip per load).  Chunks behind K are read with
        // clamped addresses and stored as zeros into stages nobody reads any more; the reads behind the last chunk are unused.
        const int k_st = (i + 2) * G2_KC, k_ld = (i + 4) * G2_KC;
        // one slot behind each of the 16 MFMAs (what fits under a 64-cycle MFMA without delaying the next one is ~10 simple
        // instructions; a pair's worth of staging in ONE gap cost 340 cycles per chunk): slots 0 .. NPC-1 the global loads,
        // then six slots of two fragment reads, then the LDS stores
        g2_sfor<0, G2_KC / 2>([&](auto sc) {
            constexpr int sl = decltype(sc)::value;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Fcur.a4[sl >> 2][sl & 3], Fcur.b[sl], acc, 0, 0, 0);
            if constexpr (sl < NPC && !(G2_ABL & 1)) {
                if constexpr (sl < WM) gload_a(Rnew, sl, k_ld); else gload_b(Rnew, sl - WM, k_ld);
            } else if constexpr (sl >= NPC && sl < NPC + 6 && !(G2_ABL & 4)) {
                fread_piece(Fnext, sr, sl - NPC);
            } else if constexpr (sl >= NPC + 6 && sl < 2 * NPC + 6 && !(G2_ABL & 2)) {
                constexpr int j = sl - NPC - 6;
                if constexpr (j < WM) lstore_a(Rold, j, k_st, sw); else lstore_b(Rold, j - WM, k_st, sw);
            }
            __builtin_amdgcn_sched_barrier(0);             // pins the interleave: one MFMA, then its slot's piece
        });
        if constexpr (!(G2_ABL & 8)) __syncthreads();
        MST_NOW(tb);
#ifdef ORN_MERGE_STAMP
        s6 += tb - ta;
#endif
    };
    float *S0 = lds, *S1 = lds + STAGE, *S2 = lds + 2 * STAGE;
    for (int i = 0; i < nch; i += 3) {
        step(F0, F1, R2, R1, S1, S2, i);                   // chunk i+4 -> R1 (its chunk i+1 went to LDS an iteration ago), chunk i+2 (R2) -> S2
        step(F1, F2, R0, R2, S2, S0, i + 1);               // chunk i+5 -> R2, chunk i+3 (R0) -> S0
        step(F2, F0, R1, R0, S0, S1, i + 2);               // chunk i+6 -> R0, chunk i+4 (R1) -> S1
    }
    MST_NOW(t2);
    if (p.epi == 1 && p.bf && bx == 0 && wn == 0 && l31 == 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (gm < p.M) {
                const float b = p.b3x3[gm] + (p.b1x3[gm] + p.b3x1[gm]);     // model.py:476,496
                p.bf[gm] = b;
                if (p.half_kind) p.biasp[(gm % p.s2) * p.Cn + gm / p.s2] = b;
            }
        }
    }
    if (gn < p.N) {
    if (p.epi == 1) {
        const int c = ec, ij = eij;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (gm >= p.M) continue;
            const float r = base[reg] + acc[reg];
            if (p.half_kind) {
                const int op = (gm % p.s2) * p.Cn + gm / p.s2;
                const size_t ib = ((size_t)ij * p.M + op) * p.Cp + c, id = ((size_t)(8 - ij) * p.Cp + c) * p.M + op;
                if (p.half_kind == 1) {
                    const __bf16 h = (__bf16)r;
                    reinterpret_cast<__bf16 *>(p.wb)[ib] = h;
                    reinterpret_cast<__bf16 *>(p.wd)[id] = h;
                } else {
                    const _Float16 h = (_Float16)r;
                    reinterpret_cast<_Float16 *>(p.wb)[ib] = h;
                    reinterpret_cast<_Float16 *>(p.wd)[id] = h;
                }
            }
            C[(long)gm * p.scm + (long)gn * p.scn] = r;
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (gm < p.M) C[(long)gm * p.scm + (long)gn * p.scn] = acc[reg];
        }
    }
    }
#ifdef ORN_MERGE_STAMP
    {
        unsigned long long t3;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MST_NOW(t3);
        if (bx == 0 && by == 0 && bz == 0 && t == 0) {
            // one record per problem shape: slot = (K / 32) % 8, 16 words each (a grouped launch runs several problems)
            unsigned long long *rec = g_mst + 16 * ((K / 32) & 7);
            rec[0] = t3 - t0; rec[1] = t1 - t0; rec[2] = t2 - t1; rec[3] = t3 - t2; rec[4] = s6; rec[5] = nch; rec[6] = M; rec[7] = N; rec[8] = K;
        }
    }
#endif
}
#define G2_LDS_FLOATS (3 * (128 * G2_KC + G2_KC * 32))      // three stages of the larger of the two tile shapes (4x1): 60 KB
#define G2_LDS_BYTES(shape_) ((shape_) == 1 ? 3 * (128 * G2_KC + G2_KC * 32) * 4 : 3 * (64 * G2_KC + G2_KC * 64) * 4)   // 60 KB / 48 KB
#define G2_LIN_ROUNDS 4                                     // output neurons per wave of a stem-layer rider work-group
template <int WM, int WN>
__device__ __forceinline__ void gemm_body2(const GemmP &p, int bx, int by, int bz, float *lds)
{
    if (p.a_vec) gemm_body2v<WM, WN, true, true>(p, bx, by, bz, lds);
    else gemm_body2v<WM, WN, false, true>(p, bx, by, bz, lds);
}

// Same tiling on v_mfma_f32_32x32x16_f16 (fp32 accumulate): operands are rounded to IEEE half while they are
// staged into LDS ([row][k], k contiguous, 144-byte rows: conflict-free ds_read_b128).  Used for the merge BACKWARD
// in the 16-bit engine modes only -- the forward merge stays exact fp32.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
#define HK 64
#define HROW (HK * 2 + 16)

__device__ __forceinline__ void gemm_body_h16(const GemmP &p, int bx, int by, int bz)
{
    __shared__ __attribute__((aligned(16))) unsigned char Ah[GT * HROW];
    __shared__ __attribute__((aligned(16))) unsigned char Bh[GT * HROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * GT, n0 = bx * GT;
    const float *A = p.A + (long)bz * p.ba;
    const float *B = p.B + (long)bz * p.bb;
    float *C = p.C + (long)bz * p.bc;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NE = GT * HK / 256;
    float ra[NE], rb[NE];
    const int tf = t & 63, tv = t >> 6;
    const int M = p.M, N = p.N, K = p.K;
    const int sam = (int)p.sam, sak = (int)p.sak, sbk = (int)p.sbk, sbn = (int)p.sbn;
    const bool akf = p.a_kfast, bnf = p.b_nfast;
    auto gload = [&](int k0) {                  // unconditional clamped loads, see gemm_body
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int v = tv + 4 * i;
            const int am = akf ? v : tf, ak = k0 + (akf ? tf : v);
            const int bn = bnf ? tf : v, bk = k0 + (bnf ? v : tf);
            const float va = A[__mul24(min(m0 + am, M - 1), sam) + __mul24(min(ak, K - 1), sak)];
            const float vb = B[__mul24(min(bk, K - 1), sbk) + __mul24(min(n0 + bn, N - 1), sbn)];
            ra[i] = va;                         // the K-tail zeroing happens at the LDS store: consuming the value here
            rb[i] = vb;                         // would put the load wait in front of this chunk's MFMAs
        }
    };
    gload(0);
    for (int k0 = 0; k0 < p.K; k0 += HK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int v = tv + 4 * i;
            const float xa = (k0 + (akf ? tf : v) < K) ? ra[i] : 0.f, xb = (k0 + (bnf ? v : tf) < K) ? rb[i] : 0.f;
            *reinterpret_cast<_Float16 *>(Ah + (akf ? v : tf) * HROW + (akf ? tf : v) * 2) = (_Float16)(xa * p.sa);
            *reinterpret_cast<_Float16 *>(Bh + (bnf ? tf : v) * HROW + (bnf ? v : tf) * 2) = (_Float16)(xb * p.sb);
        }
        __syncthreads();
        if (k0 + HK < p.K) gload(k0 + HK);
        const unsigned char *ap = Ah + (wm * 32 + l31) * HROW + hh * 16;
        const unsigned char *bp = Bh + (wn * 32 + l31) * HROW + hh * 16;
#pragma unroll
        for (int ks = 0; ks < HK / 16; ++ks) {
            const f16x8 a = *reinterpret_cast<const f16x8 *>(ap + ks * 32);
            const f16x8 b = *reinterpret_cast<const f16x8 *>(bp + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    const int gn = n0 + wn * 32 + l31;
    if (gn >= p.N) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (gm < p.M) C[(long)gm * p.scm + (long)gn * p.scn] = acc[reg] * p.so;
    }
}

__global__ void __launch_bounds__(256) k_gemm_f32(GemmP p) { gemm_body(p, blockIdx.x, blockIdx.y, blockIdx.z); }
// forward products (B contiguous in n) on the round-3 body, 64x64 tiles
__global__ void __launch_bounds__(256) k_gemm2_f32(GemmP p)
{
    __shared__ __attribute__((aligned(16))) float lds[G2_LDS_FLOATS];
    gemm_body2<2, 2>(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Grouped form: several independent problems in one launch (the engine merges all layers at once so
// the small per-layer GEMMs fill the chip together).  The table lives in device memory.
#define GEMM_MAXP 16
struct GemmGroup {
    int n;
    int h16;                    // 1: problems run on the 16-bit MFMA variant
    int shape;                  // k_gemm_f32_grouped_linear: 0 = 64x64 tiles (2x2 waves), 1 = 128x32 tiles (4x1 waves)
    int tile_start[GEMM_MAXP + 1];
    GemmP prob[GEMM_MAXP];
};

__global__ void __launch_bounds__(256) k_gemm_f32_grouped(const GemmGroup *__restrict__ g)
{
    const int bid = blockIdx.x;
    int pi = 0;
    while (pi + 1 < g->n && bid >= g->tile_start[pi + 1]) ++pi;
    const GemmP p = g->prob[pi];
    const int local = bid - g->tile_start[pi];
    const int tn = (p.N + GT - 1) / GT, tm = (p.M + GT - 1) / GT;
    const int bz = local / (tn * tm), rem = local - bz * tn * tm;
    if (g->h16) gemm_body_h16(p, rem % tn, rem / tn, bz);
    else gemm_body(p, rem % tn, rem / tn, bz);
}

// The same launch with one of the stem's linear layers riding along as trailing work-groups (4 output neurons each): the
// stem and the merge are independent latency-bound chains at the head of the step, and a graph node costs ~5 us by itself.
__global__ void __launch_bounds__(256) k_gemm_f32_grouped_linear(const GemmGroup *__restrict__ g, OrnLinearJob job, int gemm_tiles, int lin_blocks,
                                                                 MhPackAll pack, int tile_off)
{
    ORN_PRIO_HIGH();
    // one dynamic array for every role of the launch, sized for the launch's tile shape: the static 60 KB of the larger
    // shape held the S launch to two work-groups per CU, and its ~2,400 short rider work-groups queued behind the 320 GEMM
    // tiles for the remaining slots -- 19 us beyond the tiles' own 23
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // the GEMM tiles are the long latency chains: they are dispatched first (largest K first), the short work-groups behind
    // them: the stem's linear layer, then (16-bit engine modes, S launch) the parameter-side half copies of the merge
    // BACKWARD's operands
    if ((int)blockIdx.x >= gemm_tiles + lin_blocks) {
        int layer, pjob;
        const int blk = mh_pack_decode(pack, MH_TAB_T, (int)blockIdx.x - gemm_tiles - lin_blocks, layer, pjob);
        mh_pack_block(pack, layer, pjob, blk, reinterpret_cast<float (*)[65]>(lds));
        return;
    }
    if ((int)blockIdx.x >= gemm_tiles) {
#pragma unroll 1
        for (int r = 0; r < G2_LIN_ROUNDS; ++r)
            orn_linear_silu_wave(job, ((blockIdx.x - gemm_tiles) * G2_LIN_ROUNDS + r) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    const int bid = blockIdx.x + tile_off;            // (tile_off != 0: probe only, a launch of part of the tiles)
    int pi = 0;
    while (pi + 1 < g->n && bid >= g->tile_start[pi + 1]) ++pi;
    const GemmP p = g->prob[pi];
    const int local = bid - g->tile_start[pi];
    if (g->shape == 1) {
        const int tn = (p.N + 31) / 32, tm = (p.M + 127) / 128;
        const int bz = local / (tn * tm), rem = local - bz * tn * tm;
        gemm_body2<4, 1>(p, rem % tn, rem / tn, bz, lds);
    } else {
        const int tn = (p.N + 63) / 64, tm = (p.M + 63) / 64;
        const int bz = local / (tn * tm), rem = local - bz * tn * tm;
        gemm_body2<2, 2>(p, rem % tn, rem / tn, bz, lds);
    }
}

static void finish_gemm(GemmP &p)
{
    p.a_kfast = (labs(p.sak) <= labs(p.sam)) ? 1 : 0;
    p.b_nfast = (labs(p.sbn) <= labs(p.sbk)) ? 1 : 0;
    p.a_vec = p.sak == 1 ? 1 : 0;          // gemm_body2: A rows contiguous in k -> 16-byte buffer loads (4-byte alignment is enough)
    p.b_vec = p.sbn == 1 ? 1 : 0;
}

static int launch_gemm(GemmP p, int batch, hipStream_t st, const char *name)
{
    finish_gemm(p);
    static const bool old_body = orn_probe_env("ORN_MERGE_OLD_BODY") != nullptr;      // probe switch (tools/probes/merge_probe.py)
    if (p.sbn == 1 && !old_body) hipLaunchKernelGGL(k_gemm2_f32, dim3(orn_cdiv(p.N, GT), orn_cdiv(p.M, GT), batch), dim3(256), 0, st, p);
    else
    hipLaunchKernelGGL(k_gemm_f32, dim3(orn_cdiv(p.N, GT), orn_cdiv(p.M, GT), batch), dim3(256), 0, st, p);
    ORN_LAUNCH_CHECK(name);
    return 0;
}

__global__ void k_bias3(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c, int n,
                        float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + (b[i] + c[i]);     // b3x3 + (b1x3 + b3x1), model.py:476,496
}

// T[m,c,ij] = sum_k W2[m,k,ij] * W1[k,c]     (9 batches over ij)
static GemmP prob_T(const float *w1, const float *w2, int C, int O, float *T)
{
    const int K2 = 2 * C;
    GemmP p = {};
    p.A = w2; p.B = w1; p.C = T;
    p.M = O; p.N = C; p.K = K2;
    p.sam = (long)K2 * 9; p.sak = 9; p.sbk = C; p.sbn = 1; p.scm = (long)C * 9; p.scn = 9;
    p.ba = 1; p.bb = 0; p.bc = 1; p.epi = 0;
    return p;
}
// same product from the tap-major copy w2t [9][O][2C]: A rows are contiguous in k (16-byte loads) instead of a gather with a
// stride of 9 floats (64 cache lines per wave-instruction: the T launch was bound by the address path, 31 us)
static GemmP prob_T_tapmajor(const float *w1, const float *w2t, int C, int O, float *T)
{
    const int K2 = 2 * C;
    GemmP p = prob_T(w1, w2t, C, O, T);
    p.sam = K2; p.sak = 1; p.ba = (long)O * K2;
    return p;
}
// Wf[o,e] = (W3x3 + (P13 + P31))[o,e] + sum_m W3[o,m] * T[m,e]
static GemmP prob_S(const float *w3x3, const float *w3x1, const float *w1x3, const float *w3, const float *T, int C, int O,
                    float *wf)
{
    GemmP q = {};
    q.A = w3; q.B = T; q.C = wf;
    q.M = O; q.N = C * 9; q.K = O;
    q.sam = O; q.sak = 1; q.sbk = (long)C * 9; q.sbn = 1; q.scm = (long)C * 9; q.scn = 1;
    q.epi = 1; q.w3x3 = w3x3; q.w1x3 = w1x3; q.w3x1 = w3x1; q.Cch = C;
    return q;
}

int orn_launch_merge_fwd(const float *w3x3, const float *b3x3, const float *w3x1, const float *b3x1,
                         const float *w1x3, const float *b1x3, const float *w1, const float *w2, const float *w3,
                         int C, int O, float *T, float *wf, float *bf, hipStream_t st)
{
    ORN_TRY(launch_gemm(prob_T(w1, w2, C, O, T), 9, st, "merge_T"));
    ORN_TRY(launch_gemm(prob_S(w3x3, w3x1, w1x3, w3, T, C, O, wf), 1, st, "merge_S"));
    hipLaunchKernelGGL(k_bias3, dim3(orn_cdiv(O, 256)), dim3(256), 0, st, b3x3, b1x3, b3x1, O, bf);
    ORN_LAUNCH_CHECK("merge_bias");
    return 0;
}

extern "C" int orn_erb_merge_fwd(const float *w3x3, const float *b3x3, const float *w3x1, const float *b3x1,
                                 const float *w1x3, const float *b1x3, const float *w1, const float *w2,
                                 const float *w3, int C, int O, float *T, float *wf, float *bf, void *stream)
{
    ORN_REQUIRE(w3x3 && b3x3 && w3x1 && b3x1 && w1x3 && b1x3 && w1 && w2 && w3 && T && wf && bf, "erb_merge_fwd: null pointer");
    ORN_REQUIRE(C > 0 && O > 0, "erb_merge_fwd: bad sizes C=%d O=%d", C, O);
    return orn_launch_merge_fwd(w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3, C, O, T, wf, bf, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// backward (closed forms of SURVEY 8a row A3)
// ------------------------------------------------------------------------------------------------
// d1x3[o,c,0,j] = G[o,c,1,j];  d3x1[o,c,i,0] = G[o,c,i,1];  optional copy d3x3 = G; bias fan-out
__global__ void k_merge_bwd_slices(const float *__restrict__ g, const float *__restrict__ dbf, long OC, int O,
                                   float *__restrict__ d3x3, float *__restrict__ db3x3, float *__restrict__ d3x1,
                                   float *__restrict__ db3x1, float *__restrict__ d1x3, float *__restrict__ db1x3)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < OC) {
        const float *gg = g + i * 9;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = gg[k];
        if (d3x3 != g) {
#pragma unroll
            for (int k = 0; k < 9; ++k) d3x3[i * 9 + k] = v[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d1x3[i * 3 + k] = v[3 + k];
            d3x1[i * 3 + k] = v[3 * k + 1];
        }
    }
    if (i < O) {
        const float d = dbf[i];
        if (db3x3 != dbf) db3x3[i] = d;
        db3x1[i] = d;
        db1x3[i] = d;
    }
}

// dW3[o,m] = sum_e G[o,e] * T[m,e]
static GemmP prob_dW3(const float *g, const float *T, int C, int O, float *dw3)
{
    const long n = (long)C * 9;
    GemmP p = {};
    p.A = g; p.B = T; p.C = dw3; p.M = O; p.N = O; p.K = (int)n;
    p.sam = n; p.sak = 1; p.sbk = 1; p.sbn = n; p.scm = O; p.scn = 1;
    return p;
}
// dT[m,e] = sum_o W3[o,m] * G[o,e]
static GemmP prob_dT(const float *g, const float *w3, int C, int O, float *dT)
{
    const long n = (long)C * 9;
    GemmP p = {};
    p.A = w3; p.B = g; p.C = dT; p.M = O; p.N = (int)n; p.K = O;
    p.sam = 1; p.sak = O; p.sbk = n; p.sbn = 1; p.scm = n; p.scn = 1;
    return p;
}
// dW2[m,k,ij] = sum_c dT[m,c,ij] * W1[k,c]        (9 batches)
static GemmP prob_dW2(const float *dT, const float *w1, int C, int O, float *dw2)
{
    const long n = (long)C * 9;
    const int K2 = 2 * C;
    GemmP p = {};
    p.A = dT; p.B = w1; p.C = dw2; p.M = O; p.N = K2; p.K = C;
    p.sam = n; p.sak = 9; p.sbk = 1; p.sbn = C; p.scm = (long)K2 * 9; p.scn = 9;
    p.ba = 1; p.bb = 0; p.bc = 1;
    return p;
}
// dW1[k,c] = sum_ij sum_m W2[m,k,ij] * dT[m,c,ij]  (9 batched partials, then a fixed-order sum)
static GemmP prob_dW1p(const float *w2, const float *dT, int C, int O, float *dw1p)
{
    const long n = (long)C * 9;
    const int K2 = 2 * C;
    GemmP p = {};
    p.A = w2; p.B = dT; p.C = dw1p; p.M = K2; p.N = C; p.K = O;
    p.sam = 9; p.sak = (long)K2 * 9; p.sbk = n; p.sbn = 9; p.scm = C; p.scn = 1;
    p.ba = 1; p.bb = 1; p.bc = (long)K2 * C;
    return p;
}

extern "C" size_t orn_erb_merge_bwd_ws_bytes(int C, int O)
{
    // dT [O*C*9] + dW1 partials [9][2C*C]
    return orn_align(((size_t)O * C * 9 + 9 * (size_t)2 * C * C) * sizeof(float));
}

int orn_launch_merge_bwd(const float *g, const float *dbf, const float *w1, const float *w2, const float *w3,
                         const float *T, int C, int O, float *d3x3, float *db3x3, float *d3x1, float *db3x1,
                         float *d1x3, float *db1x3, float *dw1, float *dw2, float *dw3, float *ws, hipStream_t st)
{
    const int K2 = 2 * C;
    const long n = (long)C * 9;
    float *dT = ws;
    float *dw1p = ws + (size_t)O * n;
    hipLaunchKernelGGL(k_merge_bwd_slices, dim3(orn_cdiv((long)O * C, 256)), dim3(256), 0, st, g, dbf, (long)O * C, O,
                       d3x3, db3x3, d3x1, db3x1, d1x3, db1x3);
    ORN_LAUNCH_CHECK("merge_bwd_slices");
    ORN_TRY(launch_gemm(prob_dW3(g, T, C, O, dw3), 1, st, "merge_dW3"));
    ORN_TRY(launch_gemm(prob_dT(g, w3, C, O, dT), 1, st, "merge_dT"));
    ORN_TRY(launch_gemm(prob_dW2(dT, w1, C, O, dw2), 9, st, "merge_dW2"));
    ORN_TRY(launch_gemm(prob_dW1p(w2, dT, C, O, dw1p), 9, st, "merge_dW1"));
    ORN_TRY(orn_launch_reduce_rows(dw1p, 9, (size_t)K2 * C, (size_t)K2 * C, dw1, st));
    return 0;
}

extern "C" int orn_erb_merge_bwd(const float *g, const float *dbf, const float *w1, const float *w2, const float *w3,
                                 const float *T, int C, int O, float *d3x3, float *db3x3, float *d3x1, float *db3x1,
                                 float *d1x3, float *db1x3, float *dw1, float *dw2, float *dw3, void *ws,
                                 size_t ws_bytes, void *stream)
{
    ORN_REQUIRE(g && dbf && w1 && w2 && w3 && T && d3x3 && db3x3 && d3x1 && db3x1 && d1x3 && db1x3 && dw1 && dw2 && dw3 && ws,
                "erb_merge_bwd: null pointer");
    ORN_REQUIRE(C > 0 && O > 0, "erb_merge_bwd: bad sizes");
    if (ws_bytes < orn_erb_merge_bwd_ws_bytes(C, O)) {
        orn_set_error("erb_merge_bwd: workspace %zu < %zu", ws_bytes, orn_erb_merge_bwd_ws_bytes(C, O));
        return ORN_E_WS;
    }
    return orn_launch_merge_bwd(g, dbf, w1, w2, w3, T, C, O, d3x3, db3x3, d3x1, db3x1, d1x3, db1x3, dw1, dw2, dw3,
                                (float *)ws, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// W2 [O][2C][9] -> w2t [9][O][2C], all layers in one launch: R rows per work-group through LDS (coalesced reads along a row's
// (k, tap) run, coalesced writes along k).
// ------------------------------------------------------------------------------------------------
#define W2T_ROWS 4
struct W2TAll { int n; int blk_start[ORN_MAX_LAYERS + 1]; struct { const float *w2; float *w2t; int O, K2; } l[ORN_MAX_LAYERS]; };

__global__ void __launch_bounds__(256) k_w2_transpose(W2TAll a, MhPackAll pack, int tr_blocks)
{
    ORN_PRIO_HIGH();
    extern __shared__ float w2s[];
    if ((int)blockIdx.x >= tr_blocks) {                    // riders: parameter-side half copies for the merge backward
        int layer, pjob;
        const int blk = mh_pack_decode(pack, MH_TAB_PAR, (int)blockIdx.x - tr_blocks, layer, pjob);
        mh_pack_block(pack, layer, pjob, blk, reinterpret_cast<float (*)[65]>(w2s));
        return;
    }
    int li = 0;
    while (li + 1 < a.n && (int)blockIdx.x >= a.blk_start[li + 1]) ++li;
    const auto &l = a.l[li];
    const int m0 = ((int)blockIdx.x - a.blk_start[li]) * W2T_ROWS, rows = min(W2T_ROWS, l.O - m0);
    const int E = l.K2 * 9, t = threadIdx.x;
    const float *src = l.w2 + (size_t)m0 * E;
    const int n = rows * E;
    for (int i0 = t; i0 < n; i0 += 14 * 256) {            // 14 loads in flight per thread (4 rows of 2C = 192: two rounds)
        float v[14];
#pragma unroll
        for (int u = 0; u < 14; ++u) v[u] = src[min(i0 + 256 * u, n - 1)];
#pragma unroll
        for (int u = 0; u < 14; ++u)
            if (i0 + 256 * u < n) w2s[i0 + 256 * u] = v[u];
    }
    __syncthreads();
    // wave w takes (tap, row) pairs w, w + 4, ...: a pair is one contiguous run of K2 floats (no per-element division)
    const int wave = t >> 6, lane = t & 63;
    for (int pb = wave; pb < 9 * rows; pb += 3 * 4) {       // three pairs per batch: their LDS reads go out together
        for (int k = lane; k < l.K2; k += 64) {
            float x[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int pr = min(pb + 4 * a, 9 * rows - 1), ij = pr / rows, r = pr - ij * rows;
                x[a] = w2s[r * E + ij + k * 9];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int pr = pb + 4 * a;
                if (pr < 9 * rows) {
                    const int ij = pr / rows, r = pr - ij * rows;
                    l.w2t[((size_t)ij * l.O + m0 + r) * l.K2 + k] = x[a];
                }
            }
        }
    }
}

int orn_launch_w2_transpose(int n_layers, const OrnMergeLayer *L, hipStream_t st, const void *pack, int par_blocks)
{
    W2TAll a;
    a.n = 0;
    a.blk_start[0] = 0;
    size_t smem = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (!L[i].w2t) continue;
        auto &l = a.l[a.n];
        l.w2 = L[i].w2; l.w2t = L[i].w2t; l.O = L[i].O; l.K2 = 2 * L[i].C;
        a.blk_start[a.n + 1] = a.blk_start[a.n] + orn_cdiv(l.O, W2T_ROWS);
        const size_t b = (size_t)W2T_ROWS * l.K2 * 9 * sizeof(float);
        if (b > smem) smem = b;
        ++a.n;
    }
    MhPackAll pk = {};
    if (pack && par_blocks > 0) { pk = *(const MhPackAll *)pack; if (smem < 64 * 65 * sizeof(float)) smem = 64 * 65 * sizeof(float); } else par_blocks = 0;
    if (a.n == 0 && par_blocks == 0) return 0;
    ORN_REQUIRE(smem <= 64 * 1024, "w2_transpose: 2C = %zu too wide for the LDS tile", smem / (W2T_ROWS * 9 * sizeof(float)));
    hipLaunchKernelGGL(k_w2_transpose, dim3(a.blk_start[a.n] + par_blocks), dim3(256), smem, st, a, pk, a.blk_start[a.n]);
    ORN_LAUNCH_CHECK("w2_transpose");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Grouped merge for the engine: all ERB layers in four GEMM launches per step.
// ------------------------------------------------------------------------------------------------
static void group_add(GemmGroup &g, GemmP p, int batch)
{
    finish_gemm(p);
    const int tiles = (g.shape == 1 ? orn_cdiv(p.N, 32) * orn_cdiv(p.M, 128) : orn_cdiv(p.N, GT) * orn_cdiv(p.M, GT)) * batch;
    g.prob[g.n] = p;
    g.tile_start[g.n + 1] = g.tile_start[g.n] + tiles;
    ++g.n;
}

size_t orn_merge_group_bytes() { return orn_align(4 * sizeof(GemmGroup)); }

// Builds the four device-resident problem tables (host side, at engine creation; synchronous copy).
int orn_merge_groups_build(void *dev_tables, int n_layers, const OrnMergeLayer *L, int bwd_h16)
{
    ORN_REQUIRE(2 * n_layers <= GEMM_MAXP, "merge groups: too many layers");
    GemmGroup *h = new GemmGroup[4]();
    h[0].shape = 1;             // T products: N = C columns -> 128x32 tiles
    // Dispatch order of the S products.  The launch has a few more work-groups than the chip has CUs (720p: 320), all resident
    // at once, two per CU where they double up -- observed: the LAST ones join the FIRST ones' CUs.  Two tiles on one CU
    // share its matrix pipes, so the one problem whose K chain is much longer than the others' (the first block, K = O = 650)
    // must not sit at either end: it goes behind as many tiles of the others as will double up.  Speed only.
    int s_order[ORN_MAX_LAYERS], n_ord = 0;
    {
        int longest = 0, total = 0;
        for (int i = 0; i < n_layers; ++i) {
            if (L[i].O > L[longest].O) longest = i;
            total += orn_cdiv(L[i].C * 9, GT) * orn_cdiv(L[i].O, GT);
        }
        const int overflow = total > 256 ? total - 256 : 0;
        int cum = 0;
        bool placed = false;
        for (int i = n_layers - 1; i >= 0; --i) {
            if (i == longest) continue;
            if (!placed && cum >= overflow) { s_order[n_ord++] = longest; placed = true; }
            s_order[n_ord++] = i;
            cum += orn_cdiv(L[i].C * 9, GT) * orn_cdiv(L[i].O, GT);
        }
        if (!placed) s_order[n_ord++] = longest;
    }
    for (int j = 0; j < n_layers; ++j) {
        const OrnMergeLayer &l = L[s_order[j]];
        GemmP q = prob_S(l.w3x3, l.w3x1, l.w1x3, l.w3, l.T, l.C, l.O, l.wf);
        q.b3x3 = l.b3x3; q.b1x3 = l.b1x3; q.b3x1 = l.b3x1; q.bf = l.bf;
        if (l.half_kind) { q.half_kind = l.half_kind; q.wb = l.wb; q.wd = l.wd; q.biasp = l.biasp; q.s2 = l.s2; q.Cn = l.O / l.s2; q.Cp = l.Cp; }
        group_add(h[1], q, 1);
    }
    for (int i = 0; i < n_layers; ++i) {
        const OrnMergeLayer &l = L[i];
        group_add(h[0], l.w2t ? prob_T_tapmajor(l.w1, l.w2t, l.C, l.O, l.T) : prob_T(l.w1, l.w2, l.C, l.O, l.T), 9);
        // gradient operands (dWf, dT ~ 1e-6) are scaled by 2^14 when rounded to half; weights are not
        const float GS = 16384.0f;
        GemmP q;
        q = prob_dW3(l.g, l.T, l.C, l.O, l.dw3);   q.sa = GS;   q.sb = 1.0f; q.so = 1.0f / GS; group_add(h[2], q, 1);
        q = prob_dT(l.g, l.w3, l.C, l.O, l.dT);    q.sa = 1.0f; q.sb = GS;   q.so = 1.0f / GS; group_add(h[2], q, 1);
        q = prob_dW2(l.dT, l.w1, l.C, l.O, l.dw2); q.sa = GS;   q.sb = 1.0f; q.so = 1.0f / GS; group_add(h[3], q, 9);
        q = prob_dW1p(l.w2, l.dT, l.C, l.O, l.dw1p); q.sa = 1.0f; q.sb = GS; q.so = 1.0f / GS; group_add(h[3], q, 9);
    }
    h[2].h16 = h[3].h16 = bwd_h16 ? 1 : 0;
    hipError_t e = hipMemcpy(dev_tables, h, 4 * sizeof(GemmGroup), hipMemcpyHostToDevice);
    delete[] h;
    if (e != hipSuccess) { orn_set_error("merge groups: hipMemcpy failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int orn_merge_group_tiles(int which, int n_layers, const OrnMergeLayer *L)
{
    int t = 0;
    for (int i = 0; i < n_layers; ++i) {
        const int C = L[i].C, O = L[i].O, n = C * 9, K2 = 2 * C;
        switch (which) {
        case 0: t += orn_cdiv(C, 32) * orn_cdiv(O, 128) * 9; break;
        case 1: t += orn_cdiv(n, GT) * orn_cdiv(O, GT); break;
        case 2: t += orn_cdiv(O, GT) * orn_cdiv(O, GT) + orn_cdiv(n, GT) * orn_cdiv(O, GT); break;
        default: t += orn_cdiv(K2, GT) * orn_cdiv(O, GT) * 9 + orn_cdiv(C, GT) * orn_cdiv(K2, GT) * 9; break;
        }
    }
    return t;
}

int orn_launch_merge_group(const void *dev_tables, int which, int tiles, hipStream_t st)
{
    const GemmGroup *g = (const GemmGroup *)dev_tables + which;
    hipLaunchKernelGGL(k_gemm_f32_grouped, dim3(tiles), dim3(256), 0, st, g);
    ORN_LAUNCH_CHECK("merge_group");
    return 0;
}

// fp32 groups only (which = 0 / 1: the forward merge)
int orn_launch_merge_group_linear(const void *dev_tables, int which, int tiles, const OrnLinearJob &job, hipStream_t st, const void *pack,
                                  int pack_blocks)
{
    const GemmGroup *g = (const GemmGroup *)dev_tables + which;
    int lin_blocks = orn_cdiv(job.N, 4 * G2_LIN_ROUNDS);
    const size_t smem = G2_LDS_BYTES(which == 0 ? 1 : 0) > 64 * 65 * 4 ? G2_LDS_BYTES(which == 0 ? 1 : 0) : 64 * 65 * 4;
    MhPackAll pk = {};
    if (pack && pack_blocks > 0) pk = *(const MhPackAll *)pack; else pack_blocks = 0;
#ifdef ORN_PROBE_BUILD      // diagnostic builds only (ORN_BUILD_TAG + ORN_EXTRA_DEFS=-DORN_PROBE_BUILD): timing switch, results WRONG
    static const int dbg = orn_probe_env_int("ORN_MERGE_DBG", 0);
#else
    constexpr int dbg = 0;
#endif
    if (dbg & 1) pack_blocks = 0;
    if (dbg & 2) lin_blocks = 0;
    if (dbg & 4) {                                    // one launch per problem: their durations inside a real step (rocprofv3 timeline)
        GemmGroup hg;
        if (hipMemcpy(&hg, g, sizeof(hg), hipMemcpyDeviceToHost) != hipSuccess) return ORN_E_ARG;
        for (int i = 0; i < hg.n; ++i) {
            const int nt = hg.tile_start[i + 1] - hg.tile_start[i];
            hipLaunchKernelGGL(k_gemm_f32_grouped_linear, dim3(nt), dim3(256), smem, st, g, job, nt, 0, pk, hg.tile_start[i]);
        }
        ORN_LAUNCH_CHECK("merge_group_linear(dbg)");
        return 0;
    }
    hipLaunchKernelGGL(k_gemm_f32_grouped_linear, dim3(tiles + lin_blocks + pack_blocks), dim3(256), smem, st, g, job, tiles, lin_blocks, pk, 0);
    ORN_LAUNCH_CHECK("merge_group_linear");
    return 0;
}

int orn_launch_merge_bias(const float *b3x3, const float *b1x3, const float *b3x1, int O, float *bf, hipStream_t st)
{
    hipLaunchKernelGGL(k_bias3, dim3(orn_cdiv(O, 256)), dim3(256), 0, st, b3x3, b1x3, b3x1, O, bf);
    ORN_LAUNCH_CHECK("merge_bias");
    return 0;
}

int orn_launch_merge_bwd_tail(const float *g, const float *dbf, int C, int O, float *d3x3, float *db3x3, float *d3x1,
                              float *db3x1, float *d1x3, float *db1x3, const float *dw1p, float *dw1, hipStream_t st)
{
    hipLaunchKernelGGL(k_merge_bwd_slices, dim3(orn_cdiv((long)O * C, 256)), dim3(256), 0, st, g, dbf, (long)O * C, O, d3x3,
                       db3x3, d3x1, db3x1, d1x3, db1x3);
    ORN_LAUNCH_CHECK("merge_bwd_slices");
    return orn_launch_reduce_rows(dw1p, 9, (size_t)2 * C * C, (size_t)2 * C * C, dw1, st);
}

// ------------------------------------------------------------------------------------------------
// Per-layer elementwise tails of the merge, all layers in one launch each (blockIdx.y = layer).
// ------------------------------------------------------------------------------------------------
struct MiscLayers {
    int n;
    OrnMergeMisc l[ORN_MAX_LAYERS];
};

__global__ void k_merge_bias_all(MiscLayers m)
{
    const OrnMergeMisc &l = m.l[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < l.O) l.bf[i] = l.b3x3[i] + (l.b1x3[i] + l.b3x1[i]);     // model.py:476,496
}

// slices of dWf into the 1x3 / 3x1 branches, bias fan-out, and dW1 = fixed-order sum of its 9 partials
__global__ void k_merge_bwd_tail_all(MiscLayers m)
{
    ORN_PRIO_HIGH();
    const OrnMergeMisc &l = m.l[blockIdx.y];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long OC = (long)l.O * l.C;
    if (i < OC) {
        const float *gg = l.g + i * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            l.d1x3[i * 3 + k] = gg[3 + k];
            l.d3x1[i * 3 + k] = gg[3 * k + 1];
        }
    }
    if (i < l.O) {
        const float d = l.dbf[i];
        l.db3x1[i] = d;
        l.db1x3[i] = d;
    }
    const long n1 = (long)2 * l.C * l.C;
    if (i < n1) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += l.dw1p[(long)k * n1 + i];
        l.dw1[i] = acc;
    }
    if (l.dw2t) {                                  // tap-major GEMM output -> the parameter's [O][2C][3][3] layout
        const long n2 = (long)l.O * 2 * l.C;
        for (long j = i; j < n2; j += (long)gridDim.x * blockDim.x) {
            float v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = l.dw2t[(long)k * n2 + j];
#pragma unroll
            for (int k = 0; k < 9; ++k) l.dw2[j * 9 + k] = v[k];
        }
    }
}

int orn_launch_merge_bias_all(int n, const OrnMergeMisc *L, hipStream_t st)
{
    MiscLayers m;
    m.n = n;
    int maxO = 0;
    for (int i = 0; i < n; ++i) { m.l[i] = L[i]; if (L[i].O > maxO) maxO = L[i].O; }
    hipLaunchKernelGGL(k_merge_bias_all, dim3(orn_cdiv(maxO, 256), n), dim3(256), 0, st, m);
    ORN_LAUNCH_CHECK("merge_bias_all");
    return 0;
}

int orn_launch_merge_bwd_tail_all(int n, const OrnMergeMisc *L, hipStream_t st)
{
    MiscLayers m;
    m.n = n;
    long mx = 0;
    for (int i = 0; i < n; ++i) {
        m.l[i] = L[i];
        const long w = (long)L[i].O * L[i].C > (long)2 * L[i].C * L[i].C ? (long)L[i].O * L[i].C : (long)2 * L[i].C * L[i].C;
        if (w > mx) mx = w;
    }
    hipLaunchKernelGGL(k_merge_bwd_tail_all, dim3(orn_cdiv(mx, 256), n), dim3(256), 0, st, m);
    ORN_LAUNCH_CHECK("merge_bwd_tail_all");
    return 0;
}
