// A3 backward (SURVEY 8a closed forms) for the engine's 16-bit modes: dW3, dT, dW2, dW1 of ALL ERB layers on
// v_mfma_f32_32x32x16_f16 with operands read straight from global memory into MFMA fragments.
//
// The generic GEMM in orn_merge.hip stages fp32 operands of arbitrary stride through LDS element by element; for these
// weight-sized problems (a few MB, L2 resident) that is ~400 instructions per 64-deep chunk and its two backward
// launches took 31 + 62 us for 3 GFLOP (this path: 21 us pack + 25 + 23 us).  Here one "pack" launch writes IEEE-half, K-CONTIGUOUS copies of every operand (rows padded
// to 16 halfs, gradient operands pre-scaled by 2^14 as before), after which a lane's MFMA fragment -- 8 consecutive k
// of one row -- is a single 16-byte global load.  No LDS, no barriers: every wave owns one 32x32 output tile.
//
//   dW3[o,m]      = sum_e        G[o,e]      T[m,e]            A = Gh  [O ][E ]     B = Th  [O ][E ]
//   dT [m,(c,ij)] = sum_o        W3[o,m]     G[o,(c,ij)]       A = W3T [O ][O ]     B = GT  [E ][O ]
//   dW2[m,k,ij]   = sum_c        dT[m,c,ij]  W1[k,c]           A = dTt [9][O ][C ]  B = W1h [2C][C ]
//   dW1p[ij][k,c] = sum_m        W2[m,k,ij]  dT[m,c,ij]        A = W2p [9][2C][O ]  B = dTc [9][C ][O ]
//
// (E = 9C; dT is produced directly in its two consumer layouts, in half, still carrying the 2^14 scale.)
#include "orn_internal.h"
#include "orn_merge_pack.h"
#include <new>

typedef __attribute__((ext_vector_type(8))) _Float16 mh16x8;
typedef __attribute__((ext_vector_type(16))) float mf32x16;


// ---- buffer sizes (halfs) of one layer -------------------------------------------------------------------------
struct MhSizes { size_t gh, gt, th, w3t, w1h, w2p, dtt, dtc; };
static MhSizes mh_sizes(int C, int O)
{
    const int E = 9 * C, K2 = 2 * C;
    MhSizes s;
    s.gh = (size_t)r32(O) * r16(E);
    s.gt = (size_t)r32(E) * r16(O);
    s.th = (size_t)r32(O) * r16(E);
    s.w3t = (size_t)r32(O) * r16(O);
    s.w1h = (size_t)r32(K2) * r16(C);
    s.w2p = (size_t)9 * r32(K2) * r16(O);
    s.dtt = (size_t)9 * r32(O) * r16(C);
    s.dtc = (size_t)9 * r32(C) * r16(O);
    return s;
}

size_t orn_merge_h16_layer_halfs(int C, int O)
{
    const MhSizes s = mh_sizes(C, O);
    const size_t parts[8] = {s.gh, s.gt, s.th, s.w3t, s.w1h, s.w2p, s.dtt, s.dtc};
    size_t tot = 0;
    for (int i = 0; i < 8; ++i) tot += (parts[i] + 127) / 128 * 128;       // 256-byte aligned sub-buffers
    return tot;
}

// ---- pack (orn_merge_pack.h): the gradient-side jobs; the parameter-side ones ride along the forward merge ---------
__global__ void __launch_bounds__(256) k_merge_pack(MhPackAll a, int fwd /* table: MH_TAB_* */)
{
    ORN_PRIO_HIGH();
    __shared__ float tile[64][65];
    int layer, job;
    const int blk = mh_pack_decode(a, fwd, (int)blockIdx.x, layer, job);
    mh_pack_block(a, layer, job, blk, tile);
}

// ---- GEMM ------------------------------------------------------------------------------------------------------
struct MhProb {
    const mh16 *A, *B;            // A [rows >= r32(M)][lda], B [rows >= r32(N)][ldb], both k-contiguous, zero padded
    int lda, ldb, M, N, K;        // K is a multiple of 16
    long ba, bb;                  // batch strides (halfs)
    int nbatch;
    int mode;                     // 0: fp32 store  C[b*bc + m*scm + n*scn] = acc * so;   1: dT in its two half layouts
    float *C; long scm, scn, bc; float so;
    mh16 *dtt, *dtc; int Cch, ldt, ldc; long st_t, st_c;   // mode 1: n = c*9 + ij
    int tiles_n, tiles;           // 32x32 tiles per batch: tiles_m * tiles_n
};
#define MH_MAXP 16
#define MH_KB 8                   // MFMA k-steps per software-pipeline batch
struct MhGroup { int n; int tile_start[MH_MAXP + 1]; MhProb p[MH_MAXP]; };

// One work-group = one 32x32 output tile; its four waves split K (a quarter each, whole 16-deep MFMA steps) and their partial
// tiles are summed through LDS in wave order (fixed: run-to-run identical).  Rounds 1-2 gave every wave a tile of its own and
// the whole K: the launch lasted as long as one wave's chain of dependent load batches (7 for K = 864, each a round trip to
// HBM for operands written a launch ago), ~27 us for 3 GFLOP.
__global__ void __launch_bounds__(256) k_mgemm_h16(const MhGroup *__restrict__ g, int total_tiles)
{
    ORN_PRIO_HIGH();
    __shared__ float part[4][16][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x;
    int pi = 0;
    while (pi + 1 < g->n && tile >= g->tile_start[pi + 1]) ++pi;
    const MhProb &p = g->p[pi];
    const int local = tile - g->tile_start[pi];
    const int b = local / p.tiles, r = local - b * p.tiles;
    const int tm = r / p.tiles_n, tn = r - tm * p.tiles_n;
    // this wave's K range: steps [s0, s1) of 16
    const int nsteps = p.K / 16, per = (nsteps + 3) / 4;
    const int s0 = min(wave * per, nsteps), s1 = min(s0 + per, nsteps);
    const mh16 *ap = p.A + (size_t)b * p.ba + (size_t)(tm * 32 + l31) * p.lda + hh * 8 + s0 * 16;
    const mh16 *bp = p.B + (size_t)b * p.bb + (size_t)(tn * 32 + l31) * p.ldb + hh * 8 + s0 * 16;
    mf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int K = (s1 - s0) * 16;
    int k = 0;
    // MH_KB k-steps (128 deep) per iteration: the 16 fragment loads of the next iteration are in flight under this
    // iteration's MFMAs
    mh16x8 fa[MH_KB], fb[MH_KB];
    constexpr int KD = MH_KB * 16;
    if (K >= KD) {
#pragma unroll
        for (int j = 0; j < MH_KB; ++j) { fa[j] = *reinterpret_cast<const mh16x8 *>(ap + j * 16); fb[j] = *reinterpret_cast<const mh16x8 *>(bp + j * 16); }
        for (k = KD; k + KD <= K; k += KD) {
            mh16x8 na[MH_KB], nb[MH_KB];
#pragma unroll
            for (int j = 0; j < MH_KB; ++j) { na[j] = *reinterpret_cast<const mh16x8 *>(ap + k + j * 16); nb[j] = *reinterpret_cast<const mh16x8 *>(bp + k + j * 16); }
#pragma unroll
            for (int j = 0; j < MH_KB; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j], fb[j], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < MH_KB; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
        }
        // the tail's loads go out before the last full batch is consumed
        mh16x8 ta[MH_KB], tb[MH_KB];
        const int nt = (K - k) / 16;
#pragma unroll
        for (int j = 0; j < MH_KB; ++j)
            if (j < nt) { ta[j] = *reinterpret_cast<const mh16x8 *>(ap + k + j * 16); tb[j] = *reinterpret_cast<const mh16x8 *>(bp + k + j * 16); }
#pragma unroll
        for (int j = 0; j < MH_KB; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j], fb[j], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < MH_KB; ++j)
            if (j < nt) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[j], tb[j], acc, 0, 0, 0);
        k = K;
    }
    if (k < K) {                 // K < one batch: everything in one go
        mh16x8 ta[MH_KB], tb[MH_KB];
        const int nt = (K - k) / 16;
#pragma unroll
        for (int j = 0; j < MH_KB; ++j)
            if (j < nt) { ta[j] = *reinterpret_cast<const mh16x8 *>(ap + k + j * 16); tb[j] = *reinterpret_cast<const mh16x8 *>(bp + k + j * 16); }
#pragma unroll
        for (int j = 0; j < MH_KB; ++j)
            if (j < nt) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[j], tb[j], acc, 0, 0, 0);
    }
    // partial tiles -> LDS; wave w then owns accumulator registers 4w .. 4w+3 (rows 8w + {0..3} + 4 hh of the tile)
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave][i][lane] = acc[i];
    __syncthreads();
    float sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = 4 * wave + i;
        sum[i] = ((part[0][reg][lane] + part[1][reg][lane]) + part[2][reg][lane]) + part[3][reg][lane];
    }
    // D: column n = tn*32 + l31 (B row), rows m = tm*32 + (reg&3) + 8*(reg>>2) + 4*hh
    const int n = tn * 32 + l31;
    if (n >= p.N) return;
    if (p.mode == 0) {
        float *C = p.C + (size_t)b * p.bc + (size_t)n * p.scn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = 4 * wave + i, m = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (m < p.M) C[(size_t)m * p.scm] = sum[i] * p.so;
        }
    } else {
        const int c = n / 9, ij = n - c * 9;
        mh16 *t = p.dtt + (size_t)ij * p.st_t + c;                  // dTt[ij][m][c]
        mh16 *u = p.dtc + (size_t)ij * p.st_c + (size_t)c * p.ldc;   // dTc[ij][c][m]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = 4 * wave + i, m = tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (m < p.M) {
                const mh16 v = (mh16)sum[i];                         // = dT * 2^14 (B = G carried the scale)
                t[(size_t)m * p.ldt] = v;
                u[m] = v;
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct OrnMergeH16 {
    MhPackAll pack;
    int pack_blocks_grad, pack_blocks_par, pack_blocks_t;
    int tiles[2];                 // total 32x32 tiles of the two GEMM launches
};

static void mh_add(MhGroup &g, MhProb p)
{
    p.tiles_n = (p.N + 31) / 32;
    p.tiles = ((p.M + 31) / 32) * p.tiles_n;
    g.p[g.n] = p;
    g.tile_start[g.n + 1] = g.tile_start[g.n] + p.tiles * p.nbatch;
    ++g.n;
}

size_t orn_merge_h16_table_bytes() { return orn_align(2 * sizeof(MhGroup)); }
size_t orn_merge_h16_host_bytes() { return sizeof(OrnMergeH16); }

// bufs[i]: the layer's half workspace (orn_merge_h16_layer_halfs), zero-filled once by the caller and never cleared
// again (the padding must stay zero).  host: caller-owned storage of orn_merge_h16_host_bytes().
int orn_merge_h16_build(void *dev_tables, void *host, int n_layers, const OrnMergeLayer *L, void *const *bufs, OrnScaleState *sc)
{
    ORN_REQUIRE(2 * n_layers <= MH_MAXP && n_layers <= ORN_MAX_LAYERS, "merge h16: too many layers");
    OrnMergeH16 *H = new (host) OrnMergeH16();
    MhGroup *G = new MhGroup[2]();
    H->pack.n = n_layers;
    H->pack.sc = sc;
    H->pack.grad_start[0] = 0;
    H->pack.par_start[0] = 0;
    H->pack.tt_start[0] = 0;
    for (int i = 0; i < n_layers; ++i) {
        const OrnMergeLayer &l = L[i];
        const int C = l.C, O = l.O, E = 9 * C, K2 = 2 * C;
        const MhSizes s = mh_sizes(C, O);
        mh16 *base = (mh16 *)bufs[i];
        auto take = [&](size_t n) { mh16 *p = base; base += (n + 127) / 128 * 128; return p; };
        mh16 *gh = take(s.gh), *gt = take(s.gt), *th = take(s.th), *w3t = take(s.w3t), *w1h = take(s.w1h), *w2p = take(s.w2p);
        mh16 *dtt = take(s.dtt), *dtc = take(s.dtc);
        H->pack.l[i] = MhPackLayer{C, O, l.g, l.T, l.w1, l.w2, l.w3, gh, gt, th, w3t, w1h, w2p};
        const int jobs[MH_JOBS] = {orn_cdiv((long)O * E, 256 * MH_CPT), orn_cdiv((long)O * E, 256 * MH_CPT), orn_cdiv((long)K2 * C, 256 * MH_CPT),
                                   orn_cdiv(O, 64) * orn_cdiv(E, 64), orn_cdiv(O, 64) * orn_cdiv(O, 64), orn_cdiv(O, 64) * orn_cdiv(K2 * 9, 64)};
        H->pack.grad_start[2 * i + 1] = H->pack.grad_start[2 * i] + jobs[0];
        H->pack.grad_start[2 * i + 2] = H->pack.grad_start[2 * i + 1] + jobs[3];
        const int pj[3] = {jobs[2], jobs[4], jobs[5]};
        for (int j = 0; j < 3; ++j) H->pack.par_start[3 * i + j + 1] = H->pack.par_start[3 * i + j] + pj[j];
        H->pack.tt_start[i + 1] = H->pack.tt_start[i] + jobs[1];
        MhProb p;
        // dW3[o][m] = (1/GS) sum_e Gh[o][e] Th[m][e]
        p = MhProb{};
        p.A = gh; p.lda = r16(E); p.B = th; p.ldb = r16(E); p.M = O; p.N = O; p.K = r16(E); p.nbatch = 1;
        p.mode = 0; p.C = l.dw3; p.scm = O; p.scn = 1; p.so = 1.0f / MH_GS;
        mh_add(G[0], p);
        // dT[m][(c,ij)] * GS = sum_o W3T[m][o] GT[e][o]
        p = MhProb{};
        p.A = w3t; p.lda = r16(O); p.B = gt; p.ldb = r16(O); p.M = O; p.N = E; p.K = r16(O); p.nbatch = 1;
        p.mode = 1; p.dtt = dtt; p.dtc = dtc; p.Cch = C; p.ldt = r16(C); p.ldc = r16(O);
        p.st_t = (long)r32(O) * r16(C); p.st_c = (long)r32(C) * r16(O);
        mh_add(G[0], p);
        // dW2t[ij][m][k] = (1/GS) sum_c dTt[ij][m][c] W1h[k][c]   (the tail kernel interleaves it into dW2[m][k][ij])
        p = MhProb{};
        p.A = dtt; p.lda = r16(C); p.ba = (long)r32(O) * r16(C); p.B = w1h; p.ldb = r16(C); p.bb = 0;
        p.M = O; p.N = K2; p.K = r16(C); p.nbatch = 9;
        p.mode = 0; p.C = l.dw2t; p.scm = K2; p.scn = 1; p.bc = (long)O * K2; p.so = 1.0f / MH_GS;     // tap-major: coalesced stores
        mh_add(G[1], p);
        // dW1p[ij][k][c] = (1/GS) sum_m W2p[ij][k][m] dTc[ij][c][m]
        p = MhProb{};
        p.A = w2p; p.lda = r16(O); p.ba = (long)r32(K2) * r16(O); p.B = dtc; p.ldb = r16(O); p.bb = (long)r32(C) * r16(O);
        p.M = K2; p.N = C; p.K = r16(O); p.nbatch = 9;
        p.mode = 0; p.C = l.dw1p; p.scm = C; p.scn = 1; p.bc = (long)K2 * C; p.so = 1.0f / MH_GS;
        mh_add(G[1], p);
    }
    H->pack_blocks_grad = H->pack.grad_start[2 * n_layers];
    H->pack_blocks_par = H->pack.par_start[3 * n_layers];
    H->pack_blocks_t = H->pack.tt_start[n_layers];
    H->tiles[0] = G[0].tile_start[G[0].n];
    H->tiles[1] = G[1].tile_start[G[1].n];
    hipError_t e = hipMemcpy(dev_tables, G, 2 * sizeof(MhGroup), hipMemcpyHostToDevice);
    delete[] G;
    if (e != hipSuccess) { orn_set_error("merge h16: hipMemcpy failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

// the forward riders: the pack table and the block counts of its parameter-side jobs (k_merge_prep) and of T -> Th
const void *orn_merge_h16_pack(const void *host, int *par_blocks, int *t_blocks)
{
    const OrnMergeH16 *H = (const OrnMergeH16 *)host;
    *par_blocks = H->pack_blocks_par;
    *t_blocks = H->pack_blocks_t;
    return &H->pack;
}

int orn_launch_merge_h16_pack_jobs(const void *host, int which, hipStream_t st)
{
    const OrnMergeH16 *H = (const OrnMergeH16 *)host;
    const int blocks = which == MH_TAB_PAR ? H->pack_blocks_par : H->pack_blocks_t;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(k_merge_pack, dim3(blocks), dim3(256), 0, st, H->pack, which);
    ORN_LAUNCH_CHECK("merge_pack(jobs)");
    return 0;
}

// (parameter-side pack: forward), gradient-side pack, then {dW3, dT}, then {dW2, dW1 partials}; the slices / dW1 sum stay with
// orn_launch_merge_bwd_tail_all
int orn_launch_merge_h16_bwd(const void *dev_tables, const void *host, hipStream_t st, OrnScaleState *sc)
{
    const OrnMergeH16 *H = (const OrnMergeH16 *)host;
    MhPackAll pack = H->pack;
    if (sc) pack.sc = sc;                               // the flag of the step that is being launched
    hipLaunchKernelGGL(k_merge_pack, dim3(H->pack_blocks_grad), dim3(256), 0, st, pack, (int)MH_TAB_GRAD);     // G -> Gh, GT
    ORN_LAUNCH_CHECK("merge_pack");
    const MhGroup *g = (const MhGroup *)dev_tables;
    hipLaunchKernelGGL(k_mgemm_h16, dim3(H->tiles[0]), dim3(256), 0, st, g, H->tiles[0]);
    ORN_LAUNCH_CHECK("mgemm_h16(dW3,dT)");
    hipLaunchKernelGGL(k_mgemm_h16, dim3(H->tiles[1]), dim3(256), 0, st, g + 1, H->tiles[1]);
    ORN_LAUNCH_CHECK("mgemm_h16(dW2,dW1)");
    return 0;
}
