// Half (IEEE fp16), K-contiguous operand copies for the merge backward of the 16-bit engine modes (orn_merge_h16.hip).
// Where the jobs run (round 3): the ones that read only parameters -- W1 -> W1h, W3 -> W3T, W2 -> W2p -- trail the first
// launch of the forward merge (k_merge_prep, orn_merge.hip: a lean kernel, five work-groups per CU); T -> Th trails the first
// block's forward launch (orn_stage0.hip) or, without that launch, the merge's S launch; the two that read the gradient G
// stay a launch of their own in the backward.  (Rounds 1-2 put all four forward jobs behind the S launch: its GEMM tiles
// hold 229 VGPRs, every work-group of a launch is charged the same, so ~1,500 short riders queued two per CU: +19 us.)
#pragma once
#include "orn_common.h"

typedef _Float16 mh16;
#define MH_GS 16384.0f            // gradient operands are multiplied by 2^14 when rounded to half (|dWf| ~ 1e-6)
__host__ __device__ static inline int r16(int x) { return (x + 15) / 16 * 16; }
__host__ __device__ static inline int r32(int x) { return (x + 31) / 32 * 32; }

struct MhPackLayer {
    int C, O;
    const float *g, *T, *w1, *w2, *w3;
    mh16 *gh, *gt, *th, *w3t, *w1h, *w2p;
};
#define MH_JOBS 6
#define MH_CPT 8
// job 0: G -> Gh (x 2^14);  1: T -> Th;  2: W1 -> W1h   (row-major copies into padded rows, MH_CPT elements per thread)
// job 3: G -> GT (x 2^14);  4: W3 -> W3T;  5: W2 [m][(k,ij)] -> W2p [ij][k][m]   (64x64 tile transposes through LDS:
//        coalesced fp32 reads along the source rows, coalesced half writes along the destination rows)
// Three 1-D block tables: `grad` walks jobs {0, 3} of every layer (backward launch), `par` jobs {2, 4, 5} (parameters only),
// `tt` job 1 (T -> Th).
enum { MH_TAB_GRAD = 0, MH_TAB_PAR = 1, MH_TAB_T = 2 };
struct MhPackAll {
    int n;
    int grad_start[2 * ORN_MAX_LAYERS + 1];
    int par_start[3 * ORN_MAX_LAYERS + 1];
    int tt_start[ORN_MAX_LAYERS + 1];
    MhPackLayer l[ORN_MAX_LAYERS];
    OrnScaleState *sc;   // optional: an overflow of the scaled half copy of G raises its flag (the step is then skipped)
};

// one work-group of the pack (any whole number of waves); tile: 64 x 65 floats of LDS
__device__ __forceinline__ void mh_pack_block(const MhPackAll &a, int layer, int job, int blk, float (*tile)[65])
{
    const int nw = blockDim.x >> 6;
    const MhPackLayer &l = a.l[layer];
    const int C = l.C, O = l.O, E = 9 * C, K2 = 2 * C;
    if (job < 3) {
        // row-major copies into rows padded to 16 halfs.  The job's work-groups (their count is fixed by the host tables:
        // MH_CPT elements per thread) split the ROWS; a wave takes whole rows, its lanes stride the row with 14 loads in
        // flight.  (The first form took one element per thread and turned its flat index into (row, column) with a 64-bit
        // division: ~150 instructions per element -- these riders outlasted the GEMM tiles of their launch by 19 us.)
        const float *src = job == 0 ? l.g : (job == 1 ? l.T : l.w1);
        mh16 *dst = job == 0 ? l.gh : (job == 1 ? l.th : l.w1h);
        const int Q = job == 2 ? C : E, R = job == 2 ? K2 : O;
        const float sc = job == 0 ? MH_GS : 1.0f;
        const int ldd = r16(Q);
        const int nblk = (int)(((size_t)R * Q + 256 * MH_CPT - 1) / (256 * MH_CPT));
        const int rpb = (R + nblk - 1) / nblk;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int r_end = min(R, (blk + 1) * rpb);
        bool bad = false;
        for (int r = blk * rpb + wave; r < r_end; r += nw) {
            const float *sr = src + (size_t)r * Q;
            mh16 *dr = dst + (size_t)r * ldd;
            for (int q0 = lane; q0 < Q; q0 += 14 * 64) {
                float v[14];
#pragma unroll
                for (int u = 0; u < 14; ++u) v[u] = sr[min(q0 + 64 * u, Q - 1)];
#pragma unroll
                for (int u = 0; u < 14; ++u)
                    if (q0 + 64 * u < Q) {
                        const mh16 hv = (mh16)(v[u] * sc);
                        dr[q0 + 64 * u] = hv;
                        if (job == 0) bad |= !(fabsf((float)hv) <= 3.0e38f);      // every element of G passes here once
                    }
            }
        }
        if (job == 0 && bad) orn_flag_nonfinite(a.sc, __builtin_nanf(""));
        return;
    }
    // source matrix [R][Q] row-major -> destination rows q (remapped), columns r
    const float *src; mh16 *dst; int R, Q, ldd; float sc = 1.0f;
    if (job == 3)      { src = l.g;  dst = l.gt;  R = O; Q = E;      ldd = r16(O); sc = MH_GS; }
    else if (job == 4) { src = l.w3; dst = l.w3t; R = O; Q = O;      ldd = r16(O); }
    else               { src = l.w2; dst = l.w2p; R = O; Q = K2 * 9; ldd = r16(O); }
    const int tq = (Q + 63) / 64;
    const int r0 = (blk / tq) * 64, q0 = (blk % tq) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (nw == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + ty + 4 * i, q = q0 + tx;
            tile[ty + 4 * i][tx] = (r < R && q < Q) ? src[(size_t)r * Q + q] * sc : 0.f;
        }
    } else {
        for (int rr = ty; rr < 64; rr += nw) {
            const int r = r0 + rr, q = q0 + tx;
            tile[rr][tx] = (r < R && q < Q) ? src[(size_t)r * Q + q] * sc : 0.f;
        }
    }
    __syncthreads();
    for (int qq = ty; qq < 64; qq += nw) {
        const int q = q0 + qq, r = r0 + tx;
        if (q < Q && r < R) {
            size_t row = q;
            if (job == 5) { const int k = q / 9, ij = q - k * 9; row = (size_t)ij * r32(K2) + k; }
            dst[row * ldd + r] = (mh16)tile[tx][qq];
        }
    }
}

// block id of one of the three tables -> (layer, job); returns the block index inside the job
__device__ __forceinline__ int mh_pack_decode(const MhPackAll &a, int table, int bid, int &layer, int &job)
{
    const int per = table == MH_TAB_GRAD ? 2 : (table == MH_TAB_PAR ? 3 : 1);
    const int *start = table == MH_TAB_GRAD ? a.grad_start : (table == MH_TAB_PAR ? a.par_start : a.tt_start);
    int pj = 0;
    while (pj + 1 < per * a.n && bid >= start[pj + 1]) ++pj;
    layer = pj / per;
    const int j = pj - layer * per;
    job = table == MH_TAB_GRAD ? (j == 0 ? 0 : 3) : (table == MH_TAB_PAR ? (j == 0 ? 2 : (j == 1 ? 4 : 5)) : 1);
    return bid - start[pj];
}
