// Half (IEEE fp16), K-contiguous operand copies for the merge backward of the 16-bit engine modes (orn_merge_h16.hip).
// The jobs that read only parameters and the forward product T -- T -> Th, W1 -> W1h, W3 -> W3T, W2 -> W2p -- ride along the
// forward merge's S launch as trailing work-groups (orn_merge.hip; that launch is a handful of long latency chains and leaves
// most CUs idle); the two that read the gradient G stay a (smaller) launch of their own in the backward.
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
// Two 1-D block tables: `grad` walks jobs {0, 3} of every layer (backward launch), `fwd` jobs {1, 2, 4, 5} (forward).
struct MhPackAll {
    int n;
    int grad_start[2 * ORN_MAX_LAYERS + 1];
    int fwd_start[4 * ORN_MAX_LAYERS + 1];
    MhPackLayer l[ORN_MAX_LAYERS];
    OrnScaleState *sc;   // optional: an overflow of the scaled half copy of G raises its flag (the step is then skipped)
};

// one work-group (256 threads) of the pack; tile: 64 x 65 floats of LDS
__device__ __forceinline__ void mh_pack_block(const MhPackAll &a, int layer, int job, int blk, float (*tile)[65])
{
    const MhPackLayer &l = a.l[layer];
    const int C = l.C, O = l.O, E = 9 * C, K2 = 2 * C;
    if (job < 3) {
        // MH_CPT elements per thread (the dispatcher, not HBM, bounds a launch of ten thousand one-element work-groups)
        const float *src = job == 0 ? l.g : (job == 1 ? l.T : l.w1);
        mh16 *dst = job == 0 ? l.gh : (job == 1 ? l.th : l.w1h);
        const int Q = job == 2 ? C : E;
        const size_t n = (size_t)(job == 2 ? K2 : O) * Q;
        const float sc = job == 0 ? MH_GS : 1.0f;
        const int ldd = r16(Q);
#pragma unroll
        for (int i = 0; i < MH_CPT; ++i) {
            const size_t idx = ((size_t)blk * MH_CPT + i) * 256 + threadIdx.x;
            if (idx < n) {
                const int r = (int)(idx / Q), q = (int)(idx - (size_t)r * Q);
                const mh16 hv = (mh16)(src[idx] * sc);
                dst[(size_t)r * ldd + q] = hv;
                if (job == 0) orn_flag_nonfinite(a.sc, (float)hv);      // every element of G passes here once
            }
        }
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
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, q = q0 + tx;
        tile[ty + 4 * i][tx] = (r < R && q < Q) ? src[(size_t)r * Q + q] * sc : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = q0 + ty + 4 * i, r = r0 + tx;
        if (q < Q && r < R) {
            size_t row = q;
            if (job == 5) { const int k = q / 9, ij = q - k * 9; row = (size_t)ij * r32(K2) + k; }
            dst[row * ldd + r] = (mh16)tile[tx][ty + 4 * i];
        }
    }
}

// block id of one of the two tables -> (layer, job); returns the block index inside the job
__device__ __forceinline__ int mh_pack_decode(const MhPackAll &a, bool fwd, int bid, int &layer, int &job)
{
    const int per = fwd ? 4 : 2;
    const int *start = fwd ? a.fwd_start : a.grad_start;
    int pj = 0;
    while (pj + 1 < per * a.n && bid >= start[pj + 1]) ++pj;
    layer = pj / per;
    const int j = pj - layer * per;
    job = fwd ? (j == 0 ? 1 : (j == 1 ? 2 : (j == 2 ? 4 : 5))) : (j == 0 ? 0 : 3);
    return bid - start[pj];
}
