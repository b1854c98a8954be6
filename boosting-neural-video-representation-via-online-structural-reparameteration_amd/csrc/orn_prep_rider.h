// 16-bit operand copies of the merged ERB kernels (what the conv kernels DMA: wb [9][O'][Cp] forward, wd [9][Cp][O'] with
// flipped taps for the dgrad, biasp [O'], o' = (o % s2) * Cn + o / s2) as work-groups that RIDE on another launch.
// Rounds 1-2 wrote these from the epilogue of the merge's S product: 32 scattered 2-byte stores per thread and tile, ~16 us
// on the critical path of the step's first launches.  Here one work-group takes 16 consecutive o' of one layer: its 16 rows
// of Wf (contiguous 3.4 KB each) go through LDS as halves, wb leaves as whole 192-byte rows and wd as 32-byte segments (32
// rows per work-group made 64-byte segments but half as many work-groups: a CU moves ~25 GB/s, and these were bound by it).  The
// work-groups trail the first block's forward launch (orn_stage0.hip: 41 work-groups on a 256-CU chip), whose successor
// (the first 16-bit conv) is the first consumer.
#pragma once
#include "orn_common.h"

#define ORN_PREP_ROWS 16
struct OrnPrepRider {
    int n;                                   // layers
    int blk_start[ORN_MAX_LAYERS + 1];       // rider block index -> layer
    struct { const float *wf, *bf; int O, C, Cp, Cn, s2; void *wb, *wd; float *biasp; } l[ORN_MAX_LAYERS];
};
static inline size_t orn_prep_rider_lds_bytes(int Cmax) { return (size_t)ORN_PREP_ROWS * (Cmax * 9 + 2) * 2; }

// blk: rider block index (0 .. blk_start[n]); lds: >= orn_prep_rider_lds_bytes(C) bytes, 4-byte aligned; any block size
template <typename H16>
__device__ __forceinline__ void orn_prep_rider_block(const OrnPrepRider &r, int blk, void *lds)
{
    int li = 0;
    while (li + 1 < r.n && blk >= r.blk_start[li + 1]) ++li;
    const auto &l = r.l[li];
    const int t = threadIdx.x, nt = blockDim.x;
    const int E = l.C * 9, LD = E + 2;                     // row stride in halves: an odd number of dwords (conflict-free column reads)
    const int op0 = (blk - r.blk_start[li]) * ORN_PREP_ROWS;
    H16 *tile = reinterpret_cast<H16 *>(lds);
    const int rows = min(ORN_PREP_ROWS, l.O - op0);
    const int wave = t >> 6, lane = t & 63, nw = nt >> 6;
    // (index arithmetic without runtime divisions -- ~40 VALU instructions each: a first version with idx / E, idx / C per
    // element made these 48 work-groups the longest part of their launch, 50 us)
    // rows in: wave w takes rows w, w + nw, ...; lanes stride the row.  Every load of a wave's rows is in flight before its
    // first LDS store (a wave has 3-4 rows: row by row it waited out four round trips to the MALL, and the riders outlasted
    // the launch they ride on by 14 us)
    constexpr int RB = 4, LB = 14;                          // rows per batch, loads per lane and row (E <= 896 in one round)
    for (int rb = wave; rb < rows; rb += RB * nw) {
        float v[RB][LB];
#pragma unroll
        for (int a = 0; a < RB; ++a) {
            const int rr = min(rb + a * nw, rows - 1);
            const int op = op0 + rr, q = op / l.Cn, o = (op - q * l.Cn) * l.s2 + q;
            const float *src = l.wf + (size_t)o * E;
#pragma unroll
            for (int u = 0; u < LB; ++u) v[a][u] = src[min(lane + 64 * u, E - 1)];
        }
#pragma unroll
        for (int a = 0; a < RB; ++a) {
            const int rr = rb + a * nw;
            if (rr < rows) {
                H16 *dst = tile + rr * LD;
#pragma unroll
                for (int u = 0; u < LB; ++u)
                    if (lane + 64 * u < E) dst[lane + 64 * u] = (H16)v[a][u];
                const int op = op0 + rr, q = op / l.Cn, o = (op - q * l.Cn) * l.s2 + q;
                const float *src = l.wf + (size_t)o * E;
                for (int e = lane + 64 * LB; e < E; e += 64) dst[e] = (H16)src[e];      // (rows longer than 896 floats)
                if (lane == 0) l.biasp[op] = l.bf[o];
            }
        }
    }
    __syncthreads();
    H16 *wb = reinterpret_cast<H16 *>(l.wb), *wd = reinterpret_cast<H16 *>(l.wd);
    // wb [ij][o'][c]: c fastest -- wave w takes (tap, row) pairs w, w + nw, ...; a pair is one run of C halves.  Eight pairs
    // per batch: their LDS reads go out together, then their stores.
    constexpr int PB = 8;
    for (int pb = wave; pb < 9 * rows; pb += PB * nw) {
        for (int c0 = lane; c0 < l.C; c0 += 64) {
            H16 x[PB];
#pragma unroll
            for (int a = 0; a < PB; ++a) {
                const int pr = min(pb + a * nw, 9 * rows - 1), ij = pr / rows, rr = pr - ij * rows;
                x[a] = tile[rr * LD + c0 * 9 + ij];
            }
#pragma unroll
            for (int a = 0; a < PB; ++a) {
                const int pr = pb + a * nw;
                if (pr < 9 * rows) {
                    const int ij = pr / rows, rr = pr - ij * rows;
                    wb[((size_t)ij * l.O + op0 + rr) * l.Cp + c0] = x[a];
                }
            }
        }
    }
    // wd [8 - ij][c][o']: o' fastest -- a wave-instruction covers four c (lane >> 4) x 16 rows: four 32-byte segments
    const int rr = lane & (ORN_PREP_ROWS - 1), ch = lane / ORN_PREP_ROWS, hc = (l.C + 64 / ORN_PREP_ROWS - 1) / (64 / ORN_PREP_ROWS);
    for (int pb = wave; pb < 9 * hc; pb += PB * nw) {
        H16 x[PB];
#pragma unroll
        for (int a = 0; a < PB; ++a) {
            const int pr = min(pb + a * nw, 9 * hc - 1), ij = pr / hc, c = min((64 / ORN_PREP_ROWS) * (pr - ij * hc) + ch, l.C - 1);
            x[a] = tile[min(rr, rows - 1) * LD + c * 9 + ij];
        }
#pragma unroll
        for (int a = 0; a < PB; ++a) {
            const int pr = pb + a * nw;
            if (pr < 9 * hc) {
                const int ij = pr / hc, c = (64 / ORN_PREP_ROWS) * (pr - ij * hc) + ch;
                if (c < l.C && rr < rows) wd[((size_t)(8 - ij) * l.Cp + c) * l.O + op0 + rr] = x[a];
            }
        }
    }
}
