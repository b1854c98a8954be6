/* CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/cpu_ref.py header): the ERB online
 * re-parameterisation with the *specified* summation order, in plain C with true fmaf().
 *
 * Restates /root/reference/model.py:450-516 (get_equivalent_kernel_bias,
 * _fuse_1x3_3x1_branch, _fuse_1x1_3x3_1x1_branch):
 *
 *   T[m,c,i,j]  = sum_k  W2[m,k,i,j] * W1[k,c]            (model.py:510, F.conv2d with W1 as 1x1 kernel)
 *   S[o,c,i,j]  = sum_m  W3[o,m]     * T[m,c,i,j]         (model.py:513-515, matmul over the 3x3 grid)
 *   Wf          = (W3x3 + (P(w1x3) + P(w3x1))) + S        (model.py:475, 495)
 *   bf          = b3x3 + (b1x3 + b3x1)                    (model.py:476, 496)
 *
 * Both contractions are evaluated as a single k-ordered fused-multiply-add chain starting from
 * +0.0f:  acc = fmaf(a_k, b_k, acc), k = 0,1,2,...  This is exactly what one accumulator of a
 * gfx950 v_mfma_f32_*_f32 chain (or a VALU v_fma_f32 loop) computes, so the HIP merge kernel is
 * required to match this file BIT FOR BIT.  Against the reference's own ATen result (opaque
 * oneDNN/MKL order) the match is to ~1 ulp, pinned by tests/golden/merge_*.npz.
 *
 * Build: make -C oracle   ->  oracle/_build/liborn_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

/* T: [O, C, 9] out; W2: [O, 2C, 9]; W1: [2C, C] */
void orn_oracle_merge_T(const float *w1, const float *w2, int C, int O, float *T)
{
    const int K2 = 2 * C;
    for (int m = 0; m < O; ++m)
        for (int c = 0; c < C; ++c)
            for (int ij = 0; ij < 9; ++ij) {
                float acc = 0.0f;
                for (int k = 0; k < K2; ++k)
                    acc = fmaf(w2[((size_t)m * K2 + k) * 9 + ij], w1[(size_t)k * C + c], acc);
                T[((size_t)m * C + c) * 9 + ij] = acc;
            }
}

/* Wf: [O, C, 9] out, bf: [O] out. */
void orn_oracle_merge_fwd(const float *w3x3, const float *b3x3, const float *w3x1, const float *b3x1,
                          const float *w1x3, const float *b1x3, const float *w1, const float *w2,
                          const float *w3, int C, int O, float *T, float *wf, float *bf)
{
    orn_oracle_merge_T(w1, w2, C, O, T);
    const size_t n = (size_t)C * 9;
    for (int o = 0; o < O; ++o) {
        for (size_t e = 0; e < n; ++e) {
            float acc = 0.0f;
            for (int m = 0; m < O; ++m)
                acc = fmaf(w3[(size_t)o * O + m], T[(size_t)m * n + e], acc);
            const int c = (int)(e / 9), ij = (int)(e % 9), i = ij / 3, j = ij % 3;
            /* P(w1x3): row i==1 holds w1x3[o,c,0,j]; P(w3x1): column j==1 holds w3x1[o,c,i,0] */
            const float p13 = (i == 1) ? w1x3[((size_t)o * C + c) * 3 + j] : 0.0f;
            const float p31 = (j == 1) ? w3x1[((size_t)o * C + c) * 3 + i] : 0.0f;
            wf[(size_t)o * n + e] = (w3x3[(size_t)o * n + e] + (p13 + p31)) + acc;
        }
        bf[o] = b3x3[o] + (b1x3[o] + b3x1[o]);
    }
}

/* Backward of the merge, same closed forms as SURVEY 8a row A3; each contraction is one k-ordered
 * fmaf chain from +0.0f over the index order written in the comment (the HIP kernels follow it).
 *   dW3[o,m]      = sum_{e=(c,i,j)} G[o,e] * T[m,e]           e ascending
 *   dT[m,e]       = sum_o W3[o,m] * G[o,e]                    o ascending
 *   dW2[m,k,ij]   = sum_c dT[m,c,ij] * W1[k,c]                c ascending
 *   dW1[k,c]      = sum_{m,ij} W2[m,k,ij] * dT[m,c,ij]        (m,ij) ascending, ij fastest
 */
void orn_oracle_merge_bwd(const float *g, const float *w1, const float *w2, const float *w3, const float *T,
                          int C, int O, float *dT, float *dw1, float *dw2, float *dw3)
{
    const int K2 = 2 * C;
    const size_t n = (size_t)C * 9;
    for (int o = 0; o < O; ++o)
        for (int m = 0; m < O; ++m) {
            float acc = 0.0f;
            for (size_t e = 0; e < n; ++e)
                acc = fmaf(g[(size_t)o * n + e], T[(size_t)m * n + e], acc);
            dw3[(size_t)o * O + m] = acc;
        }
    for (int m = 0; m < O; ++m)
        for (size_t e = 0; e < n; ++e) {
            float acc = 0.0f;
            for (int o = 0; o < O; ++o)
                acc = fmaf(w3[(size_t)o * O + m], g[(size_t)o * n + e], acc);
            dT[(size_t)m * n + e] = acc;
        }
    for (int m = 0; m < O; ++m)
        for (int k = 0; k < K2; ++k)
            for (int ij = 0; ij < 9; ++ij) {
                float acc = 0.0f;
                for (int c = 0; c < C; ++c)
                    acc = fmaf(dT[((size_t)m * C + c) * 9 + ij], w1[(size_t)k * C + c], acc);
                dw2[((size_t)m * K2 + k) * 9 + ij] = acc;
            }
    for (int k = 0; k < K2; ++k)
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
            for (int m = 0; m < O; ++m)
                for (int ij = 0; ij < 9; ++ij)
                    acc = fmaf(w2[((size_t)m * K2 + k) * 9 + ij], dT[((size_t)m * C + c) * 9 + ij], acc);
            dw1[(size_t)k * C + c] = acc;
        }
}
