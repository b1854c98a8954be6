// Internal (non-exported) launchers shared between the op files and the engine.
#pragma once
#include "orn_common.h"

// orn_elementwise.hip
int orn_launch_linear_silu(const float *x, const int *row_idx, size_t row_stride, const float *w, const float *b,
                           int B, int K, int N, float *pre, float *y, hipStream_t st);
int orn_launch_stem_bwd(const float *embed, const int *row_idx, size_t row_stride, const float *w1, const float *pre1,
                        const float *h1, const float *pre2, const float *dh2, int B, int E, int Hd, int Nout,
                        float *dw0, float *db0, float *dw1, float *db1, float *ws, hipStream_t st, int dh2_nslab = 1,
                        OrnStemW0Job *defer_w0 = nullptr,    // defer_w0 (B == 1): the last kernel is returned as a job instead of launched
                        OrnStemL2Job *defer_l2 = nullptr);   // defer_l2 (with defer_w0): so is the first one (it then has to run in an EARLIER launch than w0)
// dh2_nslab > 1 (B == 1): dh2 holds that many partial rows of Nout floats, summed in fixed order by the first kernel
size_t orn_stem_bwd_ws_floats(int B, int Hd, int Nout);
int orn_launch_head_fwd(const float *a, const float *w, const float *b, int B, int C, size_t HW, int sigmoid,
                        float *out, hipStream_t st, bool z_input = false);   // z_input: `a` holds the pre-activation, SiLU is applied here
int orn_launch_head_bwd(const float *a, const float *w, const float *out, const float *dout, int B, int C, size_t HW,
                        int sigmoid, float *da, float *dw, float *db, float *ws, hipStream_t st);
int orn_launch_adam(float *p, const float *g, float *m, float *v, size_t n, double lr, int step, const OrnStepCur *sp,
                    double beta1, double beta2, double eps, float inv_gscale, hipStream_t st, const float *gmask = nullptr,
                    OrnScaleState *sc = nullptr,    // sc: skip the update while its flag is up
                    OrnScaleState *sc_master = nullptr,    // (engine: the entry that counts skipped steps; default sc itself)
                    OrnScaleState *mirror = nullptr,       // (engine, deferred last block) the skip decision is also stored into mirror->flag
                    bool count_skip = true);               // false: a skipped launch does not count (the step's other Adam launch does)

// orn_stage0.hip: the fp32 block below the first 16-bit one (tiny stem image), forward / backward as one launch each
bool orn_stage0_supported(int C, int O, int H, int W, int s);
int orn_stage0_slabs(int O, int s);
struct OrnPrepLayer;      // (below) 16-bit operand copies of later blocks' kernels, written by rider work-groups of this launch
int orn_launch_stage0_fwd(const float *x, const float *wf, const float *bf, int C, int O, int H, int W, int s, float *z,
                          void *xpad_next, int Cp, int precision, hipStream_t st, int n_prep = 0, const OrnPrepLayer *prep = nullptr,
                          const void *pack = nullptr, int pack_t_blocks = 0);   // pack: the merge backward's T -> Th copies as further riders
int orn_launch_stage0_bwd(const float *x, const float *wf, const float *z, const float *dxn, int nslab, int Cp, float inv_gs, int C,
                          int O, int H, int W, int s, float *slabs, float *dx, float *dwf, float *dbf, hipStream_t st,
                          const OrnScaleState *sc = nullptr);   // sc: 1/scale from the device state instead of inv_gs

// orn_merge.hip
int orn_launch_merge_fwd(const float *w3x3, const float *b3x3, const float *w3x1, const float *b3x1,
                         const float *w1x3, const float *b1x3, const float *w1, const float *w2, const float *w3,
                         int C, int O, float *T, float *wf, float *bf, hipStream_t st);
int orn_launch_merge_bwd(const float *g, const float *dbf, const float *w1, const float *w2, const float *w3,
                         const float *T, int C, int O, float *d3x3, float *db3x3, float *d3x1, float *db3x1,
                         float *d1x3, float *db1x3, float *dw1, float *dw2, float *dw3, float *ws, hipStream_t st);

// grouped merge (engine): all ERB layers per launch
struct OrnMergeLayer {
    int C, O;
    const float *w3x3, *w3x1, *w1x3, *w1, *w2, *w3;   // parameters
    const float *b3x3, *b1x3, *b3x1; float *bf;       // optional: bias merge folded into the S GEMM (null: separate launch)
    float *T, *wf;                                    // forward products
    float *w2t;                                       // optional: tap-major copy of w2, [9][O][2C] (orn_launch_w2_transpose); the T products then read 16-byte rows
    const float *g;                                   // dL/dWf (in the gradient arena)
    float *dT, *dw1p, *dw2, *dw3;                     // backward scratch / outputs
    float *dw2t;                                      // 16-bit modes: dW2 tap-major [9][O][2C], interleaved by the tail kernel
    // 16-bit fast layers (engine): the S GEMM's epilogue also writes the conv kernels' operand copies (half_kind 1 bf16 / 2 fp16)
    int half_kind, s2, Cp;
    void *wb, *wd;
    float *biasp;
};
size_t orn_merge_group_bytes();
// W2 [O][2C][3][3] -> w2t [9][O][2C] for every layer with a w2t buffer, one launch (first launch of the forward merge)
// (pack / par_blocks: the parameter-side half copies of the merge backward's operands ride behind the transposes)
int orn_launch_w2_transpose(int n_layers, const OrnMergeLayer *L, hipStream_t st, const void *pack = nullptr, int par_blocks = 0);
int orn_merge_groups_build(void *dev_tables, int n_layers, const OrnMergeLayer *L, int bwd_h16);
int orn_merge_group_tiles(int which, int n_layers, const OrnMergeLayer *L);
int orn_launch_merge_group(const void *dev_tables, int which, int tiles, hipStream_t st);
int orn_launch_merge_group_linear(const void *dev_tables, int which, int tiles, const OrnLinearJob &job, hipStream_t st,
                                  const void *pack = nullptr, int pack_blocks = 0);   // pack: trailing T -> Th pack jobs (orn_merge_h16_pack)
int orn_launch_merge_bias(const float *b3x3, const float *b1x3, const float *b3x1, int O, float *bf, hipStream_t st);
int orn_launch_merge_bwd_tail(const float *g, const float *dbf, int C, int O, float *d3x3, float *db3x3, float *d3x1,
                              float *db3x1, float *d1x3, float *db1x3, const float *dw1p, float *dw1, hipStream_t st);

// orn_merge_h16.hip: merge backward GEMMs of the 16-bit engine modes (packed half operands, fragments from global)
size_t orn_merge_h16_layer_halfs(int C, int O);
size_t orn_merge_h16_table_bytes();
size_t orn_merge_h16_host_bytes();
int orn_merge_h16_build(void *dev_tables, void *host, int n_layers, const OrnMergeLayer *L, void *const *bufs, OrnScaleState *sc);
int orn_launch_merge_h16_bwd(const void *dev_tables, const void *host, hipStream_t st, OrnScaleState *sc = nullptr);   // sc: the launching step's scale-state entry
const void *orn_merge_h16_pack(const void *host, int *par_blocks, int *t_blocks);
// the parameter-side (which = 1) or T -> Th (which = 2) pack jobs of the set as a launch of their own (instead of riders)
int orn_launch_merge_h16_pack_jobs(const void *host, int which, hipStream_t st);

// per-layer elementwise tails of the merge, all layers per launch
struct OrnMergeMisc {
    int C, O;
    const float *b3x3, *b1x3, *b3x1; float *bf;                       // forward bias
    const float *g, *dbf, *dw1p;                                      // backward inputs (dWf, dbf live in the grad arena)
    float *d3x1, *db3x1, *d1x3, *db1x3, *dw1;                         // backward outputs
    const float *dw2t; float *dw2;                                    // optional: dW2 [9][O][2C] -> [O][2C][3][3]
};
int orn_launch_merge_bias_all(int n, const OrnMergeMisc *L, hipStream_t st);
int orn_launch_merge_bwd_tail_all(int n, const OrnMergeMisc *L, hipStream_t st);

// orn_conv_f32.hip
int orn_launch_conv3x3_f32(const float *x, const float *w, const float *bias, int B, int C, int O, int H, int W,
                           int s, int epi, float *z, float *out, hipStream_t st, float *split_ws);
// head: fp32 engine only -- the last block's backward starts from the head's (out, dout) instead of da (orn_launch_head_bwd_fused_f32)
struct OrnHeadBwdFuse { const float *w, *out, *dout; int sigmoid; float *dw, *db, *hws; };
int orn_launch_conv_bwd_f32(const float *x, const float *wf, const float *z, const float *da, int B, int C, int O,
                            int H, int W, int s, float *dx, float *dwf, float *dbf, float *ws, hipStream_t st,
                            const OrnHeadBwdFuse *head = nullptr, const float *wd_ready = nullptr);   // wd_ready: the flipped / transposed kernel, already made
int orn_launch_flip_transpose_all(int n, const float *const *wf, float *const *wd, const int *O, const int *C, hipStream_t st);
int orn_head_bwd_fused_f32_blocks(int H, int W);
int orn_launch_head_bwd_fused_f32(const float *z, const float *w, const float *out, const float *dout, int Cn, int H, int W,
                                  int sigmoid, float *dy, float *dbp, float *dw, float *db, float *hws, hipStream_t st);

// orn_loss.hip
int orn_loss_init();
int orn_launch_loss(const float *pred, const float *target, const int *frame_idx, size_t frame_stride, int B, int Ch,
                    int H, int W, int loss_type, float loss_scale, float *stats, float *dpred, float *ws,
                    hipStream_t st, const OrnStepCur *cur = nullptr, float *ring = nullptr, OrnScaleState *sc = nullptr,
                    const float *tstats = nullptr,    // tstats: orn_loss_target_stats of the SAME frame table (Fusion6; indexed by *frame_idx)
                    OrnLossFinalJob *defer = nullptr);   // defer: the finalize stage is returned as a job instead of launched

// orn_conv_bf16.hip: the 16-bit MFMA fast path (channels-last buffers, see the file header).  The file is built
// twice (bf16 and, with -DORN_FP16, IEEE half); the engine reaches either build through this type-erased table.
struct OrnPrepLayer { const float *wf, *bf; int O, C, s; void *wb, *wd; float *biasp; int Cp; };   // Cp: channel stride (0: = C)
// deferred split-K reduction of a layer's wgrad slabs (wgrad called with dwf == nullptr leaves them in `slabs`)
struct OrnWgradReduce { const float *slabs; int H, W, C, O, s; float gscale; float *dwf, *dbf; OrnScaleState *sc; int smax; };   // sc (optional): 1/scale from the device state, non-finite results raise its flag
// deferred reduction of the 16-bit head backward's per-block partials (head_bwd called with dw == nullptr leaves them in ws)
struct OrnHeadFinish { const float *partial; int blocks, C; float gscale; float *dw, *db; OrnScaleState *sc; };
struct OrnWgradJob { const void *xpad, *dypad; int H, W, C, O, s; float *slabs; int smax; };   // wgrad into slabs, reduction deferred; smax > 0: at most that many split-K slabs (the reduction must be told the same)
// A5 head riding on the last block's forward (its epilogue holds all channels of an output pixel): out = act(W SiLU(z) + b).
// The launcher sets `fused` when the kernel it chose did the head; otherwise the caller launches head_fwd on z.
struct OrnHeadFuse { const float *w, *b; float *out; int sigmoid; int fused; };
struct OrnHalfOps {
    int (*conv_fwd)(const void *xpad, const void *wb, const float *bias_p, int H, int W, int Cin, int O, int s, void *z, void *apad,
                    hipStream_t st, int c_real, OrnHeadFuse *head);   // c_real <= Cin: input channels that are not zero padding; head: optional
    int (*conv_dgrad)(const void *dypad, const void *wd, int H, int W, int O, int C, const void *zprev, void *dyprev, int sp,
                      float *dx_f32, hipStream_t st, int c_real);   // c_real: output channels that are not zero padding
    size_t (*wgrad_ws_floats)(int H, int W, int O);
    int (*wgrad)(const void *xpad, const void *dypad, int H, int W, int C, int O, int s, float gscale, float *slabs, float *dwf,
                 float *dbf, hipStream_t st);
    int (*wgrad_batch)(int n, const OrnWgradJob *J, hipStream_t st, const OrnHeadFinish *hf, const OrnStemL2Job *l2, int side);   // side: launched on the engine's side stream (default wave priority)   // several layers' slabs in one launch (+ optional head finish, + the stem backward's first kernel)
    int (*wgrad_reduce_all)(int n, const OrnWgradReduce *L, hipStream_t st, const OrnStemW0Job *w0);   // all layers' reductions in one launch (+ the stem backward's last kernel)
    int (*prep_all)(int n, const OrnPrepLayer *L, hipStream_t st);
    int (*to_nhwc)(const float *src, int C, int Cp, int H, int W, void *dst, hipStream_t st);
    int (*to_nchw_f32)(const float *src, int C, int Cp, int H, int W, int nslab, float scale, float *dst, hipStream_t st, const OrnScaleState *sc);
    int (*dgrad_f32_slabs)(int H, int W, int O);
    int (*head_fwd)(const void *z, const float *w, const float *b, int C, size_t HW, int sigmoid, float *out, hipStream_t st);
    size_t (*head_bwd_ws_floats)(int C);
    int (*head_bwd_blocks)(int H, int W);
    int (*head_bwd)(const void *z, const float *w, const float *out, const float *dout, int C, int H, int W, int sigmoid, int sp,
                    float gs_up, void *dypad, float *dw, float *db, float *ws, hipStream_t st, const OrnScaleState *sc,
                    const OrnLossFinalJob *fin);   // sc: gs from the device state; fin (optional): the loss's finalize stage as one more work-group
};
const OrnHalfOps *orn_half_ops_bf16();
const OrnHalfOps *orn_half_ops_f16();
