/* orn.h -- C ABI of liborn.so: the MI355X-native (gfx950) Online-RepNeRV training hot path.
 *
 * The reference (maoqingyu1996/Boosting-Neural-Video-Representation-via-Online-Structural-
 * Reparameteration) is pure Python and exposes no FFI; the interface it *does* expose for this path is
 * the Python surface of model.py / utils.py / main_train.py.  Each entry point below names the
 * reference call site it replaces (file:line, relative to the reference root).  The Python mirror
 * in boosting-..._amd/ binds these with ctypes (see INTEGRATION.md for the stub a reference
 * maintainer would add).
 *
 * Conventions
 *   - Every tensor argument is a raw DEVICE pointer (torch.Tensor.data_ptr()).  fp32, NCHW,
 *     contiguous, weights [O,C,kh,kw] exactly as PyTorch stores them.  The caller allocates every
 *     input, output and workspace and keeps them alive until the stream has drained.  No ownership
 *     is transferred; the library allocates no device memory.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued there, nothing synchronises, every call is hipGraph-capturable.
 *   - Return value: 0 = success; < 0 = argument / shape error (text via orn_last_error);
 *     > 0 = hipError_t of a failed launch.  Nothing throws across the ABI, nothing calls exit().
 *   - All reductions are fixed-order two-pass trees: results are run-to-run bit-identical.
 *   - `ws` arguments are scratch; the required size comes from the matching *_ws_bytes().
 *   - One process drives ONE device (the deployment model of this path: one rank per GPU).  Kernels that need more than 64 KB
 *     of LDS raise their limit with hipFuncSetAttribute the first time they are launched and remember that in a process-wide
 *     flag; the attribute is per device, so a process that switches devices after its first call must not use this library
 *     on the second device.  Entry points may be called from several host threads as long as they use the same device
 *     (setting the attribute twice is harmless); orn_last_error is per thread.
 *   - Environment switches ORN_FWD_FORM1, ORN_DGRAD_FORM1, ORN_FWD2_APAD, ORN_HEAD_FUSED (read once per process) select
 *     kernel forms that are not the default for a shape; they exist for A/B measurements (tools/probes) and are covered by
 *     tests/test_gpu_bf16.py.  Results are parity-tested in every form.
 */
#ifndef ORN_H_
#define ORN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORN_VERSION 120          /* 0.1.2: + orn_loss_target_stats*, orn_engine_set_target_stats (round 3) */
/* Every entry point below is exported with default visibility; the library is built with -fvisibility=hidden, so these (and
 * the probe-only ones of orn_debug.h) are its whole dynamic symbol table. */
#define ORN_API __attribute__((visibility("default")))
#define ORN_MAX_LAYERS 8

#define ORN_OK 0
#define ORN_E_ARG (-1)           /* bad pointer / size / unsupported shape */
#define ORN_E_WS (-2)            /* workspace too small */
#define ORN_E_STATE (-3)         /* engine used in the wrong state */

ORN_API int orn_version(void);
/* Copies the last error text of the calling thread into buf (NUL-terminated); returns its length. */
ORN_API int orn_last_error(char *buf, size_t n);

/* ---- A1  PositionalEncoding.forward                                   utils.py:121-129 ------
 * out[b, 2i] = sin(arg), out[b, 2i+1] = cos(arg), arg = fp32(fp32(pos[b]*fp32(lbase^i))*fp32(pi)).
 * lbase_pow[i] = (float)(lbase**i) is computed by the caller in double (Python `lbase ** i`). */
ORN_API int orn_pe_fwd(const float *pos, int B, const float *lbase_pow, int levels, float *out, void *stream);

/* ---- A2  MLP stem (Linear+SiLU, Linear+SiLU)                         model.py:174-188,612 ----
 * pre1/h1: [B,Hd], pre2/h2: [B,Nout] (h2 is the block input viewed [B,C,fc_h,fc_w]). */
ORN_API int orn_stem_fwd(const float *embed, const float *w0, const float *b0, const float *w1, const float *b1,
                 int B, int E, int Hd, int Nout, float *pre1, float *h1, float *pre2, float *h2,
                 void *stream);
/* dh2 [B,Nout] -> dw0,db0,dw1,db1 (overwritten).  ws: B*(Nout + 258*Hd) floats. */
ORN_API int orn_stem_bwd(const float *embed, const float *w1, const float *pre1, const float *h1, const float *pre2,
                 const float *dh2, int B, int E, int Hd, int Nout, float *dw0, float *db0, float *dw1,
                 float *db1, float *ws, void *stream);

/* ---- A3  NeRVBlock.get_equivalent_kernel_bias (online ERB merge)     model.py:450-516 --------
 * T[O,C,3,3] (kept for the backward), wf[O,C,3,3], bf[O].  Bit-exact against oracle/merge_ref.c:
 * both contractions are single k-ordered fmaf chains. */
ORN_API int orn_erb_merge_fwd(const float *w3x3, const float *b3x3, const float *w3x1, const float *b3x1,
                      const float *w1x3, const float *b1x3, const float *w1, const float *w2,
                      const float *w3, int C, int O, float *T, float *wf, float *bf, void *stream);
ORN_API size_t orn_erb_merge_bwd_ws_bytes(int C, int O);
/* g = dL/dwf [O,C,3,3], dbf = dL/dbf [O] -> gradients of the 9 branch tensors (overwritten). */
ORN_API int orn_erb_merge_bwd(const float *g, const float *dbf, const float *w1, const float *w2, const float *w3,
                      const float *T, int C, int O, float *d3x3, float *db3x3, float *d3x1, float *db3x1,
                      float *d1x3, float *db1x3, float *dw1, float *dw2, float *dw3, void *ws,
                      size_t ws_bytes, void *stream);

/* ---- A4  NeRVBlock.forward: conv3x3(pad 1)+bias -> PixelShuffle(s) -> SiLU   model.py:539,567 --
 * x [B,C,H,W]; wf [O,C,3,3]; bf [O]; O = Cn*s*s.  z (pre-activation, post-shuffle) and
 * a = SiLU(z): [B,Cn,H*s,W*s].  z may be NULL for inference (decode). */
ORN_API int orn_conv3x3_ps_silu_fwd(const float *x, const float *wf, const float *bf, int B, int C, int O, int H,
                            int W, int s, float *z, float *a, void *stream);
ORN_API size_t orn_conv3x3_ps_silu_bwd_ws_bytes(int B, int C, int O, int H, int W);
/* da [B,Cn,Hs,Ws] -> dx [B,C,H,W] (NULL to skip), dwf [O,C,3,3], dbf [O] (overwritten). */
ORN_API int orn_conv3x3_ps_silu_bwd(const float *x, const float *wf, const float *z, const float *da, int B, int C,
                            int O, int H, int W, int s, float *dx, float *dwf, float *dbf, void *ws,
                            size_t ws_bytes, void *stream);

/* ---- A4 on the bf16 MFMA path (fp32 accumulate): same contract as the two calls above for B = 1,
 * C <= 96 (zero-padded to 96 in the staging), O % 32 == 0 and O % (s*s) == 0.  Inputs/outputs stay fp32 NCHW; the channels-last bf16
 * staging (DESIGN.md "data layout") lives in `ws`, which the caller must zero-fill once before the
 * first use (the one-pixel borders are never written).  The engine uses the same kernels without the
 * layout conversions.  fwd: `a` may be NULL (the last block's form: z only), then `z` must not be. */
ORN_API size_t orn_conv3x3_ps_silu_bf16_ws_bytes(int C, int O, int H, int W, int s);
ORN_API int orn_conv3x3_ps_silu_fwd_bf16(const float *x, const float *wf, const float *bf, int C, int O, int H, int W,
                                 int s, float *z, float *a, void *ws, size_t ws_bytes, void *stream);
ORN_API int orn_conv3x3_ps_silu_bwd_bf16(const float *x, const float *wf, const float *z, const float *da, int C, int O,
                                 int H, int W, int s, float *dx, float *dwf, float *dbf, void *ws,
                                 size_t ws_bytes, void *stream);

/* The forward conv kernel on the engine's own channels-last bf16 buffers (DESIGN.md "data layout"):
 * xpad [H+2][W+2][C] zero-bordered, wb [9][O'][C], bias_p [O'] (o' = (i*s+j)*Cn + n), z [H*s][W*s][Cn],
 * apad [H*s+2][W*s+2][Cn] or NULL.  C == 96; O % 32 == 0 (O % 96 == 0 for the two-work-group form the large layers take).
 * SLACK: the raw entry points of this group tile N by 96 or 128 channels and compute a ragged last tile on whatever lies behind
 * the operand (results of the missing channels are dropped, never stored).  When O % 128 != 0 the caller must therefore keep
 * `wb` / `wd` READABLE for 96*C elements past their [9][O'][C] / [9][C][O'] extent and `dypad` for 128 elements past its
 * [H+2][W+2][O'] extent (any finite or non-finite contents; never written).  The engine and the fp32-layout hooks above pad
 * their own workspaces; a caller that allocates exact sizes at e.g. O = 864 risks a fault at a page boundary. */
ORN_API int orn_conv_nhwc_bf16_fwd(const void *xpad, const void *wb, const float *bias_p, int H, int W, int C, int O,
                           int s, void *z, void *apad, void *stream);
/* the same kernel built for IEEE half (precision 2 of the engine, the mode bench.py's headline runs in): buffers hold fp16 */
ORN_API int orn_conv_nhwc_f16_fwd(const void *xpad, const void *wb, const float *bias_p, int H, int W, int C, int O,
                          int s, void *z, void *apad, void *stream);

/* Same for the two backward kernels (dgrad fused with SiLU' + un-shuffle into the previous layer's dypad;
 * wgrad + dbias into PyTorch-layout dwf [O][96][3][3], dbf [O]) (the timing-only ablation switch of
 * tools/probes lives in orn_debug.h). */
ORN_API int orn_dgrad_nhwc_bf16(const void *dypad, const void *wd, int H, int W, int O, int C, const void *zprev,
                        void *dyprev, int sp, void *stream);
ORN_API size_t orn_wgrad_nhwc_bf16_ws_bytes(int H, int W, int O);
ORN_API int orn_wgrad_nhwc_bf16(const void *xpad, const void *dypad, int H, int W, int C, int O, int s, float *slabs,
                        float *dwf, float *dbf, void *stream);
/* IEEE-half twins of the two backward kernels and of the fp32-layout block hooks (same arguments, half buffers, same
 * workspace sizes): every 16-bit kernel the engine's fp16 mode launches can be checked per op against the oracle. */
ORN_API int orn_dgrad_nhwc_f16(const void *dypad, const void *wd, int H, int W, int O, int C, const void *zprev,
                       void *dyprev, int sp, void *stream);
ORN_API int orn_wgrad_nhwc_f16(const void *xpad, const void *dypad, int H, int W, int C, int O, int s, float *slabs,
                       float *dwf, float *dbf, void *stream);
ORN_API int orn_conv3x3_ps_silu_fwd_f16(const float *x, const float *wf, const float *bf, int C, int O, int H, int W, int s,
                                float *z, float *a, void *ws, size_t ws_bytes, void *stream);
ORN_API int orn_conv3x3_ps_silu_bwd_f16(const float *x, const float *wf, const float *z, const float *da, int C, int O, int H,
                                int W, int s, float *dx, float *dwf, float *dbf, void *ws, size_t ws_bytes, void *stream);

/* ---- A5  head: 1x1 conv -> (tanh+1)/2 or sigmoid                      model.py:621-622 --------
 * a [B,C,H,W]; w [3,C,1,1]; b [3]; out [B,3,H,W]. */
ORN_API int orn_head_fwd(const float *a, const float *w, const float *b, int B, int C, int H, int W, int sigmoid,
                 float *out, void *stream);
ORN_API size_t orn_head_bwd_ws_bytes(int B, int C, int H, int W);
ORN_API int orn_head_bwd(const float *a, const float *w, const float *out, const float *dout, int B, int C, int H,
                 int W, int sigmoid, float *da, float *dw, float *db, void *ws, size_t ws_bytes,
                 void *stream);

/* ---- A7 + A10  loss_fn (L2 | L1 | Fusion6) and psnr_fn               utils.py:139-199 ---------
 * pred/target [B,Ch,H,W].  stats (device, 8 floats):
 *   [0] loss  [1] mean|p-t|  [2] mean (p-t)^2  [3] ssim (0 unless Fusion6)  [4] psnr = -10 log10(mse)
 * dpred = dLoss/dpred * loss_scale (NULL: forward only). */
#define ORN_LOSS_L2 0
#define ORN_LOSS_L1 1
#define ORN_LOSS_FUSION6 2
ORN_API size_t orn_loss_ws_bytes(int B, int Ch, int H, int W);
ORN_API int orn_loss_fwd_bwd(const float *pred, const float *target, int B, int Ch, int H, int W, int loss_type,
                     float loss_scale, float *stats, float *dpred, void *ws, size_t ws_bytes, void *stream);
/* Fusion6, engine only: the TARGET side of the SSIM statistics (utils.py:160 -> pytorch_msssim ssim: the 11-tap Gaussian of
 * t and of t*t on the valid map) of `n` resident frames [n][Ch][H][W] -> out [n][2][Ch][H-10][W-10] (orn_loss_target_stats_bytes).
 * A video's frames never change during its fit, so the engine can be handed this table once (orn_engine_set_target_stats) and
 * its steps then filter three maps (p, p*p, p*t) instead of five; same taps and fmaf order as the in-step form: results are
 * bit-identical with and without it. */
ORN_API size_t orn_loss_target_stats_bytes(int n, int Ch, int H, int W);
ORN_API int orn_loss_target_stats(const float *frames, int n, int Ch, int H, int W, float *out, void *stream);

/* ---- N3  msssim_fn: pytorch_msssim.ms_ssim(pred, target, data_range=1, size_average=True)   utils.py:201-211
 * Logging metric of the reference's train/eval loops (main_train.py:254); synchronises the stream (not for the
 * captured training step).  out: one device float.  min(H, W) must exceed 160. */
ORN_API size_t orn_msssim_ws_bytes(int B, int Ch, int H, int W);
ORN_API int orn_msssim(const float *pred, const float *target, int B, int Ch, int H, int W, float *out, void *ws,
               size_t ws_bytes, void *stream);

/* ---- A9  optim.Adam.step over one flat arena                          main_train.py:196,250 ---
 * p,g,m,v: n floats each.  step = 1-based global step.  weight decay 0, amsgrad off.
 * Hyper-parameters are doubles (as Python holds them): 1-beta, lr/(1-beta1^t) and sqrt(1-beta2^t)
 * are formed in double and rounded to fp32 once, exactly as torch.optim.Adam does. */
ORN_API int orn_adam_step(float *p, const float *g, float *m, float *v, size_t n, double lr, double beta1,
                  double beta2, double eps, int step, void *stream);

/* ---- A11  the whole per-frame training step as one engine           main_train.py:229-254 ----
 * The engine owns no memory: the caller passes four parameter-shaped arenas (params, grads, adam m,
 * adam v) laid out by `param_off` and one workspace.  Tensors inside an arena keep the PyTorch
 * shapes, so state_dict tensors can be views of `params`. */
typedef struct orn_layer_desc {
    int32_t C, O, s, H, W;          /* conv in-ch, conv out-ch (= Cn*s*s), stride, input H, W */
    /* offsets in floats into the parameter arenas; -1 = tensor absent */
    int64_t w3x3, b3x3;             /* ERB 3x3 branch, or the single conv of vanilla / deploy */
    int64_t w3x1, b3x1, w1x3, b1x3; /* ERB only */
    int64_t w1, w2, w3;             /* ERB only: 1x1 -> 3x3 -> 1x1 */
} orn_layer_desc;

typedef struct orn_engine_desc {
    int32_t n_layers;
    int32_t erb;                    /* 1: ERB online merge; 0: single 3x3 conv per block */
    int32_t embed_len, stem_dim, fc_h, fc_w, fc_dim;
    int32_t sigmoid;                /* head activation (model.py:622) */
    int32_t loss_type;              /* ORN_LOSS_* */
    int32_t precision;              /* 0: fp32 everywhere; 1: bf16, 2: IEEE fp16 activations + 16-bit MFMA convs (fp32 accumulate) */
    double beta1, beta2, eps;
    int64_t stem_w0, stem_b0, stem_w1, stem_b1, head_w, head_b;
    int64_t n_params;               /* arena length in floats */
    orn_layer_desc layer[ORN_MAX_LAYERS];
} orn_engine_desc;

/* One entry of the per-step schedule (device array, uploaded per epoch by the caller). */
typedef struct orn_step_sched {
    int32_t frame;                  /* index into frames / embed table */
    int32_t step;                   /* 1-based global optimiser step (Adam bias correction) */
    float lr;                       /* adjust_lr() value for this step (utils.py:240-259) */
    float pad;
} orn_step_sched;

typedef struct orn_engine orn_engine;

ORN_API size_t orn_engine_ws_bytes(const orn_engine_desc *d);
ORN_API int orn_engine_create(const orn_engine_desc *d, float *params, float *grads, float *adam_m, float *adam_v,
                      void *ws, size_t ws_bytes, orn_engine **out);
ORN_API void orn_engine_destroy(orn_engine *e);
/* Forward only (decode): embed [E] device -> img [3,H,W] device. */
ORN_API int orn_engine_decode(orn_engine *e, const float *embed, float *img, void *stream);
/* One optimiser step.  frames [n_frames,3,H,W], embeds [n_frames,E] (device); sched: device array,
 * `cursor` a device int32 the step reads and post-increments, so `n` back-to-back steps consume
 * sched[cursor..cursor+n).  stats_out: device [n_slots][8] ring written at slot (cursor % n_slots). */
ORN_API int orn_engine_train_step(orn_engine *e, const float *frames, const float *embeds,
                          const orn_step_sched *sched, int32_t *cursor, float *stats_out, int32_t n_slots,
                          void *stream);
/* Optional 0/1 gradient mask in the arena layout (device, float, 16-byte aligned; null removes it): gradients are
 * multiplied by it before Adam.  The prune fine-tune of main_eval.py:213-531 -- torch.nn.utils.prune keeps
 * weight = weight_orig * mask, so pruned entries never receive a gradient -- and its frozen tensors. */
ORN_API int orn_engine_set_grad_mask(orn_engine *e, const float *mask);
/* Optional table of orn_loss_target_stats for the frame table the following steps are given (device, 16-byte aligned, caller-owned,
 * alive while set; null removes it).  Fusion6 only; other loss types ignore it. */
ORN_API int orn_engine_set_target_stats(orn_engine *e, const float *stats);
/* One eager training step with HIP events around the conv launches, on the launch stream; synchronises it.  ms_out (host,
 * 2*n_layers + 2 floats): [i] forward conv of layer i, [n_layers + i] dgrad launch of layer i (0: none), [2*n_layers] the
 * batched wgrad launch of the 16-bit layers, [2*n_layers + 1] its split-K reduction (0 in fp32 mode, where each layer's
 * backward is one multi-kernel call reported under its dgrad slot).  Measurement hook for bench.py's roofline leg -- no
 * reference counterpart. */
ORN_API int orn_engine_profile_step(orn_engine *e, const float *frames, const float *embeds, const orn_step_sched *sched,
                            int32_t *cursor, float *stats_out, int32_t n_slots, float *ms_out, void *stream);
/* n_steps optimiser steps enqueued on `stream` without a graph (main_train.py:229-254, the loop body n times).  Engines in a
 * 16-bit mode with at least two blocks on the fast path run them PIPELINED: the last block's weight gradient, its slab reduction,
 * merge backward, Adam update (with the head's) and next merge forward run on a second stream the engine owns, forked off at the end
 * of the backward and joined in front of that block's forward conv of the next step, so that this chain overlaps the latency-bound
 * launches of the step boundary.  Same arithmetic, same order of every sum: bit-identical to orn_engine_train_step.  All work is
 * joined back into `stream` before the call returns (stream order on `stream` covers it). */
ORN_API int orn_engine_train_steps(orn_engine *e, const float *frames, const float *embeds, const orn_step_sched *sched,
                           int32_t *cursor, float *stats_out, int32_t n_slots, int32_t n_steps, void *stream);
/* Capture one train step into a hipGraph on `stream` and replay it n times (same arguments as above). */
ORN_API int orn_engine_train_steps_graph(orn_engine *e, const float *frames, const float *embeds,
                                 const orn_step_sched *sched, int32_t *cursor, float *stats_out,
                                 int32_t n_slots, int32_t n_steps, void *stream);
/* Dynamic loss scale + non-finite guard (the reference trains in fp32 and has neither; torch.cuda.amp.GradScaler is the
 * model).  The 16-bit gradient tensors of precision 2 travel multiplied by a scale held in device memory (2^20 at creation;
 * 1 for the other precisions).  A step whose gradients (or loss) are not finite leaves parameters and Adam moments untouched
 * and halves the scale; 2000 clean steps double it again up to its initial value.  All of it happens on the device, inside
 * the captured step.  out8 (host): {scale, 1/scale, ceiling, flag, steps skipped, clean steps, halvings, 0}; synchronises.
 * Where this differs from GradScaler:
 *  - granularity: the flag is PER STEP (one scale-state entry per step of the unrolled graph: an overflowing step skips itself
 *    only, the clean steps before and behind it in the same graph launch update the parameters), but the scale changes only
 *    where the device-side schedule advances, once per graph launch (orn_engine_train_steps_graph replays groups of 4 steps;
 *    orn_engine_train_step advances every step): one halving per group however many of its steps overflowed;
 *  - Adam's step numbers of a group are fixed when it is launched: a step skipped INSIDE a group leaves the bias corrections of
 *    the at most 3 steps behind it one count ahead (the next group is exact again);
 *  - the merge backward of the 16-bit modes rounds the UN-scaled weight gradient times 2^14 to IEEE half: |dWf| > 4 raises the
 *    same flag, and no loss scale cures that.  Such a fit has diverged; main_train restores the start of the epoch and
 *    continues in a wider precision (bf16 keeps that operand format, fp32 does not have it);
 *  - Adam's step count, in the device schedule and in the checkpoint's optimizer entry, excludes skipped steps;
 *  - orn_engine_train_steps (pipelined form) advances the schedule every step, like orn_engine_train_step.  The skip decision of a
 *    step is taken by the Adam launch on the caller's stream, behind every detector that runs there (the loss, the hand-off into
 *    the fp32 part, the lower blocks' slab reduction and merge-backward pack, the head's dW finish is on the side stream but the dy it
 *    sums is covered by the lower blocks' detectors: a non-finite dy of the last block reaches them through the dgrad chain), and
 *    the side stream's Adam launch (last block + head) follows that decision.  Two detectors of the side branch run BEHIND the
 *    decision: the last block's slab reduction (unreachable alone, by the argument above) and the fp16 copy of its merged-kernel
 *    gradient (|dWf| > 4).  If one of them fires alone, only the side stream's update of that step is skipped (the parameters stay
 *    finite; the step is not counted as skipped).  The serial forms skip the whole step in that case. */
ORN_API int orn_engine_scale_state(orn_engine *e, float *out8);
/* Overrides the live scale (>= 1) and, if gs_max > 0, its ceiling: tests inject an overflowing step this way. */
ORN_API int orn_engine_set_grad_scale(orn_engine *e, float gs, float gs_max);
/* Merged (deploy) kernel/bias of layer i, valid after a decode / train step (model.py:395-448). */
ORN_API int orn_engine_fused_kernel(orn_engine *e, int layer, const float **wf, const float **bf);

#ifdef __cplusplus
}
#endif
#endif /* ORN_H_ */
