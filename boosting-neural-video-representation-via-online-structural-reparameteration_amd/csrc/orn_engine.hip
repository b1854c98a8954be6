// A11  The per-frame training step of main_train.py:229-254 as one native engine:
//   stem -> N x {online ERB merge, conv3x3+PixelShuffle+SiLU} -> head -> loss (+PSNR) -> backward
//   -> Adam, every launch on the caller's stream, capturable as one hipGraph.
// The engine owns no device memory: four parameter-shaped arenas + one workspace come from the
// caller (PyTorch is only the allocator).
#include "orn_internal.h"
#include <new>
#include <cstdlib>

struct LayerBuf {
    float *T, *wf, *bf;   // merge products (ERB) -- wf/bf alias the params for vanilla/deploy
    float *w2t;           // tap-major copy of W2 (ERB): the T products read contiguous rows
    float *dT, *dw1p;     // merge backward scratch (ERB)
    float *wd32;          // fp32 layers: flipped / transposed merged kernel for the dgrad (made for all layers in one launch)
    void *mh16;           // half operand copies of the merge backward (16-bit modes, orn_merge_h16.hip)
    float *dw2t;          // dW2 tap-major [9][O][2C] (16-bit modes)
    float *z, *a;         // block output (pre-activation, activation)          [fp32 layers]
    float *da;            // gradient wrt the block output                       [fp32 layers]
    // bf16 fast path (precision 1, layers >= ff): channels-last bf16, see orn_conv_bf16.hip
    uint16_t *xpad;       // conv input, zero-bordered [H+2][W+2][C]   (16-bit elements: bf16 or fp16)
    uint16_t *zb;         // pre-activation [Hs][Ws][Cn]
    uint16_t *dypad;      // gradient wrt conv output, zero-bordered [H+2][W+2][O']
    uint16_t *wb, *wd;    // merged kernel in 16 bit: forward / dgrad operand layouts
    float *biasp;         // bias in o' order
    float *wslab;         // split-K slabs of this layer's wgrad (reduced for all layers at the end of the backward)
};

struct orn_engine {
    orn_engine_desc d;
    float *params, *grads, *m, *v;
    const float *gmask;              // optional 0/1 gradient mask (prune fine-tune), same layout as the arenas
    const float *tstats;             // optional orn_loss_target_stats of the resident video (Fusion6)
    float *ws;
    // workspace carve
    float *pre1, *h1, *pre2, *h2, *dh2;
    float *img, *dimg, *stats;
    float *loss_ws;
    float *scratch;                  // shared scratch for the backward kernels
    float *head_ws;                  // 16-bit head backward: per-block partials (live until the deferred finish)
    OrnStepCur *cur;                 // state of the step in flight (device)
    LayerBuf L[ORN_MAX_LAYERS];
    int Hout, Wout, Cn_last;
    const OrnHalfOps *ops;           // 16-bit fast-path kernels (bf16 or fp16 build)
    float gs;                        // INITIAL gradient scale of the 16-bit gradient tensors (1 for bf16, 2^20 for fp16)
    OrnScaleState *sc;               // device: the live scale + non-finite flag (dynamic loss scaling, skipped steps)
    // (ERB) the grouped merge launches work on a SET of layers: all of them (a serial step, decode), or -- in the pipelined
    // form of the step, below -- all but the last block on the caller's stream and the last block alone on the side stream
    struct MergeSet {
        int n;                          // layers in the set (0: set not built)
        OrnMergeLayer ml[ORN_MAX_LAYERS];   // the layers as the merge launchers see them
        int layer[ORN_MAX_LAYERS];      // their indices in the engine
        void *tables;                   // device-resident grouped-GEMM problem tables
        int tiles[4];
        void *mh_tables, *mh_host;      // 16-bit modes: tables of the packed-operand merge backward (device / host)
    } mset[3];
    int ff;                          // first layer on the bf16 fast path (== n_layers: none)
    float *dxn;                      // fp32 NHWC dgrad output of layer ff (converted to NCHW for the fp32 part)
    size_t stem_ws;                  // floats of `scratch` the stem backward uses (the fused first block's dx slabs sit behind them)
    bool stage0;                     // layer 0 is the only fp32 block and fits orn_stage0.hip: fused forward / backward
    // graph cache: one captured train step, and ORN_GRAPH_UNROLL steps back to back (the schedule is device-side, so
    // a longer graph is the same nodes repeated; it amortises the ~8 us gap between graph launches)
    hipGraph_t graph, graph_u;
    hipGraphExec_t graph_exec, graph_exec_u;
    const void *g_frames, *g_embeds, *g_sched, *g_cursor, *g_stats;
    int g_slots;
    hipStream_t g_stream;
    // Pipelined form of the step (orn_engine_train_steps; 16-bit engines with >= 2 blocks on the fast path): the LAST block's
    // weight gradient, slab reduction, merge backward, Adam update and next merge forward run on a second stream (`side`),
    // forked off the caller's stream inside the backward (side_fork_at: behind the last block's dgrad) and joined in front of the
    // last block's forward conv of the NEXT step.  That chain -- one full-chip MFMA launch and a tail of small ones -- then runs beside the latency-bound
    // launches of the step boundary (merge backward / Adam / merge forward of the lower blocks, stem, first blocks) instead of
    // in front of them.  Same arithmetic as the serial step (bit-identical results); what it needs:
    //   ev_fork   main -> side: dy / x of the last block, the head's partials and the step's copied schedule state are final
    //   ev_adam   main -> side: the step's skip decision has been taken (the side's Adam launch follows it)
    //   ev_wgrad  side -> main: the last block's input buffer may be overwritten (by the forward conv of the block below it)
    //   ev_below  side -> main: the weight gradient of the block BELOW the last one, which the side stream computes first (beside that
    //                           block's own dgrad launch on the caller's stream: 230 work-groups on 512 slots at 720p, and a wgrad
    //                           of 216), is complete: its slabs may be reduced, its input buffer overwritten
    //   ev_join   side -> main: the last block's merged kernel (and the head's parameters) of the next step are ready
    // cur_side / sc_side: this step's schedule entry and loss scale, copied by the loss's finalize stage (the main stream advances
    // to the next step while the side branch still reads them); sc_side->flag also collects what the side branch's own two
    // detectors raise behind the main commit (include/orn.h, the loss-scale comment).
    hipStream_t side;
    hipEvent_t ev_fork, ev_adam, ev_wgrad, ev_join, ev_below;
    OrnStepCur *cur_side;
    OrnScaleState *sc_side;
    size_t side_lo;                  // parameters [side_lo, n_params) belong to the side branch's Adam launch (last block + head)
    bool pipe_ok;                    // this engine can run the pipelined form
    bool side_busy;                  // a side branch is in flight (joined by the next step's forward or at the end of the call)
    // orn_engine_profile_step: HIP events around every forward conv launch of an eager step
    bool prof;
    hipEvent_t prof_ev[4 * ORN_MAX_LAYERS + 4];      // pairs: forward conv of layer i, dgrad launch of layer i, wgrad batch, its reduction
};

#define ORN_GRAPH_UNROLL 4
static_assert(ORN_SCALE_SLOTS >= ORN_GRAPH_UNROLL, "one scale-state entry per step of the unrolled graph");

static inline size_t al(size_t floats) { return orn_align(floats * 4) / 4; }
static bool side_takes_below(const orn_engine *e);

extern "C" size_t orn_conv3x3_ps_silu_bwd_ws_bytes(int B, int C, int O, int H, int W);
extern "C" size_t orn_erb_merge_bwd_ws_bytes(int C, int O);
extern "C" size_t orn_head_bwd_ws_bytes(int B, int C, int H, int W);
extern "C" size_t orn_loss_ws_bytes(int B, int Ch, int H, int W);

static int check_desc(const orn_engine_desc *d)
{
    ORN_REQUIRE(d, "engine: null desc");
    ORN_REQUIRE(d->n_layers >= 1 && d->n_layers <= ORN_MAX_LAYERS, "engine: n_layers=%d out of range", d->n_layers);
    ORN_REQUIRE(d->precision >= 0 && d->precision <= 2, "engine: precision %d not built", d->precision);
    ORN_REQUIRE(d->embed_len > 0 && d->stem_dim > 0 && d->fc_h > 0 && d->fc_w > 0 && d->fc_dim > 0, "engine: bad stem geometry");
    int C = d->fc_dim, H = d->fc_h, W = d->fc_w;
    for (int i = 0; i < d->n_layers; ++i) {
        const orn_layer_desc &l = d->layer[i];
        ORN_REQUIRE(l.C == C && l.H == H && l.W == W,
                    "engine: layer %d geometry (C=%d,H=%d,W=%d) does not chain (expected %d,%d,%d)", i, l.C, l.H, l.W, C, H, W);
        ORN_REQUIRE(l.s >= 1 && l.O > 0 && l.O % (l.s * l.s) == 0, "engine: layer %d: O=%d not divisible by s^2", i, l.O);
        ORN_REQUIRE(l.w3x3 >= 0 && l.b3x3 >= 0, "engine: layer %d: missing 3x3 weight/bias", i);
        if (d->erb)
            ORN_REQUIRE(l.w3x1 >= 0 && l.b3x1 >= 0 && l.w1x3 >= 0 && l.b1x3 >= 0 && l.w1 >= 0 && l.w2 >= 0 && l.w3 >= 0,
                        "engine: layer %d: missing ERB branch tensors", i);
        C = l.O / (l.s * l.s); H *= l.s; W *= l.s;
    }
    ORN_REQUIRE(d->n_params > 0 && d->n_params % 4 == 0, "engine: n_params must be a positive multiple of 4");
    ORN_REQUIRE(d->loss_type >= 0 && d->loss_type <= 2, "engine: bad loss_type %d", d->loss_type);
    return 0;
}

// Computes the workspace layout (in floats); if e != null also fills its pointers.
#define ORN_FAST_C 96          // input channels per pixel of every 16-bit channels-last buffer
static bool layer_is_fast(const orn_layer_desc &l)
{
    // O % 96: whole 32-channel MFMA blocks forward (a last N tile of 96 is ragged) and whole 96-channel K chunks in the dgrad
    return l.C == ORN_FAST_C && l.O % 96 == 0 && l.O % (l.s * l.s) == 0;
}

// First layer from which every layer (and the head) can run on the bf16 MFMA path.
static int first_fast_layer(const orn_engine_desc *d)
{
    if (d->precision == 0) return d->n_layers;
    const orn_layer_desc &last = d->layer[d->n_layers - 1];
    const int cn = last.O / (last.s * last.s);
    if (!(cn == 32 || cn == 64 || cn == 96 || cn == 128)) return d->n_layers;
    int ff = d->n_layers;
    while (ff > 0 && layer_is_fast(d->layer[ff - 1])) --ff;
    // one narrower layer below them may join, zero-padded to 96 input channels (its fp32 NCHW input is converted
    // anyway): 3.7x the FLOPs of C=26 on a 16x faster pipe, and none of the fp32 path's small launches
    if (ff > 0 && ff < d->n_layers) {
        const orn_layer_desc &l = d->layer[ff - 1];
        if (l.C < ORN_FAST_C && l.O % 96 == 0 && l.O % (l.s * l.s) == 0) --ff;
    }
    return ff;
}

static size_t layout(const orn_engine_desc *d, orn_engine *e)
{
    size_t off = 0;
    const int ff = first_fast_layer(d);
    float *dxn = nullptr;
    float *base = e ? e->ws : nullptr;
    auto take = [&](size_t floats) { float *p = base ? base + off : nullptr; off += al(floats); return p; };
    const int Nout = d->fc_h * d->fc_w * d->fc_dim;
    float *pre1 = take(d->stem_dim), *h1 = take(d->stem_dim), *pre2 = take(Nout), *h2 = take(Nout), *dh2 = take(Nout);
    size_t scratch = orn_stem_bwd_ws_floats(1, d->stem_dim, Nout);
    LayerBuf L[ORN_MAX_LAYERS] = {};
    int Cn = d->fc_dim, H = d->fc_h, W = d->fc_w;
    for (int i = 0; i < d->n_layers; ++i) {
        const orn_layer_desc &l = d->layer[i];
        const size_t wsz = (size_t)l.O * l.C * 9;
        if (d->erb) {
            L[i].T = take(wsz); L[i].wf = take(wsz); L[i].bf = take(l.O);
            L[i].w2t = (2 * l.C * 9 * 4 * 4 <= 64 * 1024) ? take((size_t)9 * l.O * 2 * l.C) : nullptr;
            L[i].dT = take(wsz); L[i].dw1p = take((size_t)9 * 2 * l.C * l.C);
            if (d->precision != 0) {
                L[i].mh16 = take((orn_merge_h16_layer_halfs(l.C, l.O) + 1) / 2);
                L[i].dw2t = take((size_t)9 * l.O * 2 * l.C);
            }
        }
        Cn = l.O / (l.s * l.s); H = l.H * l.s; W = l.W * l.s;
        const size_t asz = (size_t)Cn * H * W;
        size_t s1;
        if (i < ff) {
            L[i].z = take(asz); L[i].a = take(asz); L[i].da = take(asz);
            L[i].wd32 = take(wsz);
            s1 = orn_conv3x3_ps_silu_bwd_ws_bytes(1, l.C, l.O, l.H, l.W) / 4;
            const size_t s0 = al(orn_stem_bwd_ws_floats(1, d->stem_dim, Nout)) + (size_t)orn_stage0_slabs(l.O, l.s) * l.C * l.H * l.W;   // fused first block
            if (s0 > s1) s1 = s0;
        } else {
            // halfs are carved as floats (2 per float)
            const size_t wpz = (size_t)l.O * ORN_FAST_C * 9 + 96 * ORN_FAST_C;      // + the rows a ragged last N tile reads past the end
            L[i].xpad = (uint16_t *)take(((size_t)(l.H + 2) * (l.W + 2) * ORN_FAST_C + 1) / 2);
            L[i].zb = (uint16_t *)take((asz + 1) / 2);
            L[i].dypad = (uint16_t *)take(((size_t)(l.H + 2) * (l.W + 2) * l.O + 128 + 1) / 2);   // + a ragged wgrad tile's over-read
            L[i].wb = (uint16_t *)take((wpz + 1) / 2);
            L[i].wd = (uint16_t *)take((wpz + 1) / 2);
            L[i].biasp = take(l.O);
            L[i].wslab = take(orn_half_ops_bf16()->wgrad_ws_floats(l.H, l.W, l.O));
            const OrnHalfOps *ops = orn_half_ops_bf16();     // sizes do not depend on the element type
            if (i == ff) dxn = take((size_t)l.H * l.W * ORN_FAST_C * ops->dgrad_f32_slabs(l.H, l.W, l.O));
            s1 = al(ops->wgrad_ws_floats(l.H, l.W, l.O));
            const size_t sd = (size_t)l.H * l.W * ORN_FAST_C * ops->dgrad_f32_slabs(l.H, l.W, l.O);
            if (ops->dgrad_f32_slabs(l.H, l.W, l.O) > 1 && sd > s1) s1 = sd;
        }
        if (d->erb) { const size_t s2 = orn_erb_merge_bwd_ws_bytes(l.C, l.O) / 4; if (s2 > s1) s1 = s2; }
        if (s1 > scratch) scratch = s1;
    }
    const size_t isz = (size_t)3 * H * W;
    float *img = take(isz), *dimg = take(isz), *stats = take(8);
    float *loss_ws = take(orn_loss_ws_bytes(1, 3, H, W) / 4);
    const size_t s3 = (ff < d->n_layers) ? orn_half_ops_bf16()->head_bwd_ws_floats(Cn) : orn_head_bwd_ws_bytes(1, Cn, H, W) / 4;
    if (s3 > scratch) scratch = s3;
    float *scr = take(scratch);
    // (fp32 engine: partials of the fused head backward of the last block, stride 2 only)
    const orn_layer_desc &ll = d->layer[d->n_layers - 1];
    float *head_ws = (ff < d->n_layers) ? take(orn_half_ops_bf16()->head_bwd_ws_floats(Cn))
                   : (ll.s == 2 ? take((size_t)(orn_head_bwd_fused_f32_blocks(ll.H, ll.W) + 1) * (3 * Cn + 3)) : nullptr);
    float *cur = take((sizeof(OrnStepCur) * ORN_GRAPH_UNROLL + 3) / 4 + 16);    // one cursor state per step of the unrolled graph
    float *scs = take(sizeof(OrnScaleState) * ORN_SCALE_SLOTS / 4);     // one entry per step of the unrolled graph (entry 0 = master)
    float *mtab[3], *mhtab[3];
    for (int k = 0; k < 3; ++k) {
        mtab[k] = d->erb ? take(orn_merge_group_bytes() / 4) : nullptr;
        mhtab[k] = (d->erb && d->precision != 0) ? take(orn_merge_h16_table_bytes() / 4) : nullptr;
    }
    float *cur_side = take((sizeof(OrnStepCur) + 3) / 4 + 16);
    float *sc_side = take(sizeof(OrnScaleState) / 4 + 16);
    if (e) {
        e->pre1 = pre1; e->h1 = h1; e->pre2 = pre2; e->h2 = h2; e->dh2 = dh2;
        e->img = img; e->dimg = dimg; e->stats = stats; e->loss_ws = loss_ws; e->scratch = scr; e->head_ws = head_ws;
        e->cur = (OrnStepCur *)cur;
        e->sc = (OrnScaleState *)scs;
        for (int i = 0; i < d->n_layers; ++i) e->L[i] = L[i];
        e->Hout = H; e->Wout = W; e->Cn_last = Cn;
        e->ff = ff; e->dxn = dxn;
        e->stem_ws = al(orn_stem_bwd_ws_floats(1, d->stem_dim, Nout));
        const orn_layer_desc &l0 = d->layer[0];
        e->stage0 = d->precision != 0 && ff == 1 && d->n_layers > 1 && orn_stage0_supported(l0.C, l0.O, l0.H, l0.W, l0.s);
        for (int k = 0; k < 3; ++k) { e->mset[k].tables = mtab[k]; e->mset[k].mh_tables = mhtab[k]; }
        e->cur_side = (OrnStepCur *)cur_side;
        e->sc_side = (OrnScaleState *)sc_side;
    }
    return off * 4;
}

extern "C" size_t orn_engine_ws_bytes(const orn_engine_desc *d)
{
    if (check_desc(d) != 0) return 0;
    return layout(d, nullptr);
}

extern "C" int orn_engine_create(const orn_engine_desc *d, float *params, float *grads, float *adam_m, float *adam_v,
                                 void *ws, size_t ws_bytes, orn_engine **out)
{
    ORN_TRY(check_desc(d));
    ORN_REQUIRE(params && out && ws, "engine_create: null pointer");
    ORN_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)adam_m | (uintptr_t)adam_v | (uintptr_t)ws) % 256 == 0,
                "engine_create: arenas and workspace must be 256-byte aligned");
    const size_t need = layout(d, nullptr);
    if (ws_bytes < need) {
        orn_set_error("engine_create: workspace %zu < %zu", ws_bytes, need);
        return ORN_E_WS;
    }
    ORN_TRY(orn_loss_init());
    orn_engine *e = new (std::nothrow) orn_engine();
    ORN_REQUIRE(e, "engine_create: out of host memory");
    e->d = *d;
    e->params = params; e->grads = grads; e->m = adam_m; e->v = adam_v;
    e->gmask = nullptr;
    e->tstats = nullptr;
    e->ws = (float *)ws;
    e->graph = nullptr; e->graph_exec = nullptr; e->graph_u = nullptr; e->graph_exec_u = nullptr;
    e->prof = false;
    for (int k = 0; k < 3; ++k) { e->mset[k].n = 0; e->mset[k].mh_host = nullptr; }
    e->side = nullptr; e->ev_fork = e->ev_adam = e->ev_wgrad = e->ev_join = e->ev_below = nullptr;
    e->pipe_ok = false; e->side_busy = false; e->side_lo = 0;
    for (int i = 0; i < 4 * ORN_MAX_LAYERS + 4; ++i) e->prof_ev[i] = nullptr;
    e->ops = (d->precision == 2) ? orn_half_ops_f16() : orn_half_ops_bf16();
    e->gs = (d->precision == 2) ? 1048576.0f : 1.0f;
    layout(d, e);
    {   // the one-pixel borders of the channels-last buffers must be (and stay) zero
        hipError_t rc = hipMemset(ws, 0, need);
        if (rc == hipSuccess) rc = hipDeviceSynchronize();
        if (rc != hipSuccess) { orn_set_error("engine_create: hipMemset failed: %s", hipGetErrorString(rc)); delete e; return (int)rc; }
        OrnScaleState s0[ORN_SCALE_SLOTS] = {};
        for (int r = 0; r < ORN_SCALE_SLOTS; ++r) { s0[r].gs = e->gs; s0[r].inv_gs = 1.0f / e->gs; s0[r].gs_max = e->gs; }
        rc = hipMemcpy(e->sc, s0, sizeof(s0), hipMemcpyHostToDevice);
        if (rc == hipSuccess) rc = hipMemcpy(e->sc_side, s0, sizeof(OrnScaleState), hipMemcpyHostToDevice);
        if (rc != hipSuccess) { orn_set_error("engine_create: scale state upload failed: %s", hipGetErrorString(rc)); delete e; return (int)rc; }
    }
    // Can this engine run the pipelined step?  16-bit mode with the last block AND the block below it on the fast path (the
    // hand-off points are their buffers), gradients present, and an arena in which the last block's tensors and the head's lie
    // behind everything else (one Adam launch per stream).  ORN_NO_PIPELINE: tools/probes A/B.
    {
        const int nl = d->n_layers;
        const orn_layer_desc &ll = d->layer[nl - 1];
        bool ok = d->precision != 0 && e->ff < nl - 1 && grads && adam_m && adam_v && orn_probe_env("ORN_NO_PIPELINE") == nullptr;
        if (ok) {
            const int64_t lo_all[9] = {ll.w3x3, ll.b3x3, ll.w3x1, ll.b3x1, ll.w1x3, ll.b1x3, ll.w1, ll.w2, ll.w3};
            int64_t lo = d->n_params;
            for (int k = 0; k < (d->erb ? 9 : 2); ++k) if (lo_all[k] >= 0 && lo_all[k] < lo) lo = lo_all[k];
            ok = d->head_w >= lo && d->head_b >= lo && d->stem_w0 < lo && d->stem_b0 < lo && d->stem_w1 < lo && d->stem_b1 < lo && lo % 64 == 0;
            for (int i = 0; i < nl - 1 && ok; ++i) {
                const orn_layer_desc &l = d->layer[i];
                const int64_t o[9] = {l.w3x3, l.b3x3, l.w3x1, l.b3x1, l.w1x3, l.b1x3, l.w1, l.w2, l.w3};
                for (int k = 0; k < (d->erb ? 9 : 2); ++k) ok = ok && o[k] < lo;
            }
            e->side_lo = (size_t)lo;
        }
        if (ok) {
            // (A high-priority side stream returns ~2 us per step -- the branch's small launches queue less behind the 8-wave forward
            // launches of the middle blocks -- but while such a queue exists, even idle, every OTHER stream of the process loses: the
            // fp32 engine's step, run next to an idle fp16 engine, went 6.66 -> 8.25 ms.  Default priority.)
            hipError_t rc = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking);
            hipEvent_t *evs[5] = {&e->ev_fork, &e->ev_adam, &e->ev_wgrad, &e->ev_join, &e->ev_below};
            // hipEventDisableSystemFence: these events order streams of ONE device; what the flag gives up is visibility to the host and
            // to other devices at the record (hip_runtime_api.h), which the caller's own synchronisation of its stream provides.  The
            // record is then a lighter packet: ~4 us per step (3 records / waits on the caller's stream).  ORN_EVENT_FENCE=1: the default events.
            const unsigned evflags = hipEventDisableTiming | (orn_probe_env("ORN_EVENT_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);
            for (int k = 0; k < 5 && rc == hipSuccess; ++k) rc = hipEventCreateWithFlags(evs[k], evflags);
            if (rc != hipSuccess) { orn_set_error("engine_create: side stream: %s", hipGetErrorString(rc)); orn_engine_destroy(e); return (int)rc; }
        }
        e->pipe_ok = ok;
    }
    if (!d->erb)
        for (int i = 0; i < d->n_layers; ++i) {
            e->L[i].T = nullptr;
            e->L[i].wf = params + d->layer[i].w3x3;
            e->L[i].bf = params + d->layer[i].b3x3;
        }
    else {
        auto fill = [&](int i, OrnMergeLayer &m, bool half_in_S) {
            const orn_layer_desc &l = d->layer[i];
            m.C = l.C; m.O = l.O;
            m.w3x3 = params + l.w3x3; m.w3x1 = params + l.w3x1; m.w1x3 = params + l.w1x3;
            m.w1 = params + l.w1; m.w2 = params + l.w2; m.w3 = params + l.w3;
            m.T = e->L[i].T; m.wf = e->L[i].wf; m.w2t = e->L[i].w2t;
            m.b3x3 = params + l.b3x3; m.b1x3 = params + l.b1x3; m.b3x1 = params + l.b3x1; m.bf = e->L[i].bf;
            m.g = grads ? grads + l.w3x3 : nullptr;
            m.dT = e->L[i].dT; m.dw1p = e->L[i].dw1p;
            m.dw2 = grads ? grads + l.w2 : nullptr; m.dw3 = grads ? grads + l.w3 : nullptr;
            m.dw2t = e->L[i].dw2t;
            m.half_kind = 0; m.s2 = l.s * l.s; m.Cp = ORN_FAST_C; m.wb = m.wd = nullptr; m.biasp = nullptr;
            // the merged kernel's 16-bit operand copies: rider work-groups of the first block's launch (forward()) when that
            // launch exists, else (and always on the side stream) the epilogue of the merge's S product
            if (i >= e->ff && half_in_S) {
                m.half_kind = d->precision; m.wb = e->L[i].wb; m.wd = e->L[i].wd; m.biasp = e->L[i].biasp;
            }
        };
        // set 0: every layer; set 1: all but the last block (caller's stream of the pipelined step); set 2: the last block (side stream)
        const int nl = d->n_layers;
        for (int k = 0; k < (e->pipe_ok ? 3 : 1); ++k) {
            orn_engine::MergeSet &ms = e->mset[k];
            const int i0 = k == 2 ? nl - 1 : 0, i1 = k == 1 ? nl - 1 : nl;
            ms.n = 0;
            void *bufs[ORN_MAX_LAYERS];
            for (int i = i0; i < i1; ++i) {
                fill(i, ms.ml[ms.n], k == 2 || !e->stage0);
                bufs[ms.n] = e->L[i].mh16;
                ms.layer[ms.n++] = i;
            }
            int rc = orn_merge_groups_build(ms.tables, ms.n, ms.ml, d->precision != 0);
            if (rc == 0 && d->precision != 0) {
                ms.mh_host = malloc(orn_merge_h16_host_bytes());
                rc = ms.mh_host ? orn_merge_h16_build(ms.mh_tables, ms.mh_host, ms.n, ms.ml, bufs, k == 2 ? e->sc_side : e->sc) : ORN_E_ARG;
            }
            if (rc != 0) { orn_engine_destroy(e); return rc; }
            for (int q = 0; q < 4; ++q) ms.tiles[q] = orn_merge_group_tiles(q, ms.n, ms.ml);
        }
    }
    *out = e;
    return 0;
}

extern "C" void orn_engine_destroy(orn_engine *e)
{
    if (!e) return;
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    if (e->graph_exec_u) (void)hipGraphExecDestroy(e->graph_exec_u);
    if (e->graph_u) (void)hipGraphDestroy(e->graph_u);
    for (int i = 0; i < 4 * ORN_MAX_LAYERS + 4; ++i)
        if (e->prof_ev[i]) (void)hipEventDestroy(e->prof_ev[i]);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    hipEvent_t evs[5] = {e->ev_fork, e->ev_adam, e->ev_wgrad, e->ev_join, e->ev_below};
    for (int k = 0; k < 5; ++k) if (evs[k]) (void)hipEventDestroy(evs[k]);
    for (int k = 0; k < 3; ++k) free(e->mset[k].mh_host);
    delete e;
}

// Optional 0/1 mask multiplied into the gradients before Adam (null: none).  Changes what a captured step does, so
// the graph cache is dropped.  main_eval.py:213-531 (prune fine-tune): pruned weights keep a zero gradient.
extern "C" int orn_engine_set_grad_mask(orn_engine *e, const float *mask)
{
    ORN_REQUIRE(e, "engine_set_grad_mask: null engine");
    ORN_REQUIRE((uintptr_t)mask % 16 == 0, "engine_set_grad_mask: mask must be 16-byte aligned");
    e->gmask = mask;
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    if (e->graph_exec_u) { (void)hipGraphExecDestroy(e->graph_exec_u); e->graph_exec_u = nullptr; }
    if (e->graph_u) { (void)hipGraphDestroy(e->graph_u); e->graph_u = nullptr; }
    return 0;
}

// orn_loss_target_stats of the frame table the next steps will be given (null removes it).  Changes what a captured step
// does, so the graph cache is dropped.
extern "C" int orn_engine_set_target_stats(orn_engine *e, const float *stats)
{
    ORN_REQUIRE(e, "engine_set_target_stats: null engine");
    ORN_REQUIRE((uintptr_t)stats % 16 == 0, "engine_set_target_stats: stats must be 16-byte aligned");
    e->tstats = stats;
    if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    if (e->graph_exec_u) { (void)hipGraphExecDestroy(e->graph_exec_u); e->graph_exec_u = nullptr; }
    if (e->graph_u) { (void)hipGraphDestroy(e->graph_u); e->graph_u = nullptr; }
    return 0;
}

// Loss-scale state: out8 = {gs, 1/gs, gs_max, flag, skipped, good, backoffs, 0} (host floats; synchronises the device).
extern "C" int orn_engine_scale_state(orn_engine *e, float *out8)
{
    ORN_REQUIRE(e && out8, "engine_scale_state: null pointer");
    OrnScaleState sa[ORN_SCALE_SLOTS];
    hipError_t rc = hipDeviceSynchronize();
    if (rc == hipSuccess) rc = hipMemcpy(sa, e->sc, sizeof(sa), hipMemcpyDeviceToHost);
    if (rc != hipSuccess) { orn_set_error("engine_scale_state: %s", hipGetErrorString(rc)); return (int)rc; }
    const OrnScaleState &s = sa[0];
    int any = 0;                                      // flag: a step since the last advance met a non-finite value
    for (int r = 0; r < ORN_SCALE_SLOTS; ++r) any |= sa[r].flag;
    out8[0] = s.gs; out8[1] = s.inv_gs; out8[2] = s.gs_max; out8[3] = (float)any; out8[4] = (float)s.skipped;
    out8[5] = (float)s.good; out8[6] = (float)s.backoffs; out8[7] = 0.f;
    return 0;
}

// Overrides the live gradient scale (tests: a scale far too large makes the 16-bit gradients overflow; tuning).  gs_max > 0
// also replaces the ceiling the scale may grow back to.
extern "C" int orn_engine_set_grad_scale(orn_engine *e, float gs, float gs_max)
{
    ORN_REQUIRE(e && gs >= 1.0f, "engine_set_grad_scale: scale must be >= 1");
    OrnScaleState sa[ORN_SCALE_SLOTS];
    hipError_t rc = hipDeviceSynchronize();
    if (rc == hipSuccess) rc = hipMemcpy(sa, e->sc, sizeof(sa), hipMemcpyDeviceToHost);
    for (int r = 0; r < ORN_SCALE_SLOTS; ++r) {
        sa[r].gs = gs; sa[r].inv_gs = 1.0f / gs;
        if (gs_max > 0.f) sa[r].gs_max = gs_max;
    }
    sa[0].good = 0;
    if (rc == hipSuccess) rc = hipMemcpy(e->sc, sa, sizeof(sa), hipMemcpyHostToDevice);
    if (rc != hipSuccess) { orn_set_error("engine_set_grad_scale: %s", hipGetErrorString(rc)); return (int)rc; }
    return 0;
}

extern "C" int orn_engine_fused_kernel(orn_engine *e, int layer, const float **wf, const float **bf)
{
    ORN_REQUIRE(e && wf && bf && layer >= 0 && layer < e->d.n_layers, "engine_fused_kernel: bad arguments");
    *wf = e->L[layer].wf;
    *bf = e->L[layer].bf;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// `count` consecutive steps at once (thread j -> cur[j]): the unrolled graph advances once for all its steps
__global__ void k_advance(const orn_step_sched *__restrict__ sched, int32_t *cursor, int32_t n_slots, double beta1,
                          double beta2, OrnStepCur *cur_all, int count, OrnScaleState *sc)
{
    const int32_t c0 = *cursor;
    __shared__ int32_t skipped;
    if (threadIdx.x == 0) {
        // the steps of the group that has just run: any overflow halves the scale (once per advance); clean groups are credited
        // now that they HAVE run, and ORN_SCALE_GROWTH_INTERVAL clean steps double the scale again, up to its initial value
        int nflag = 0;
        for (int r = 0; r < ORN_SCALE_SLOTS; ++r)
            if (sc[r].flag) { nflag += 1; sc[r].flag = 0; }
        float gs = sc->gs;
        if (nflag) {
            gs = fmaxf(gs * 0.5f, 1.0f);
            sc->good = 0; sc->backoffs += 1;
        } else {
            sc->good += sc->launched;
            if (sc->good >= ORN_SCALE_GROWTH_INTERVAL && gs < sc->gs_max) { gs *= 2.0f; sc->good = 0; }
        }
        sc->launched = count;
        for (int r = 0; r < ORN_SCALE_SLOTS; ++r) { sc[r].gs = gs; sc[r].inv_gs = 1.0f / gs; }
        skipped = sc->skipped;
    }
    __syncthreads();
    if (blockIdx.x == 0 && (int)threadIdx.x < count) {
        const int32_t c = c0 + threadIdx.x;
        OrnStepCur *cur = cur_all + threadIdx.x;
        orn_step_sched s = sched[c];
        s.step = max(s.step - skipped, 1);              // torch: a skipped optimizer.step() does not advance Adam's step count
        cur->frame = s.frame;
        cur->step = s.step;
        cur->lr = s.lr;
        // torch.optim.Adam: bias corrections in double, rounded to fp32 once
        const double bc1 = 1.0 - pow(beta1, (double)s.step), bc2 = 1.0 - pow(beta2, (double)s.step);
        cur->step_size = (float)((double)s.lr / bc1);
        cur->sqrt_bc2 = (float)sqrt(bc2);
        cur->slot = n_slots > 0 ? c % n_slots : 0;
        if (threadIdx.x == 0) *cursor = c0 + count;
    }
}

// fp32 engine: the last block's backward starts from the head's (out, dout) and reads z (k_head_bwd_fused_f32), so its training
// forward needs no stored activation either
static bool f32_head_on_z(const orn_engine *e)
{
    static const bool no_head_fuse = orn_probe_env("ORN_F32_HEAD_UNFUSED") != nullptr;       // tools/probes A/B
    const orn_engine_desc &d = e->d;
    return e->ff >= d.n_layers && e->head_ws && d.layer[d.n_layers - 1].s == 2 && !no_head_fuse;
}

// set: which layers this forward merges (0 all; 1: all but the last block, whose merged kernel the side branch of the previous
// pipelined step leaves behind -- the forward then waits for that branch where it touches the last block's buffers)
static int forward(orn_engine *e, const float *embeds, const int *row_idx, bool keep_z, hipStream_t st, int set = 0)
{
    const orn_engine_desc &d = e->d;
    const int Nout = d.fc_h * d.fc_w * d.fc_dim;
    float *P = e->params;
    const OrnLinearJob lin1 = {embeds, row_idx, (size_t)d.embed_len, P + d.stem_w0, P + d.stem_b0, 1, d.embed_len, d.stem_dim, e->pre1, e->h1};
    const OrnLinearJob lin2 = {e->h1, nullptr, 0, P + d.stem_w1, P + d.stem_b1, 1, d.stem_dim, Nout, e->pre2, e->h2};
    const float *x = e->h2;
    const int nl = d.n_layers, ff = e->ff;
    const void *pack_t = nullptr;
    int pack_t_blocks = 0;
    if (d.erb) {
        // online re-parameterisation of every layer (model.py:534): weights only, so all layers up front -- two grouped
        // launches (T, then S with the bias merge b3x3 + (b1x3 + b3x1) in its first tile column), each carrying one of
        // the stem's two linear layers as extra work-groups
        // half operand copies for the merge BACKWARD (training step, 16-bit modes) as riders: the parameter-side ones behind
        // the W2 transposes, T -> Th behind the first block's launch (or, without it, behind the S products)
        const orn_engine::MergeSet &ms = e->mset[set];
        int pk_par = 0, pk_t = 0;
        const void *pk = (keep_z && ms.mh_host) ? orn_merge_h16_pack(ms.mh_host, &pk_par, &pk_t) : nullptr;
        ORN_TRY(orn_launch_w2_transpose(ms.n, ms.ml, st, pk, pk_par));
        ORN_TRY(orn_launch_merge_group_linear(ms.tables, 0, ms.tiles[0], lin1, st));
        ORN_TRY(orn_launch_merge_group_linear(ms.tables, 1, ms.tiles[1], lin2, st, e->stage0 ? nullptr : pk, e->stage0 ? 0 : pk_t));
        pack_t = e->stage0 ? pk : nullptr; pack_t_blocks = e->stage0 ? pk_t : 0;
    } else {
        ORN_TRY(orn_launch_linear_silu(lin1.x, lin1.row_idx, lin1.row_stride, lin1.w, lin1.bias, 1, lin1.K, lin1.N, lin1.pre, lin1.y, st));
        ORN_TRY(orn_launch_linear_silu(lin2.x, nullptr, 0, lin2.w, lin2.bias, 1, lin2.K, lin2.N, lin2.pre, lin2.y, st));
    }
    if (ff < nl && !d.erb) {      // 16-bit operand copies of every fast layer's kernel, one launch (ERB: written by the merge's S launch)
        OrnPrepLayer pl[ORN_MAX_LAYERS];
        const int np = (set == 1 ? nl - 1 : nl) - ff;       // (pipelined step: the side branch prepares the last block's)
        for (int i = ff; i < ff + np; ++i) {
            const orn_layer_desc &l = d.layer[i];
            pl[i - ff] = OrnPrepLayer{e->L[i].wf, e->L[i].bf, l.O, l.C, l.s, e->L[i].wb, e->L[i].wd, e->L[i].biasp, ORN_FAST_C};
        }
        ORN_TRY(e->ops->prep_all(np, pl, st));
    }
    OrnHeadFuse hf = {};
    for (int i = 0; i < nl; ++i) {
        const orn_layer_desc &l = d.layer[i];
        LayerBuf &b = e->L[i];
        if (i < ff) {
            if (e->prof) (void)hipEventRecord(e->prof_ev[2 * i], st);
            if (e->stage0) { // conv + PixelShuffle + SiLU straight into layer 1's 16-bit input (b.z in the fused pair's own layout)
                // ERB: the 16-bit operand copies of the later blocks' merged kernels ride on this launch (orn_prep_rider.h)
                OrnPrepLayer pl[ORN_MAX_LAYERS];
                int np = 0;
                if (d.erb)
                    for (int j = ff; j < (set == 1 ? nl - 1 : nl); ++j) {
                        const orn_layer_desc &lj = d.layer[j];
                        pl[np++] = OrnPrepLayer{e->L[j].wf, e->L[j].bf, lj.O, lj.C, lj.s, e->L[j].wb, e->L[j].wd, e->L[j].biasp, ORN_FAST_C};
                    }
                ORN_TRY(orn_launch_stage0_fwd(x, b.wf, b.bf, l.C, l.O, l.H, l.W, l.s, keep_z ? b.z : nullptr, e->L[1].xpad, ORN_FAST_C,
                                              d.precision, st, np, pl, pack_t, pack_t_blocks));
            }
            else
            {
                // fp32 engine, training step, last block: nothing but the head reads its activation, and the head's backward is fused with
                // this block's (it reads z): store z only and let the head's forward form SiLU(z) (354 MB less written at 720p)
                const bool z_only = keep_z && i == nl - 1 && f32_head_on_z(e);
                ORN_TRY(orn_launch_conv3x3_f32(x, b.wf, b.bf, 1, l.C, l.O, l.H, l.W, l.s, 1, keep_z ? b.z : nullptr, z_only ? nullptr : b.a, st, nullptr));
            }
            if (e->prof) (void)hipEventRecord(e->prof_ev[2 * i + 1], st);
            x = b.a;
        } else {
            if (i == ff && !e->stage0) ORN_TRY(e->ops->to_nhwc(x, l.C, ORN_FAST_C, l.H, l.W, b.xpad, st));
            if (e->prof) (void)hipEventRecord(e->prof_ev[2 * i], st);
            // the last block's kernel may run the head in its epilogue (it then holds every channel of an output pixel)
            const bool last = (i + 1 == nl);
            // pipelined step: the side branch of the previous step still reads the last block's input buffer (its weight gradient)
            // until ev_wgrad, and writes that block's merged kernel (and the head's parameters) until ev_join
            if (e->side_busy && i + 3 == nl && side_takes_below(e)) ORN_HIP(hipStreamWaitEvent(st, e->ev_below, 0));   // (this conv overwrites the input of the block below the last)
            if (e->side_busy && i + 2 == nl) ORN_HIP(hipStreamWaitEvent(st, e->ev_wgrad, 0));
            if (e->side_busy && last) { ORN_HIP(hipStreamWaitEvent(st, e->ev_join, 0)); e->side_busy = false; }
            if (last) hf = OrnHeadFuse{P + d.head_w, P + d.head_b, e->img, d.sigmoid, 0};
            ORN_TRY(e->ops->conv_fwd(b.xpad, b.wb, b.biasp, l.H, l.W, ORN_FAST_C, l.O, l.s, b.zb, last ? nullptr : e->L[i + 1].xpad, st, l.C,
                                     last ? &hf : nullptr));
            if (e->prof) (void)hipEventRecord(e->prof_ev[2 * i + 1], st);
        }
    }
    if (ff < nl) {
        if (!hf.fused)
            ORN_TRY(e->ops->head_fwd(e->L[nl - 1].zb, P + d.head_w, P + d.head_b, e->Cn_last, (size_t)e->Hout * e->Wout, d.sigmoid, e->img, st));
    } else
    {
        const bool z_only = keep_z && f32_head_on_z(e);
        ORN_TRY(orn_launch_head_fwd(z_only ? e->L[nl - 1].z : x, P + d.head_w, P + d.head_b, 1, e->Cn_last, (size_t)e->Hout * e->Wout, d.sigmoid, e->img,
                                    st, z_only));
    }
    return 0;
}

extern "C" int orn_engine_decode(orn_engine *e, const float *embed, float *img, void *stream)
{
    ORN_REQUIRE(e && embed && img, "engine_decode: null pointer");
    hipStream_t st = (hipStream_t)stream;
    ORN_TRY(forward(e, embed, nullptr, false, st));
    hipError_t rc = hipMemcpyAsync(img, e->img, (size_t)3 * e->Hout * e->Wout * 4, hipMemcpyDeviceToDevice, st);
    if (rc != hipSuccess) { orn_set_error("engine_decode: copy failed: %s", hipGetErrorString(rc)); return (int)rc; }
    return 0;
}

// adv_count > 0: advance the device-side schedule by that many steps first (states e->cur[0 .. adv_count)); the step itself
// runs on e->cur[cur_idx]
static OrnMergeMisc merge_misc(const orn_engine *e, int i)
{
    const orn_layer_desc &l = e->d.layer[i];
    float *G = e->grads;
    OrnMergeMisc m = {};
    m.C = l.C; m.O = l.O;
    m.g = G + l.w3x3; m.dbf = G + l.b3x3; m.dw1p = e->L[i].dw1p;
    m.d3x1 = G + l.w3x1; m.db3x1 = G + l.b3x1; m.d1x3 = G + l.w1x3; m.db1x3 = G + l.b1x3;
    m.dw1 = G + l.w1;
    if (e->d.precision != 0) { m.dw2t = e->L[i].dw2t; m.dw2 = G + l.w2; }
    return m;
}

// Split-K slabs of the LAST block's weight gradient on an engine that can run the pipelined step: fewer, longer work-groups (one
// per CU, or little more) leave room on every CU for the launches that run beside it.  Measured on the 720p step (tools/probes, round 4):
// 40 / 32 / 24 slabs = 1.050 / 1.015..1.024 / 1.019..1.023 ms per pipelined step in the final form of the branch (32 and 24
// are equal within the noise of a box, 24 moves less data).  The serial forms of the step use the same count, so that all forms of
// the step give bit-identical results.
static int last_smax(const orn_engine *e)
{
    static const int smax = orn_probe_env_int("ORN_SIDE_SMAX", 24);
    return e->pipe_ok ? smax : 0;
}

// Where the side branch forks off the backward: behind the dgrad launch of this layer (n_layers: behind the head's backward);
// -1: behind the lower blocks' weight gradients and their reduction, in front of the merge backward.  Default: behind the dgrad of
// the block below the last one.  Measured on the 720p step, one box (tools/probes/mode_ab.sh; serial 1.079 ms): fork behind layer
// 5 (the head) 1.038, 4 1.058, 3 1.024, 2 1.032, 1 1.035, 0 1.067, -1 1.050 ms -- the last block's own dgrad and the launch behind
// it are full-chip MFMA launches that the side branch's weight gradient only thrashes; behind them the caller's stream runs
// under-filled launches, and the earlier the branch starts the earlier it is back for the next forward.
// The side stream also takes the weight gradient of the block below the last one (ahead of the last block's), and then forks one
// launch earlier, behind the last block's dgrad: that wgrad (216 work-groups at 720p) runs beside the block's own dgrad launch (230
// work-groups on 512 slots), and the lower blocks' batched wgrad on the caller's stream shrinks to a third (95 -> 40 us at 720p).
// Same box: 1.037 -> 1.021 ms per step.  ORN_SIDE_BELOW=0: tools/probes A/B.
static bool side_takes_below(const orn_engine *e)
{
    static const bool on = orn_probe_env_int("ORN_SIDE_BELOW", 1) != 0;
    return on && e->pipe_ok && e->d.n_layers - 2 > e->ff;
}

static int below_smax(const orn_engine *e)
{
    static const int smax = orn_probe_env_int("ORN_BELOW_SMAX", 0);      // probe: slab cap of the block below the last (0: the default rule)
    return side_takes_below(e) ? smax : 0;
}

static int side_fork_at(const orn_engine *e)
{
    static const int at = orn_probe_env_int("ORN_SIDE_FORK", -2);
    if (at == -2 && side_takes_below(e)) return e->d.n_layers - 1;
    if (at == -2) return e->d.n_layers - 2 > e->ff ? e->d.n_layers - 2 : -1;      // default
    if (side_takes_below(e) && at > e->d.n_layers - 1) return e->d.n_layers - 1;      // (that wgrad needs the last block's dgrad output)
    return at > e->d.n_layers ? e->d.n_layers : at;
}

// The side branch of a pipelined step, first half (enqueued at the fork point, side_fork_at): the weight gradient of the block below the
// last one (side_takes_below), then the last block's weight gradient (+ the head's dW / db finish), its slab reduction, its merge backward.
static int side_branch_backward(orn_engine *e, hipStream_t st)
{
    const orn_engine_desc &d = e->d;
    const int nl = d.n_layers;
    const orn_layer_desc &l = d.layer[nl - 1];
    float *G = e->grads;
    hipStream_t sd = e->side;
    OrnScaleState *sc = e->sc_side;                     // this step's scale, copied by the loss's finalize stage; late detections
    ORN_HIP(hipEventRecord(e->ev_fork, st));
    ORN_HIP(hipStreamWaitEvent(sd, e->ev_fork, 0));
    if (side_takes_below(e)) {      // (its dy is the output of the dgrad launch this branch forks behind; its slabs are reduced on the caller's stream)
        const orn_layer_desc &lb = d.layer[nl - 2];
        const OrnWgradJob wb = {e->L[nl - 2].xpad, e->L[nl - 2].dypad, lb.H, lb.W, lb.C, lb.O, lb.s, e->L[nl - 2].wslab, below_smax(e)};
        ORN_TRY(e->ops->wgrad_batch(1, &wb, sd, nullptr, nullptr, 1));
        ORN_HIP(hipEventRecord(e->ev_below, sd));
    }
    const OrnWgradJob wj = {e->L[nl - 1].xpad, e->L[nl - 1].dypad, l.H, l.W, l.C, l.O, l.s, e->L[nl - 1].wslab, last_smax(e)};
    const OrnHeadFinish hf = {e->head_ws, e->ops->head_bwd_blocks(e->Hout, e->Wout), e->Cn_last, 1.0f / e->gs, G + d.head_w, G + d.head_b, sc};
    ORN_TRY(e->ops->wgrad_batch(1, &wj, sd, &hf, nullptr, 1));
    ORN_HIP(hipEventRecord(e->ev_wgrad, sd));           // the last block's input buffer is free again
    const OrnWgradReduce wr = {e->L[nl - 1].wslab, l.H, l.W, l.C, l.O, l.s, 1.0f / e->gs, G + l.w3x3, G + l.b3x3, sc, last_smax(e)};
    ORN_TRY(e->ops->wgrad_reduce_all(1, &wr, sd, nullptr));
    if (d.erb) {
        const orn_engine::MergeSet &ms = e->mset[2];
        ORN_TRY(orn_launch_merge_h16_bwd(ms.mh_tables, ms.mh_host, sd, sc));
        const OrnMergeMisc mm = merge_misc(e, nl - 1);
        ORN_TRY(orn_launch_merge_bwd_tail_all(1, &mm, sd));
    }
    return 0;
}

// Second half (enqueued behind the caller's Adam launch, whose skip decision it follows): Adam over the last block's and the head's
// parameters, then the last block's merge forward (or, without ERB, its 16-bit operand copies) for the NEXT step.
// more: another step follows in this call (the last step of a call skips the merge forward: the next call's first step merges every
// block itself, the parameters may have been touched in between)
static int side_branch_update(orn_engine *e, bool more)
{
    const orn_engine_desc &d = e->d;
    const int nl = d.n_layers;
    hipStream_t sd = e->side;
    const size_t lo = e->side_lo;
    ORN_HIP(hipStreamWaitEvent(sd, e->ev_adam, 0));
    ORN_TRY(orn_launch_adam(e->params + lo, e->grads + lo, e->m + lo, e->v + lo, (size_t)d.n_params - lo, 0.0, 1, e->cur_side, d.beta1, d.beta2,
                            d.eps, 1.0f, sd, e->gmask ? e->gmask + lo : nullptr, e->sc_side, nullptr, nullptr, false));
    if (!more) {
    } else if (d.erb) {
        const orn_engine::MergeSet &ms = e->mset[2];
        int pk_par = 0, pk_t = 0;
        const void *pk = ms.mh_host ? orn_merge_h16_pack(ms.mh_host, &pk_par, &pk_t) : nullptr;
        const OrnLinearJob none = {};
        // The pack jobs that ride on these launches in the serial step (half copies for the NEXT merge backward of this block: needed
        // a whole step later) are launches of their own BEHIND the join event here: the branch's S launch runs while the 8-wave forward
        // conv of the block below holds all but a few CUs, and 160 rider work-groups in front of its 84 tiles cost it those.
        static const bool late_packs = orn_probe_env("ORN_SIDE_RIDERS") == nullptr;
        ORN_TRY(orn_launch_w2_transpose(ms.n, ms.ml, sd, late_packs ? nullptr : pk, late_packs ? 0 : pk_par));
        ORN_TRY(orn_launch_merge_group_linear(ms.tables, 0, ms.tiles[0], none, sd));
        ORN_TRY(orn_launch_merge_group_linear(ms.tables, 1, ms.tiles[1], none, sd, late_packs ? nullptr : pk, late_packs ? 0 : pk_t));   // the S epilogue writes the 16-bit operand copies
        if (late_packs && pk) {
            ORN_HIP(hipEventRecord(e->ev_join, sd));
            ORN_TRY(orn_launch_merge_h16_pack_jobs(ms.mh_host, 1, sd));
            ORN_TRY(orn_launch_merge_h16_pack_jobs(ms.mh_host, 2, sd));
            e->side_busy = true;
            return 0;
        }
    } else {
        const orn_layer_desc &l = d.layer[nl - 1];
        const OrnPrepLayer pl = {e->L[nl - 1].wf, e->L[nl - 1].bf, l.O, l.C, l.s, e->L[nl - 1].wb, e->L[nl - 1].wd, e->L[nl - 1].biasp, ORN_FAST_C};
        ORN_TRY(e->ops->prep_all(1, &pl, sd));
    }
    ORN_HIP(hipEventRecord(e->ev_join, sd));
    e->side_busy = true;
    return 0;
}

// pipe: the pipelined form (struct orn_engine, `side`): the last block's weight-gradient chain leaves this step on the side stream
static int train_step(orn_engine *e, const float *frames, const float *embeds, const orn_step_sched *sched,
                      int32_t *cursor, float *stats_out, int32_t n_slots, hipStream_t st, int adv_count = 1, int cur_idx = 0,
                      bool pipe = false, bool more = false)
{
    const orn_engine_desc &d = e->d;
    float *P = e->params, *G = e->grads;
    const int Nout = d.fc_h * d.fc_w * d.fc_dim;
    const size_t HWo = (size_t)e->Hout * e->Wout;
    OrnStepCur *cur = e->cur + cur_idx;
    if (adv_count > 0) {
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, sched, cursor, n_slots, d.beta1, d.beta2, e->cur, adv_count, e->sc /* all entries */);
        ORN_LAUNCH_CHECK("advance");
    }
    OrnScaleState *const sc = e->sc + cur_idx;          // this step's entry: its own flag, the shared scale
    const int *fidx = &cur->frame;
    ORN_TRY(forward(e, embeds, fidx, true, st, (pipe && e->side_busy) ? 1 : 0));
    const int nl = d.n_layers, ff = e->ff;
    OrnLossFinalJob fin = {};
    ORN_TRY(orn_launch_loss(e->img, frames, fidx, 3 * HWo, 1, 3, e->Hout, e->Wout, d.loss_type, 1.0f, e->stats, e->dimg,
                            e->loss_ws, st, cur, stats_out, sc, d.loss_type == ORN_LOSS_FUSION6 ? e->tstats : nullptr,
                            ff < nl ? &fin : nullptr));     // 16-bit engine: the finalize stage rides on the head's backward launch
    if (pipe) { fin.cur_copy = e->cur_side; fin.sc_copy = e->sc_side; }     // (the previous step's side branch was joined in forward())
    if (ff < nl)
        ORN_TRY(e->ops->head_bwd(e->L[nl - 1].zb, P + d.head_w, e->img, e->dimg, e->Cn_last, e->Hout, e->Wout, d.sigmoid,
                                 d.layer[nl - 1].s, e->gs, e->L[nl - 1].dypad, nullptr, nullptr, e->head_ws, st, sc, &fin));   // dW / db: finished with the wgrad batch
    const bool head_fused32 = f32_head_on_z(e);
    const OrnHeadBwdFuse hfuse = {P + d.head_w, e->img, e->dimg, d.sigmoid, G + d.head_w, G + d.head_b, e->head_ws};
    if (ff >= nl && !head_fused32)
        ORN_TRY(orn_launch_head_bwd(e->L[nl - 1].a, P + d.head_w, e->img, e->dimg, 1, e->Cn_last, HWo, d.sigmoid, e->L[nl - 1].da,
                                    G + d.head_w, G + d.head_b, e->scratch, st));
    if (ff >= nl) {
        // fp32 engine: every layer's dgrad kernel (flipped taps, transposed) in one launch instead of one per layer
        const float *wfs[ORN_MAX_LAYERS]; float *wds[ORN_MAX_LAYERS]; int Os[ORN_MAX_LAYERS], Cs[ORN_MAX_LAYERS];
        for (int i = 0; i < nl; ++i) { wfs[i] = e->L[i].wf; wds[i] = e->L[i].wd32; Os[i] = d.layer[i].O; Cs[i] = d.layer[i].C; }
        ORN_TRY(orn_launch_flip_transpose_all(nl, wfs, wds, Os, Cs, st));
    }
    if (pipe && side_fork_at(e) >= nl) ORN_TRY(side_branch_backward(e, st));
    for (int i = nl - 1; i >= 0; --i) {
        const orn_layer_desc &l = d.layer[i];
        LayerBuf &b = e->L[i];
        const float *x = (i == 0) ? e->h2 : e->L[i - 1].a;
        float *dx = (i == 0) ? e->dh2 : e->L[i - 1].da;
        // dWf / dbf land directly in the 3x3 branch's gradient slots (dW3x3 = dWf, db3x3 = dbf)
        if (e->prof) (void)hipEventRecord(e->prof_ev[2 * ORN_MAX_LAYERS + 2 * i], st);
        if (i >= ff) {
            // (the wgrads of all fast layers run as one multi-problem launch after the dgrad chain, see below)
            if (i > ff) {
                // few pixel tiles: input-chunk split through fp32 partial slabs in the scratch, finished into dypad
                float *part = e->ops->dgrad_f32_slabs(l.H, l.W, l.O) > 1 ? e->scratch : nullptr;
                ORN_TRY(e->ops->conv_dgrad(b.dypad, b.wd, l.H, l.W, l.O, l.C, e->L[i - 1].zb, e->L[i - 1].dypad, d.layer[i - 1].s,
                                           part, st, l.C));
            } else {
                ORN_TRY(e->ops->conv_dgrad(b.dypad, b.wd, l.H, l.W, l.O, ORN_FAST_C, nullptr, nullptr, 1, e->dxn, st, l.C));
                if (!e->stage0)
                ORN_TRY(e->ops->to_nchw_f32(e->dxn, l.C, ORN_FAST_C, l.H, l.W, e->ops->dgrad_f32_slabs(l.H, l.W, l.O), 1.0f / e->gs, dx,
                                            st, sc));
            }
        } else if (e->stage0) {
            const orn_layer_desc &l1 = d.layer[1];
            ORN_TRY(orn_launch_stage0_bwd(x, b.wf, b.z, e->dxn, e->ops->dgrad_f32_slabs(l1.H, l1.W, l1.O), ORN_FAST_C, 1.0f / e->gs, l.C, l.O,
                                          l.H, l.W, l.s, e->scratch + e->stem_ws, nullptr, G + l.w3x3, G + l.b3x3, st, sc));
        } else
        ORN_TRY(orn_launch_conv_bwd_f32(x, b.wf, b.z, b.da, 1, l.C, l.O, l.H, l.W, l.s, dx, G + l.w3x3, G + l.b3x3, e->scratch, st,
                                        (i == nl - 1 && head_fused32) ? &hfuse : nullptr, ff >= nl ? b.wd32 : nullptr));
        if (e->prof) (void)hipEventRecord(e->prof_ev[2 * ORN_MAX_LAYERS + 2 * i + 1], st);
        if (pipe && i == side_fork_at(e)) ORN_TRY(side_branch_backward(e, st));
    }
    // stem backward; with a wgrad batch behind it, its last kernel (needed by Adam only) rides along that launch
    OrnStemW0Job w0job;
    OrnStemL2Job l2job;
    const bool defer_w0 = ff < nl;
    ORN_TRY(orn_launch_stem_bwd(embeds, fidx, d.embed_len, P + d.stem_w1, e->pre1, e->h1, e->pre2,
                                e->stage0 ? e->scratch + e->stem_ws : e->dh2, 1, d.embed_len, d.stem_dim, Nout, G + d.stem_w0,
                                G + d.stem_b0, G + d.stem_w1, G + d.stem_b1, e->scratch, st,
                                e->stage0 ? orn_stage0_slabs(d.layer[0].O, d.layer[0].s) : 1, defer_w0 ? &w0job : nullptr, defer_w0 ? &l2job : nullptr));
    if (ff < nl) {
        // Weight gradients of every fast layer: nothing on the dgrad chain needs them, so they run here as ONE launch (the
        // small layers' 72 / 216 / 360 work-groups pack behind the last block's 504 instead of leaving CUs idle one launch
        // at a time: -37 us per 720p step), followed by ONE launch that reduces every layer's split-K slabs (-27 us).
        OrnWgradJob wj[ORN_MAX_LAYERS];
        int nj = 0;
        const int n_main = pipe ? nl - 1 : nl;      // (pipelined step: the last block's is the side branch's)
        for (int i = n_main - 1; i >= ff; --i) {    // largest first
            if (pipe && i == nl - 2 && side_takes_below(e)) continue;      // (the side stream's)
            const orn_layer_desc &l = d.layer[i];
            wj[nj++] = OrnWgradJob{e->L[i].xpad, e->L[i].dypad, l.H, l.W, l.C, l.O, l.s, e->L[i].wslab, (i == nl - 1) ? last_smax(e) : (i == nl - 2 ? below_smax(e) : 0)};
        }
        const OrnHeadFinish hf = {e->head_ws, e->ops->head_bwd_blocks(e->Hout, e->Wout), e->Cn_last, 1.0f / e->gs, G + d.head_w, G + d.head_b, sc};
        if (e->prof) (void)hipEventRecord(e->prof_ev[4 * ORN_MAX_LAYERS], st);
        ORN_TRY(e->ops->wgrad_batch(nj, wj, st, pipe ? nullptr : &hf, &l2job, 0));
        if (e->prof) (void)hipEventRecord(e->prof_ev[4 * ORN_MAX_LAYERS + 1], st);
        OrnWgradReduce wr[ORN_MAX_LAYERS];
        for (int i = ff; i < n_main; ++i) {
            const orn_layer_desc &l = d.layer[i];
            wr[i - ff] = OrnWgradReduce{e->L[i].wslab, l.H, l.W, l.C, l.O, l.s, 1.0f / e->gs, G + l.w3x3, G + l.b3x3, sc, (i == nl - 1) ? last_smax(e) : (i == nl - 2 ? below_smax(e) : 0)};
        }
        if (e->prof) (void)hipEventRecord(e->prof_ev[4 * ORN_MAX_LAYERS + 2], st);
        if (pipe && side_takes_below(e)) ORN_HIP(hipStreamWaitEvent(st, e->ev_below, 0));
        ORN_TRY(e->ops->wgrad_reduce_all(n_main - ff, wr, st, &w0job));
        if (e->prof) (void)hipEventRecord(e->prof_ev[4 * ORN_MAX_LAYERS + 3], st);
    }
    if (pipe && side_fork_at(e) < 0) ORN_TRY(side_branch_backward(e, st));     // fork: the last block's wgrad .. merge backward, on the side stream
    if (d.erb) {
        // merge backward of every layer (closed forms): dW3 & dT, then dW2 & dW1, then the slices
        const orn_engine::MergeSet &ms = e->mset[pipe ? 1 : 0];
        if (d.precision != 0) {
            ORN_TRY(orn_launch_merge_h16_bwd(ms.mh_tables, ms.mh_host, st, sc));
        } else {
            ORN_TRY(orn_launch_merge_group(ms.tables, 2, ms.tiles[2], st));
            ORN_TRY(orn_launch_merge_group(ms.tables, 3, ms.tiles[3], st));
        }
        OrnMergeMisc mm[ORN_MAX_LAYERS];
        for (int k = 0; k < ms.n; ++k) mm[k] = merge_misc(e, ms.layer[k]);
        ORN_TRY(orn_launch_merge_bwd_tail_all(ms.n, mm, st));
    }
    if (pipe) {
        // this stream's Adam launch covers everything below the last block; its skip decision is mirrored for the side branch's launch
        ORN_TRY(orn_launch_adam(P, G, e->m, e->v, e->side_lo, 0.0, 1, cur, d.beta1, d.beta2, d.eps, 1.0f, st, e->gmask, sc, e->sc, e->sc_side));
        ORN_HIP(hipEventRecord(e->ev_adam, st));
        return side_branch_update(e, more);
    }
    ORN_TRY(orn_launch_adam(P, G, e->m, e->v, (size_t)d.n_params, 0.0, 1, cur, d.beta1, d.beta2, d.eps, 1.0f, st, e->gmask, sc, e->sc));
    return 0;
}

extern "C" int orn_engine_train_step(orn_engine *e, const float *frames, const float *embeds,
                                     const orn_step_sched *sched, int32_t *cursor, float *stats_out, int32_t n_slots,
                                     void *stream)
{
    ORN_REQUIRE(e && frames && embeds && sched && cursor, "engine_train_step: null pointer");
    ORN_REQUIRE(e->grads && e->m && e->v, "engine_train_step: engine was created without grads / Adam arenas");
    return train_step(e, frames, embeds, sched, cursor, stats_out, n_slots, (hipStream_t)stream);
}

// n_steps steps enqueued on `stream` (no graph): where the engine can (struct orn_engine, `side`) in the pipelined form, the
// last block's weight-gradient chain of step k running on the engine's own second stream beside the boundary of steps k / k + 1.
// Everything is joined back into `stream` before the call returns, so the caller's stream order covers all of it.
extern "C" int orn_engine_train_steps(orn_engine *e, const float *frames, const float *embeds, const orn_step_sched *sched,
                                      int32_t *cursor, float *stats_out, int32_t n_slots, int32_t n_steps, void *stream)
{
    ORN_REQUIRE(e && frames && embeds && sched && cursor && n_steps >= 0, "engine_train_steps: bad arguments");
    ORN_REQUIRE(e->grads && e->m && e->v, "engine_train_steps: engine was created without grads / Adam arenas");
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    for (int k = 0; k < n_steps && rc == 0; ++k) rc = train_step(e, frames, embeds, sched, cursor, stats_out, n_slots, st, 1, 0, e->pipe_ok, k + 1 < n_steps);
    if (e->side_busy) {
        const hipError_t hrc = hipStreamWaitEvent(st, e->ev_join, 0);
        e->side_busy = false;
        if (hrc != hipSuccess && rc == 0) { orn_set_error("engine_train_steps: join: %s", hipGetErrorString(hrc)); rc = (int)hrc; }
    }
    return rc;
}

// One EAGER training step with HIP events around every layer's forward conv launch, on the launch stream:
// ms_out[i] = duration of layer i's forward conv kernel inside a real step (bench.py's roofline leg).  Synchronises.
extern "C" int orn_engine_profile_step(orn_engine *e, const float *frames, const float *embeds, const orn_step_sched *sched,
                                       int32_t *cursor, float *stats_out, int32_t n_slots, float *ms_out, void *stream)
{
    ORN_REQUIRE(e && frames && embeds && sched && cursor && ms_out, "engine_profile_step: null pointer");
    ORN_REQUIRE(e->grads && e->m && e->v, "engine_profile_step: engine was created without grads / Adam arenas");
    hipStream_t st = (hipStream_t)stream;
    const int nl = e->d.n_layers;
    for (int i = 0; i < 4 * ORN_MAX_LAYERS + 4; ++i)
        if (!e->prof_ev[i]) {
            hipError_t rc = hipEventCreate(&e->prof_ev[i]);
            if (rc != hipSuccess) { orn_set_error("engine_profile_step: hipEventCreate: %s", hipGetErrorString(rc)); return (int)rc; }
        }
    e->prof = true;
    const int trc = train_step(e, frames, embeds, sched, cursor, stats_out, n_slots, st);
    e->prof = false;
    if (trc != 0) return trc;
    hipError_t rc = hipStreamSynchronize(st);
    if (rc != hipSuccess) { orn_set_error("engine_profile_step: sync: %s", hipGetErrorString(rc)); return (int)rc; }
    auto span = [&](int a, float *out) -> int {
        float ms = 0.f;
        hipError_t r = hipEventElapsedTime(&ms, e->prof_ev[a], e->prof_ev[a + 1]);
        if (r != hipSuccess) { orn_set_error("engine_profile_step: elapsed: %s", hipGetErrorString(r)); return (int)r; }
        *out = ms;
        return 0;
    };
    const bool fast = e->ff < nl;
    for (int i = 0; i < nl; ++i) {
        ORN_TRY(span(2 * i, &ms_out[i]));
        ORN_TRY(span(2 * ORN_MAX_LAYERS + 2 * i, &ms_out[nl + i]));
    }
    ms_out[2 * nl] = ms_out[2 * nl + 1] = 0.f;
    if (fast) {
        ORN_TRY(span(4 * ORN_MAX_LAYERS, &ms_out[2 * nl]));
        ORN_TRY(span(4 * ORN_MAX_LAYERS + 2, &ms_out[2 * nl + 1]));
    }
    return 0;
}

extern "C" int orn_engine_train_steps_graph(orn_engine *e, const float *frames, const float *embeds,
                                            const orn_step_sched *sched, int32_t *cursor, float *stats_out,
                                            int32_t n_slots, int32_t n_steps, void *stream)
{
    ORN_REQUIRE(e && frames && embeds && sched && cursor && n_steps >= 0, "engine_train_steps_graph: bad arguments");
    ORN_REQUIRE(e->grads && e->m && e->v, "engine_train_steps_graph: engine was created without grads / Adam arenas");
    hipStream_t st = (hipStream_t)stream;
    const bool same = e->graph_exec && e->g_frames == frames && e->g_embeds == embeds && e->g_sched == sched &&
                      e->g_cursor == cursor && e->g_stats == stats_out && e->g_slots == n_slots && e->g_stream == st;
    if (!same) {
        if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
        if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
        if (e->graph_exec_u) { (void)hipGraphExecDestroy(e->graph_exec_u); e->graph_exec_u = nullptr; }
        if (e->graph_u) { (void)hipGraphDestroy(e->graph_u); e->graph_u = nullptr; }
        for (int pass = 0; pass < 2; ++pass) {
            const int reps = pass == 0 ? 1 : ORN_GRAPH_UNROLL;
            hipGraph_t *g = pass == 0 ? &e->graph : &e->graph_u;
            hipGraphExec_t *ge = pass == 0 ? &e->graph_exec : &e->graph_exec_u;
            hipError_t rc = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (rc != hipSuccess) { orn_set_error("graph: BeginCapture failed: %s", hipGetErrorString(rc)); return (int)rc; }
            int trc = 0;
            for (int r = 0; r < reps && trc == 0; ++r) trc = train_step(e, frames, embeds, sched, cursor, stats_out, n_slots, st, r == 0 ? reps : 0, r);
            rc = hipStreamEndCapture(st, g);
            if (trc != 0) { if (*g) { (void)hipGraphDestroy(*g); *g = nullptr; } return trc; }
            if (rc != hipSuccess) { orn_set_error("graph: EndCapture failed: %s", hipGetErrorString(rc)); return (int)rc; }
            rc = hipGraphInstantiate(ge, *g, nullptr, nullptr, 0);
            if (rc != hipSuccess) { orn_set_error("graph: Instantiate failed: %s", hipGetErrorString(rc)); return (int)rc; }
        }
        e->g_frames = frames; e->g_embeds = embeds; e->g_sched = sched; e->g_cursor = cursor; e->g_stats = stats_out;
        e->g_slots = n_slots; e->g_stream = st;
    }
    int left = n_steps;
    // The host needs longer to launch the unrolled graph than the single-step one, and on an idle stream that launch is exposed (the
    // device waits for it): a call that will launch several graphs starts with ONE single step, and the unrolled launches queue up
    // behind it while it runs.  (20-step timed region after a synchronise: see DESIGN 6.)
    static const bool first_single = orn_probe_env("ORN_GRAPH_NO_FIRST_SINGLE") == nullptr;
    if (first_single && left > ORN_GRAPH_UNROLL) {
        hipError_t rc = hipGraphLaunch(e->graph_exec, st);
        if (rc != hipSuccess) { orn_set_error("graph: Launch failed: %s", hipGetErrorString(rc)); return (int)rc; }
        left -= 1;
    }
    while (left > 0) {
        const bool big = left >= ORN_GRAPH_UNROLL;
        hipError_t rc = hipGraphLaunch(big ? e->graph_exec_u : e->graph_exec, st);
        if (rc != hipSuccess) { orn_set_error("graph: Launch failed: %s", hipGetErrorString(rc)); return (int)rc; }
        left -= big ? ORN_GRAPH_UNROLL : 1;
    }
    return 0;
}
