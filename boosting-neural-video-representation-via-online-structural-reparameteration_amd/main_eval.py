"""`python -m orn_amd.main_eval <flags>`: the evaluation path of the reference's main_eval.py:
load `model_latest(.pth|_deploy.pth)` -> global L1 prune (--prune_ratio) -> switch to deploy -> per-axis 8-bit
quantisation + Huffman size estimate (--quant_bit) -> decode every frame (PSNR, decoder FPS, bits per pixel).

`--finetune` (main_eval.py:213-545): load the TRAIN-mode checkpoint, prune the stem Linear weights and every conv
branch weight together (global L1), fine-tune `--finetune_epochs` on the native engine, then deploy/quantise/evaluate.
The reference's behaviour is kept, quirks included: Q1 -- with ERB the online merge reads `.weight`
directly, so torch's pruning hooks never fire and the branch conv weights stay frozen at their pruned-at-t0 values
(only stem weights under their mask, all biases and the head train); NeRV_vanilla convs train under their masks.
Q2 -- the LR comes from adjust_lr(epoch % total_epochs) with epoch continuing from the checkpoint (>= args.epochs),
i.e. the cosine evaluated past its end (0 at e = 300, 4e-5 x lr at 301, ...).  Q3 is `quantize_per_tensor`'s."""
import os
import time

import torch

from . import checkpoint, data as odata, eval_utils, model as omodel, ops, utils
from .main_train import build_parser


ERB_BRANCHES = ('rbr_3x3_branch', 'rbr_3x1_branch', 'rbr_1x3_branch', 'rbr_1x1_3x3_1x1_branch_1x1_1',
                'rbr_1x1_3x3_1x1_branch_3x3', 'rbr_1x1_3x3_1x1_branch_1x1_2')


def _prune_finetune(model, args, PE, ckpt_path, vid_index=0):
    """main_eval.py:213-531 on the native engine (see the module docstring for the reference quirks kept)."""
    from . import engine as oeng
    named = dict(model.named_parameters())
    prunable = {}
    for k in named:                                            # main_eval.py:296-302: stem Linear weights
        if k.startswith('stem.') and k.endswith('weight'):
            prunable[k] = named[k].detach()
    for i, blk in enumerate(model.layers):                     # main_eval.py:305-340 (ERB) / 244-264 (vanilla)
        names = ERB_BRANCHES if blk.branch_type == 'ERB' else ('branch',)
        for b in names:
            k = f'layers.{i}.{b}.weight'
            if k in named:
                prunable[k] = named[k].detach()
    originals = {k: v.clone() for k, v in prunable.items() if k.startswith('stem.')}     # weight_orig keeps them under the mask
    masks = eval_utils.global_l1_prune_masks(prunable, args.prune_ratio)
    zero = sum(int((m == 0).sum()) for m in masks.values())
    tot = sum(m.numel() for m in masks.values())
    print(f'global L1 prune of {len(masks)} tensors: {zero}/{tot} = {zero / tot:.3f} (asked {args.prune_ratio})')
    with torch.no_grad():
        for k, m in masks.items():
            named[k].mul_(m)
    eng = oeng.TrainEngine(model, loss_type=args.loss_type, beta=args.beta, precision=args.precision)
    gmask = {}
    for k, m in masks.items():                                 # Q1: ERB branch convs are frozen, the rest trains masked
        frozen = k.startswith('layers.') and args.branch_type == 'ERB'
        gmask[k] = torch.zeros_like(m) if frozen else m
    eng.set_grad_mask(gmask)
    hw = eng.out_hw
    from .main_train import load_frames
    frames, pos = load_frames(args, hw, eng.device, args.dataset, vid_index, args.frame_gap)
    n = frames.shape[0]
    eng.set_video(frames, PE(pos))
    try:
        start_epoch = int(torch.load(ckpt_path, map_location='cpu', weights_only=True).get('epoch', args.epochs))
    except Exception:
        start_epoch = args.epochs
    total = start_epoch + args.finetune_epochs
    g = torch.Generator()
    step = 0
    for epoch in range(start_epoch, total):
        g.manual_seed(args.manualSeed + epoch)
        order = torch.randperm(n, generator=g).tolist()
        entries = []
        for i, f in enumerate(order):
            step += 1                                          # fresh optimiser: main_eval.py:496 clears its state
            entries.append((f, step, utils.lr_value(epoch % total, i, n, args)))      # Q2
        eng.set_schedule(entries)
        eng.run(n)
        st = eng.stats(n)
        if (epoch - start_epoch) % max(1, args.finetune_epochs // 10) == 0 or epoch == total - 1:
            print(f'fine-tune epoch {epoch + 1}/{total} lr {float(st[-1, 5]):.2e} PSNR {float(st[:, 4].mean()):.2f}', flush=True)
    eng.set_grad_mask(None)
    torch.cuda.synchronize()
    # what stays pruned in the reference's state dict after the fine-tune: the stem Linear layers (weight_orig + weight_mask;
    # the ERB branch modules disappear in switch_to_deploy, a vanilla block keeps its pruned `branch`)
    keep = {k: m for k, m in masks.items() if k.startswith('stem.') or args.branch_type != 'ERB'}
    return keep, {k: v for k, v in originals.items() if k in keep}


def main(argv=None):
    p = build_parser()
    p.add_argument('--finetune', action='store_true')
    p.add_argument('--finetune_epochs', type=int, default=100)
    p.add_argument('--cycles', type=int, default=1)
    p.add_argument('--video', default=None, help='which video of a multi-video training job (--dataset a,b,c or a multi-rank '
                                                  '--synthetic job: synthetic<k>) to evaluate; default: the first')
    args = p.parse_args(argv)
    args.warmup = int(args.warmup * args.epochs)          # main_eval.py:106 (as main_train.parse_args): warm-up in epochs
    outf = os.path.join('result', args.outf, f'{args.suffix}')
    # a job that fitted several videos wrote one sub-directory per video (main_train.video_outf)
    videos = [d for d in args.dataset.split(',') if d]
    vid_index = 0
    if args.synthetic and args.video:
        vid_index = int(args.video[len('synthetic'):])
        if os.path.isdir(os.path.join(outf, args.video)):
            outf = os.path.join(outf, args.video)
    elif len(videos) > 1:
        name = args.video or videos[0]
        if name not in videos:
            raise ValueError(f'--video {name} is not one of --dataset {videos}')
        vid_index = videos.index(name)
        args.dataset = name
        outf = os.path.join(outf, name)
    PE = utils.PositionalEncoding(args.embed)
    deploy_file = os.path.join(outf, 'model_latest_deploy.pth')
    train_file = os.path.join(outf, 'model_latest.pth')
    finetune = args.finetune and args.prune_ratio < 1
    path = deploy_file if (args.branch_type == 'ERB' and os.path.exists(deploy_file) and not finetune) else train_file
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    sd = checkpoint.load_state_dict_file(path)
    torch.manual_seed(args.manualSeed)
    model = omodel.Generator(embed_length=PE.embed_length, stem_dim_num=args.stem_dim_num, fc_hw_dim=args.fc_hw_dim,
                             expansion=args.expansion, num_blocks=args.num_blocks, norm=args.norm, act=args.act, bias=True,
                             reduction=args.reduction, conv_type=args.conv_type, stride_list=args.strides,
                             sin_res=args.single_res, lower_width=args.lower_width, sigmoid=args.sigmoid, deploy=False,
                             branch_type=args.branch_type)
    kind = checkpoint.load_into(model, sd)
    model = model.cuda()
    masks, originals = {}, None
    if finetune:
        masks, originals = _prune_finetune(model, args, PE, path, vid_index) if vid_index else _prune_finetune(model, args, PE, path)
    if kind != 'deploy':
        for blk in model.layers:
            blk.switch_to_deploy() if blk.branch_type == 'ERB' else None
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    n_param = sum(v.numel() for v in sd.values())
    if args.prune_ratio < 1 and not finetune:
        # main_eval.py:551-650: global L1 over the stem Linear weights and the blocks' single convs (deploy-state
        # rbr_reparam for ERB, branch / rbr_reparam for NeRV_vanilla) -- not the head, not the biases
        prunable = {k: v for k, v in sd.items()
                    if k.endswith('.weight') and (k.startswith('stem.') or (k.startswith('layers.') and k.split('.')[2] in ('rbr_reparam', 'branch')))}
        masks = eval_utils.global_l1_prune_masks(prunable, args.prune_ratio)
        kept = sum(int(m.sum()) for m in masks.values())
        print(f'global L1 prune of {len(masks)} tensors: {1 - kept / sum(m.numel() for m in masks.values()):.3f} of the weights masked')
    # the state dict as the reference sees it from here on: pruned tensors as weight_orig + weight_mask (never removed)
    sd = eval_utils.pruned_state_dict(sd, masks, originals)
    bits = None
    if args.quant_bit != -1:
        sd, bits, count, _ = eval_utils.quantized_model_bits(sd, args.quant_bit, args.quant_axis)
        print(f'quantised to {args.quant_bit} bit: {bits / 8 / 1e6:.3f} MB entropy-coded over the {count} non-zero entries '
              f'({bits / max(count, 1):.2f} bits each; optimal prefix code, dahuffman parity unpinned)')
    model.load_state_dict(eval_utils.fold_pruned(sd))
    hw = [model.fc_h, model.fc_w]
    for s_ in args.strides:
        hw = [hw[0] * s_, hw[1] * s_]
    from .main_train import load_frames
    frames, pos = load_frames(args, hw, 'cuda', args.dataset, vid_index, args.test_gap)       # val_dataset: CustomDataSet(frame_gap=test_gap)
    n = frames.shape[0]
    embeds = PE(pos)
    psnrs = []
    torch.cuda.synchronize()
    t0 = time.time()
    dumped = []
    with torch.no_grad():
        for k in range(n):
            img = model(embeds[k:k + 1])[0]
            st, _ = ops.loss_stats(img, frames[k:k + 1], 'L2', want_grad=False)
            psnrs.append(st[4])
            if args.dump_images:                      # main_eval.py:795-803 (written after the timed loop)
                dumped.append(img[0].detach())
    torch.cuda.synchronize()
    fps = n / (time.time() - t0)
    psnr = float(torch.stack(psnrs).mean())
    if args.dump_images:
        from PIL import Image
        visual_dir = os.path.join(outf, 'visualize')
        os.makedirs(visual_dir, exist_ok=True)
        print(f'Saving predictions to {visual_dir}')
        for k, im in enumerate(dumped):              # torchvision.utils.save_image: x*255 + 0.5, clamp, uint8, HWC
            arr = im.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
            Image.fromarray(arr).save(os.path.join(visual_dir, f'pred_{k}.png'))
    with torch.no_grad():                                            # untimed, as the FPS above is the decoder's
        ms = [utils.msssim_fn([model(embeds[k:k + 1])[0]], [frames[k:k + 1]])[0, 0] for k in range(n)]
    msg = f'Eval: PSNR {psnr:.2f} dB, MS-SSIM {float(torch.stack(ms).mean()):.4f}, decode {fps:.1f} FPS, params {n_param / 1e6:.3f} M'
    if bits is not None:
        msg += f', bpp {bits / (n * hw[0] * hw[1]):.4f}'
    print(msg)
    return psnr


if __name__ == '__main__':
    main()
