"""`python -m orn_amd.main_eval <flags>`: the evaluation path of the reference's main_eval.py:
load `model_latest(.pth|_deploy.pth)` -> global L1 prune (--prune_ratio) -> switch to deploy -> per-axis 8-bit
quantisation + Huffman size estimate (--quant_bit) -> decode every frame (PSNR, decoder FPS, bits per pixel).

Scope note: the reference's optional prune *fine-tune* loop (main_eval.py:450-531) is not reproduced -- with ERB
it trains no conv weights at a near-zero LR; `--finetune` raises NotImplementedError."""
import os
import time

import torch

from . import checkpoint, data as odata, eval_utils, model as omodel, ops, utils
from .main_train import build_parser


def main(argv=None):
    p = build_parser()
    p.add_argument('--finetune', action='store_true')
    p.add_argument('--finetune_epochs', type=int, default=100)
    p.add_argument('--cycles', type=int, default=1)
    args = p.parse_args(argv)
    if args.finetune:
        raise NotImplementedError('prune fine-tuning (main_eval.py:450-531) is outside the built path; see module docstring')
    outf = os.path.join('result', args.outf, f'{args.suffix}')
    PE = utils.PositionalEncoding(args.embed)
    deploy_file = os.path.join(outf, 'model_latest_deploy.pth')
    train_file = os.path.join(outf, 'model_latest.pth')
    path = deploy_file if (args.branch_type == 'ERB' and os.path.exists(deploy_file)) else train_file
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
    if kind != 'deploy':
        for blk in model.layers:
            blk.switch_to_deploy() if blk.branch_type == 'ERB' else None
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    n_param = sum(v.numel() for v in sd.values())
    if args.prune_ratio < 1:                       # main_eval.py:269-273: weights of stem / conv layers, global L1
        prunable = {k: v for k, v in sd.items() if k.endswith('weight')}
        masks = eval_utils.global_l1_prune_masks(prunable, args.prune_ratio)
        for k, m in masks.items():
            sd[k] = sd[k] * m
        kept = sum(int(m.sum()) for m in masks.values())
        print(f'pruned {1 - kept / sum(m.numel() for m in masks.values()):.3f} of the weights')
    bits = None
    if args.quant_bit != -1:
        sd, bits, count = eval_utils.quantized_model_bits(sd, args.quant_bit, args.quant_axis)
        print(f'quantised to {args.quant_bit} bit: {bits / 8 / 1e6:.3f} MB entropy-coded ({bits / count:.2f} bits/param)')
    model.load_state_dict(sd)
    hw = [model.fc_h, model.fc_w]
    for s_ in args.strides:
        hw = [hw[0] * s_, hw[1] * s_]
    frames = (odata.synthetic_video(args.synthetic, hw[0], hw[1], seed=1234) if args.synthetic
              else odata.load_png_dir(f'../data/{args.dataset.lower()}', args.vid, args.test_gap))
    n = frames.shape[0]
    embeds = PE(torch.tensor([float(k) / n for k in range(n)], dtype=torch.float32))
    psnrs = []
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.no_grad():
        for k in range(n):
            img = model(embeds[k:k + 1])[0]
            st, _ = ops.loss_stats(img, frames[k:k + 1], 'L2', want_grad=False)
            psnrs.append(st[4])
    torch.cuda.synchronize()
    fps = n / (time.time() - t0)
    psnr = float(torch.stack(psnrs).mean())
    with torch.no_grad():                                            # untimed, as the FPS above is the decoder's
        ms = [utils.msssim_fn([model(embeds[k:k + 1])[0]], [frames[k:k + 1]])[0, 0] for k in range(n)]
    msg = f'Eval: PSNR {psnr:.2f} dB, MS-SSIM {float(torch.stack(ms).mean()):.4f}, decode {fps:.1f} FPS, params {n_param / 1e6:.3f} M'
    if bits is not None:
        msg += f', bpp {bits / (n * hw[0] * hw[1]):.4f}'
    print(msg)
    return psnr


if __name__ == '__main__':
    main()
