"""Host-side mirror of the reference's utils surface on the hot path (utils.py:110-259)."""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


class PositionalEncoding(nn.Module):
    """utils.py:110-129.  The reference evaluates this on the CPU and copies the 320-byte result to
    the GPU (main_train.py:234-235); here the same fp32 arithmetic runs in `orn_pe_fwd` and the
    result is born on the device."""

    def __init__(self, pe_embed):
        super().__init__()
        self.pe_embed = pe_embed.lower()
        if self.pe_embed == 'none':
            self.embed_length = 1
        else:
            self.lbase, self.levels = [float(x) for x in pe_embed.split('_')]
            self.levels = int(self.levels)
            self.embed_length = 2 * self.levels

    def forward(self, pos):
        if self.pe_embed == 'none':
            return pos[:, None]
        if not pos.is_cuda:
            pos = pos.to('cuda')
        return ops.pe_forward(pos, self.lbase, self.levels)


def loss_fn(pred, target, args):
    """utils.py:139-189 for the loss types the hot path builds (L2, L1, Fusion6)."""
    return ops.LossFn.apply(pred, target.detach(), args.loss_type)


def psnr_fn(output_list, target_list):
    """utils.py:191-199: -10 log10(batch-mean MSE), expanded to [B, stages]."""
    psnr_list = []
    for output, target in zip(output_list, target_list):
        stats, _ = ops.loss_stats(output.detach(), target.detach(), 'L2', want_grad=False)
        psnr_list.append(stats[4].view(1, 1).expand(output.size(0), -1))
    return torch.cat(psnr_list, dim=1)


def msssim_fn(output_list, target_list):
    """utils.py:201-211: MS-SSIM per stage when H >= 160, else 0; expanded to [B, stages]."""
    vals = []
    for output, target in zip(output_list, target_list):
        if output.size(-2) >= 160:
            vals.append(ops.ms_ssim(output.float().detach(), target.detach()).view(1))
        else:
            vals.append(torch.zeros(1, device=output.device))
    ms = torch.cat(vals, dim=0)
    return ms.view(1, -1).expand(output_list[-1].size(0), -1)


def lr_value(cur_epoch, cur_iter, data_size, args):
    """The multiplier arithmetic of utils.py:240-259, in Python doubles."""
    cur_epoch = cur_epoch + (float(cur_iter) / data_size)
    if args.lr_type == 'cosine':
        lr_mult = 0.5 * (math.cos(math.pi * (cur_epoch - args.warmup) / (args.epochs - args.warmup)) + 1.0)
    elif args.lr_type == 'step':
        lr_mult = 0.1 ** (sum(cur_epoch >= np.array(args.lr_steps)))
    elif args.lr_type in ('const', 'plateau'):
        lr_mult = 1
    else:
        raise NotImplementedError
    if cur_epoch < args.warmup:
        lr_mult = 0.1 + 0.9 * cur_epoch / args.warmup
    return args.lr * lr_mult


def adjust_lr(optimizer, cur_epoch, cur_iter, data_size, args):
    """utils.py:240-259."""
    lr = lr_value(cur_epoch, cur_iter, data_size, args)
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr
    return lr


def RoundTensor(x, num=2, group_str=False):
    """utils.py:213-238 (log formatting)."""
    if group_str:
        return '/'.join(','.join(str(round(ele, num)) for ele in x[i].tolist()) for i in range(x.size(0)))
    return ','.join(str(round(ele, num)) for ele in x.flatten().tolist())
