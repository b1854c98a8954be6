"""Checkpoint interchange with the reference (main_train.py:292-358, main_eval.py:216-237).

File = torch.save of {'epoch', 'state_dict', 'train_best_psnr', 'train_best_msssim', 'val_best_psnr',
'val_best_msssim', 'optimizer'}; `state_dict` uses the reference's key layout (stem.{0,2}.*, layers.N.<branch>.*,
head_layers.K.*; deploy files carry layers.N.rbr_reparam.{weight,bias}).  Loading is always
`weights_only=True` (nothing from the file is executed)."""
import os
from typing import Dict, Optional

import torch

TRAIN_ERB_SUFFIXES = ('rbr_3x3_branch.weight', 'rbr_3x3_branch.bias', 'rbr_3x1_branch.weight', 'rbr_3x1_branch.bias',
                      'rbr_1x3_branch.weight', 'rbr_1x3_branch.bias', 'rbr_1x1_3x3_1x1_branch_1x1_1.weight',
                      'rbr_1x1_3x3_1x1_branch_3x3.weight', 'rbr_1x1_3x3_1x1_branch_1x1_2.weight')


def state_dict_kind(sd: Dict[str, torch.Tensor]) -> str:
    """'deploy' | 'ERB' | 'NeRV_vanilla' from the key layout (read_pth.py documents the same three)."""
    keys = list(sd.keys())
    if any(k.endswith('rbr_reparam.weight') for k in keys):
        return 'deploy'
    if any(k.endswith('rbr_3x3_branch.weight') for k in keys):
        return 'ERB'
    if any(k.endswith('.branch.weight') for k in keys):
        return 'NeRV_vanilla'
    raise ValueError('unrecognised NeRV state dict layout')


def strip_profiler_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """thop leaves total_ops / total_params buffers in reference checkpoints (main_train.py:414-417)."""
    return {k: v for k, v in sd.items() if not (k.endswith('total_ops') or k.endswith('total_params'))}


def deploy_state_dict(model) -> Dict[str, torch.Tensor]:
    """The reference's `*_deploy.pth` state dict (main_train.py:332-346): every block merged into rbr_reparam."""
    out = {}
    done = set()
    for k, v in model.state_dict().items():            # keep the reference's key order: stem, layers.N, head
        if not k.startswith('layers.'):
            out[k] = v.detach().cpu().clone()
            continue
        i = int(k.split('.')[1])
        if i in done:
            continue
        done.add(i)
        blk = model.layers[i]
        if getattr(blk, 'deploy', False):
            w, b = blk.rbr_reparam.weight, blk.rbr_reparam.bias
        elif blk.branch_type == 'ERB':
            with torch.no_grad():
                w, b = blk.get_equivalent_kernel_bias()
        else:
            w, b = blk.branch.weight, blk.branch.bias
        out[f'layers.{i}.rbr_reparam.weight'] = w.detach().cpu().clone()
        out[f'layers.{i}.rbr_reparam.bias'] = b.detach().cpu().clone()
    return out


def deploy_param_count(model) -> int:
    """Parameters of the deploy-state model (main_train.py:362-364, "Deploy Rep-Model Params")."""
    return sum(v.numel() for v in deploy_state_dict(model).values())


def save(path: str, model, epoch: int, optimizer_state=None, train_best_psnr=None, val_best_psnr=None, deploy: bool = False,
         train_best_msssim=None, val_best_msssim=None):
    """One checkpoint file in the reference's layout (main_train.py:293-301); the four best-so-far entries are 0-d tensors
    as the reference leaves them."""
    sd = deploy_state_dict(model) if deploy else {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    zero = torch.tensor(0)

    def t(x):
        return zero if x is None else torch.as_tensor(x).detach().cpu().reshape(())
    ck = {'epoch': epoch, 'state_dict': sd,
          'train_best_psnr': t(train_best_psnr), 'train_best_msssim': t(train_best_msssim),
          'val_best_psnr': t(val_best_psnr), 'val_best_msssim': t(val_best_msssim),
          'optimizer': optimizer_state if optimizer_state is not None else {}}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)
    return ck


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """Reads a reference (or our) checkpoint safely; accepts a bare state dict too."""
    ck = torch.load(path, map_location='cpu', weights_only=True)
    sd = ck['state_dict'] if isinstance(ck, dict) and 'state_dict' in ck else ck
    return strip_profiler_keys(sd)


def load_into(model, sd: Dict[str, torch.Tensor], strict: bool = True):
    """load_state_dict with the reference's tolerance for profiler keys; switches ERB blocks to deploy when the
    file is a deploy checkpoint (main_eval.py:534-545 does the reverse order: load train file, then switch)."""
    sd = strip_profiler_keys(sd)
    kind = state_dict_kind(sd)
    if kind == 'deploy':
        for blk in model.layers:
            if not getattr(blk, 'deploy', False):
                blk.switch_to_deploy_structure()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise KeyError(f'state dict mismatch: missing {missing}, unexpected {unexpected}')
    return kind
