"""Host-side compression utilities of the evaluation path: per-tensor / per-axis affine
quantisation (utils.py:11-67), global L1 unstructured pruning (main_eval.py:269-273, torch.nn.utils.prune
semantics, which keep a pruned weight as weight_orig + weight_mask in the state dict) and the entropy-coded size estimate
(main_eval.py:652-729, which uses `dahuffman`; an optimal prefix code over the same level histogram is computed here --
parity of the bit count unpinned, the package is not in the reference tree; the histogram itself is pinned by
tests/golden/prune.npz).  One-shot, milliseconds: plain torch on whatever device the tensors are on."""
import heapq
from collections import Counter
from typing import Dict, Iterable, Tuple

import torch


def quantize_per_tensor(t: torch.Tensor, bit: int = 8, axis: int = -1) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils.py:11-67, including its quirks: zeros are ignored when finding min/max, levels are
    round((t - min) / ((max - min) / 2**bit)) in [0, 2**bit], an all-equal slice has scale 0 and collapses to min."""
    if axis == -1:
        t_valid = t != 0
        t_min, t_max = t[t_valid].min(), t[t_valid].max()
        scale = (t_max - t_min) / 2 ** bit
    elif axis in (0, 1):
        mins, maxs = [], []
        for i in range(t.size(axis)):
            sl = t[i] if axis == 0 else t[:, i]
            valid = sl != 0
            if valid.sum():
                mins.append(sl[valid].min())
                maxs.append(sl[valid].max())
            else:
                mins.append(torch.zeros((), dtype=t.dtype, device=t.device))
                maxs.append(torch.zeros((), dtype=t.dtype, device=t.device))
        mn, mx = torch.stack(mins).to(t.device), torch.stack(maxs).to(t.device)
        scale = (mx - mn) / 2 ** bit
        if t.dim() == 4:
            shape = (-1, 1, 1, 1) if axis == 0 else (1, -1, 1, 1)
        elif t.dim() == 2:
            shape = (-1, 1) if axis == 0 else (1, -1)
        else:
            raise UnboundLocalError('quantize_per_tensor: per-axis mode needs a 2-D or 4-D tensor (the reference fails the same way)')
        scale, t_min = scale.view(shape), mn.view(shape)
    else:
        raise ValueError(f'axis {axis}')
    quant_t = ((t - t_min) / (scale + 1e-19)).round()
    new_t = t_min + scale * quant_t
    return quant_t, new_t


def global_l1_prune_masks(tensors: Dict[str, torch.Tensor], amount: float) -> Dict[str, torch.Tensor]:
    """prune.global_unstructured(..., L1Unstructured, amount): zero the `amount` fraction of smallest |w| over all
    tensors together.  Returns 0/1 masks keyed like `tensors` (torch prunes exactly round(amount*N) entries)."""
    flat = torch.cat([v.detach().abs().flatten() for v in tensors.values()])
    n_prune = int(round(amount * flat.numel()))
    masks = {k: torch.ones_like(v) for k, v in tensors.items()}
    if n_prune == 0:
        return masks
    idx = torch.topk(flat, n_prune, largest=False).indices
    fm = torch.ones_like(flat)
    fm[idx] = 0
    off = 0
    for k, v in tensors.items():
        masks[k] = fm[off:off + v.numel()].view_as(v)
        off += v.numel()
    return masks


def huffman_total_bits(symbols: Iterable[int]) -> int:
    """Total code length of an optimal prefix code for the symbol stream (= what dahuffman's encoder emits,
    up to its end-of-stream symbol)."""
    return huffman_bits_from_counts(Counter(symbols).values())


def huffman_bits_from_counts(counts: Iterable[int]) -> int:
    counts = [int(c) for c in counts]
    if len(counts) <= 1:
        return sum(counts)                   # one symbol still costs one bit each
    heap = [(f, i, 0) for i, f in enumerate(counts)]
    heapq.heapify(heap)
    total = 0
    nxt = len(heap)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        total += a[0] + b[0]                  # every merge adds one bit to all symbols below it
        heapq.heappush(heap, (a[0] + b[0], nxt, 0))
        nxt += 1
    return total


def pruned_state_dict(sd: Dict[str, torch.Tensor], masks: Dict[str, torch.Tensor], originals: Dict[str, torch.Tensor] = None):
    """The state dict of a model after torch.nn.utils.prune (never removed by the reference): a pruned `X.weight` appears as
    `X.weight_orig` (the UNMASKED values) + `X.weight_mask`, in that order behind the module's other parameters.
    originals: unmasked values where `sd` already holds weight * mask (e.g. after a masked fine-tune)."""
    mods, by_mod = [], {}
    for k in sd:
        mod = k.rsplit('.', 1)[0]
        if mod not in by_mod:
            by_mod[mod] = []
            mods.append(mod)
        by_mod[mod].append(k)
    out = {}
    for mod in mods:             # torch: the module's remaining parameters, then weight_orig (re-registered last), then the mask buffer
        pruned = [k for k in by_mod[mod] if k in masks]
        for k in by_mod[mod]:
            if k not in masks:
                out[k] = sd[k]
        for k in pruned:
            v = sd[k]
            out[k + '_orig'] = v if originals is None or k not in originals else torch.where(masks[k] != 0, v, originals[k].to(v))
        for k in pruned:
            out[k + '_mask'] = masks[k].to(sd[k])
    return out


def fold_pruned(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """weight = weight_orig * weight_mask for every pruned pair (what torch's forward pre-hook computes)."""
    out = {}
    for k, v in sd.items():
        if k.endswith('_mask'):
            continue
        if k.endswith('_orig'):
            out[k[:-5]] = v * sd[k[:-5] + '_mask']
        else:
            out[k] = v
    return out


def quantized_model_bits(state_dict: Dict[str, torch.Tensor], bit: int = 8, axis: int = 0):
    """main_eval.py:652-691 on a state dict as `model.state_dict()` gives it there (pruned tensors as weight_orig +
    weight_mask, see pruned_state_dict): every tensor is quantised -- per `axis` when it is 2-D / 4-D and not a bias
    (main_eval.py:661), per tensor otherwise -- and ONE Huffman table codes the levels of the entries that are non-zero in the
    tensor being quantised (main_eval.py:664-667: `mask_cpu = v != 0; quant_v_cpu[mask_cpu]`).  Q3 comes with it: a 0/1
    weight_mask quantises to all ones (its non-zero entries have min = max = 1, scale 0), so the pruning is undone in the
    de-quantised model, and each of its ones adds a level-0 symbol to the stream.
    Returns (de-quantised state dict, total bits, number of coded values, {level: count}).
    The total is that of an optimal prefix code over the level histogram; the reference takes its code lengths from
    `dahuffman` (absent from the reference tree and from this image: parity of the bit count is unpinned -- dahuffman adds an
    end-of-stream symbol of frequency 1 to the table, which can lengthen other codes by a bit)."""
    new_sd, hist = {}, Counter()
    count = 0
    for k, v in state_dict.items():
        large = v.dim() in (2, 4) and 'bias' not in k
        q, nv = quantize_per_tensor(v.float(), bit, axis if large else -1)
        new_sd[k] = nv.to(v.dtype)
        valid = q[v != 0].flatten()
        count += valid.numel()
        vals, cnts = torch.unique(valid.to(torch.float64), return_counts=True)
        for a_, c_ in zip(vals.tolist(), cnts.tolist()):
            hist[a_] += c_
    return new_sd, huffman_bits_from_counts(hist.values()), count, dict(hist)
