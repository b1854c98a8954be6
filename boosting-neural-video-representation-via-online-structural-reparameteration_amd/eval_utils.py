"""Host-side compression utilities of the evaluation path: per-tensor / per-axis affine
quantisation (utils.py:11-67), global L1 unstructured pruning (main_eval.py:269-273, torch.nn.utils.prune
semantics) and the entropy-coded size estimate (main_eval.py:652-729, which uses `dahuffman`; a canonical Huffman
code has the same total length, so the bit count is reproduced without the package -- parity unpinned, the
package is not in the reference tree).  One-shot, milliseconds: plain torch on whatever device the tensors are on."""
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
    freq = Counter(symbols)
    if len(freq) <= 1:
        return sum(freq.values())            # one symbol still costs one bit each
    heap = [(f, i, 0) for i, f in enumerate(freq.values())]
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


def quantized_model_bits(state_dict: Dict[str, torch.Tensor], bit: int = 8, axis: int = 0):
    """main_eval.py:652-729: quantise every tensor, entropy-code all levels with one Huffman table.
    Returns (de-quantised state dict, total bits, number of coded values)."""
    new_sd, symbols = {}, []
    for k, v in state_dict.items():
        ax = axis if (v.dim() in (2, 4)) else -1
        q, nv = quantize_per_tensor(v.float(), bit, ax)
        new_sd[k] = nv
        symbols += q.flatten().to(torch.int64).tolist()
    return new_sd, huffman_total_bits(symbols), len(symbols)
