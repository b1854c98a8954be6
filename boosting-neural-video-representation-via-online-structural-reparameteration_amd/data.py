"""Synthetic video resident in HBM (there is no dataset access): the frame-ingest side of the hot
path (CustomDataSet, model.py:11-70) is replaced by frames generated on the device."""
import math

import torch


def synthetic_video(frames: int, h: int, w: int, seed: int = 1234, device='cuda', noise: float = 0.1) -> torch.Tensor:
    """[frames,3,h,w] fp32 in [0,1]: V[k] = clip(0.5 + sum_4 (0.25/4) sin(2pi(fx x + fy y + k/frames + ph))
    + noise*U(-1,1)) with 4 random (fx, fy, ph) per channel (SURVEY 8d recipe).  Deterministic per seed."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    ys = torch.linspace(0, 1, h, device=device).view(1, 1, h, 1)
    xs = torch.linspace(0, 1, w, device=device).view(1, 1, 1, w)
    ks = (torch.arange(frames, dtype=torch.float32, device=device) / frames).view(frames, 1, 1, 1)
    v = torch.full((frames, 3, h, w), 0.5, device=device)
    for _ in range(4):
        fx = (torch.rand(3, generator=g) * 6 + 0.5).view(1, 3, 1, 1).to(device)
        fy = (torch.rand(3, generator=g) * 6 + 0.5).view(1, 3, 1, 1).to(device)
        ph = torch.rand(3, generator=g).view(1, 3, 1, 1).to(device)
        v += (0.25 / 4) * torch.sin(2 * math.pi * (fx * xs + fy * ys + ks + ph))
    gd = torch.Generator(device=device).manual_seed(seed)
    noise_amp = noise
    noise = torch.rand((frames, 3, h, w), generator=gd, device=device)
    v += noise_amp * (noise * 2 - 1)
    return v.clamp_(0, 1).contiguous()


def load_png_dir(main_dir: str, vid_list=(None,), frame_gap: int = 1, device='cuda') -> torch.Tensor:
    """CustomDataSet (model.py:11-70) without DataLoader workers: every `frame_gap`-th PNG of `main_dir`
    (optionally filtered by video id prefix), ToTensor semantics (uint8/255 -> fp32 CHW), portrait frames
    transposed to landscape (model.py:66-67), uploaded once and kept resident in HBM."""
    import os
    import numpy as np
    from PIL import Image
    names = sorted(f for f in os.listdir(main_dir) if f.lower().endswith(('.png', '.jpg', '.jpeg')))
    if vid_list and vid_list[0] is not None:
        names = [f for f in names if any(f.startswith(f'{v}_') or f.startswith(f'{v:03d}') for v in vid_list)]
    names = names[::frame_gap]
    if not names:
        raise FileNotFoundError(f'no frames in {main_dir}')
    frames = []
    for f in names:
        img = np.asarray(Image.open(os.path.join(main_dir, f)).convert('RGB'), dtype=np.uint8)
        t = torch.from_numpy(img).permute(2, 0, 1)
        if t.shape[1] > t.shape[2]:
            t = t.permute(0, 2, 1)
        frames.append(t)
    return (torch.stack(frames).to(device).float() / 255.0).contiguous()
