"""Synthetic video resident in HBM (there is no dataset access): the frame-ingest side of the hot
path (CustomDataSet, model.py:11-70) is replaced by frames generated on the device."""
import math

import torch


TEXTURE_CUTOFF = 20.0      # default band limit of texture_video (tools/psnr_parity.py sweeps it)


def texture_video(frames: int, h: int, w: int, seed: int = 1234, device='cuda', cutoff: float = None) -> torch.Tensor:
    """[frames,3,h,w] fp32 in [0,1], no white noise: three low-pass-filtered random fields (Gaussian spectrum, `cutoff` cycles per
    image height at 1 sigma) that drift across the frame at different integer velocities, under a slow brightness wave.
    Content a NeRV fits to 30-45 dB -- well above any noise floor and far below the resolution where the last bits of the
    arithmetic decide -- which is what a precision-parity run needs.  Deterministic per seed."""
    cutoff = TEXTURE_CUTOFF if cutoff is None else cutoff
    g = torch.Generator(device='cpu').manual_seed(seed)
    H2, W2 = h + 2 * frames + 8, w + 2 * frames + 8                 # room for the drift
    fy = torch.fft.fftfreq(H2).view(H2, 1) * H2 * (h / H2)          # cycles per image height
    fx = torch.fft.rfftfreq(W2).view(1, W2 // 2 + 1) * W2 * (h / W2)
    env = torch.exp(-(fy ** 2 + fx ** 2) / (2 * cutoff ** 2))
    out = torch.empty(frames, 3, h, w, device=device)
    vel = [(1, 2), (2, -1), (-1, 1)]
    for c in range(3):
        spec = torch.complex(torch.randn(H2, W2 // 2 + 1, generator=g), torch.randn(H2, W2 // 2 + 1, generator=g)) * env
        tex = torch.fft.irfft2(spec, s=(H2, W2))
        tex = ((tex - tex.mean()) / tex.std()).to(device)
        vy, vx = vel[c]
        for k in range(frames):
            oy, ox = frames + 4 + vy * k // 2, frames + 4 + vx * k // 2
            out[k, c] = tex[oy:oy + h, ox:ox + w]
    ks = (torch.arange(frames, dtype=torch.float32, device=device) / max(frames, 1)).view(frames, 1, 1, 1)
    ph = torch.rand(3, generator=g).view(1, 3, 1, 1).to(device)
    out = 0.5 + 0.17 * out + 0.04 * torch.sin(2 * math.pi * (ks + ph))
    return out.clamp_(0, 1).contiguous()


def synthetic_video(frames: int, h: int, w: int, seed: int = 1234, device='cuda', noise: float = 0.1, kind: str = 'waves') -> torch.Tensor:
    """[frames,3,h,w] fp32 in [0,1]: V[k] = clip(0.5 + sum_4 (0.25/4) sin(2pi(fx x + fy y + k/frames + ph))
    + noise*U(-1,1)) with 4 random (fx, fy, ph) per channel (SURVEY 8d recipe).  Deterministic per seed.
    kind='texture': texture_video (drifting band-limited textures, no white noise)."""
    if kind == 'texture':
        return texture_video(frames, h, w, seed=seed, device=device)
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
