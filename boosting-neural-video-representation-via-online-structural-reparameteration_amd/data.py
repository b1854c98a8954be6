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


class FrameDir:
    """CustomDataSet (model.py:11-70) without the DataLoader: same indexing, quirks included.

    * every directory entry counts (sorted os.listdir, model.py:26-27): N_all of them;
    * normalised time of entry i is float(i) / N_all (model.py:37);
    * `vid_list` (when it holds no None) is a list of frame INDICES that subsets the time table only
      (model.py:40-41) -- the file table stays whole, so item k pairs file k*gap with time frame_idx[vid_list[k*gap]];
    * len = len(time table) // frame_gap (floor, model.py:50); item k reads position k * frame_gap (model.py:60-68);
    * ToTensor semantics (uint8 / 255 -> fp32 CHW); a frame taller than wide is transposed (model.py:66-67)."""

    def __init__(self, main_dir: str, vid_list=(None,), frame_gap: int = 1):
        import os
        self.main_dir = main_dir
        self.frame_path = sorted(os.listdir(main_dir))
        n_all = len(self.frame_path)
        self.frame_idx = [float(x) / n_all for x in range(n_all)]
        if vid_list is not None and None not in vid_list:
            self.frame_idx = [self.frame_idx[i] for i in vid_list]
        self.frame_gap = frame_gap

    def __len__(self):
        return len(self.frame_idx) // self.frame_gap

    def item(self, idx: int):
        """(uint8 CHW tensor, normalised time as a Python float) of sample idx."""
        import os
        import numpy as np
        from PIL import Image
        valid = idx * self.frame_gap
        img = np.asarray(Image.open(os.path.join(self.main_dir, self.frame_path[valid])).convert('RGB'), dtype=np.uint8)
        t = torch.from_numpy(img.copy()).permute(2, 0, 1)
        if t.shape[1] > t.shape[2]:
            t = t.permute(0, 2, 1)
        return t, self.frame_idx[valid]

    def load(self, device='cuda'):
        """All samples at once: frames [n,3,H,W] fp32 in [0,1] resident on `device`, times [n] fp32 (torch.tensor of the
        Python doubles, as model.py:68 makes them)."""
        if len(self) == 0:
            raise FileNotFoundError(f'no frames in {self.main_dir}')
        items = [self.item(k) for k in range(len(self))]
        # ToTensor's arithmetic on the host (uint8 -> fp32, / 255 correctly rounded), then one upload: a device-side division
        # may be compiled to a reciprocal multiply and differ in the last bit
        frames = torch.stack([f.float().div(255.0) for f, _ in items]).contiguous().to(device)
        return frames, torch.tensor([t for _, t in items], dtype=torch.float32)


def load_png_dir(main_dir: str, vid_list=(None,), frame_gap: int = 1, device='cuda'):
    """(frames [n,3,H,W] fp32 on `device`, times [n] fp32) of a frame directory with CustomDataSet's indexing (FrameDir)."""
    return FrameDir(main_dir, vid_list, frame_gap).load(device)
