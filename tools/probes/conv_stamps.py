"""Phase stamps of the 16-bit conv kernel (needs the diagnostic build:
   ORN_BUILD_TAG=stamp ORN_EXTRA_DEFS=-DORN_CONV_STAMP python -m orn_amd._build ; run with ORN_LIB_PATH=.../liborn_stamp.so).
Per work-group (wave 0): s_memtime at N-tile start / after the prologue rendezvous / end of the main loop / end of the epilogue,
and s_memrealtime (100 MHz) around the whole work-group -> in-kernel clock.  usage: conv_stamps.py <fwd|dgrad> [layer]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = {4: (360, 640), 3: (180, 320), 2: (90, 160)}[layer]
C, O, s = 96, 384, 2
dev = 'cuda'
torch.manual_seed(0)
bf = torch.bfloat16
xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=bf); xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(bf)
wb = (torch.randn(9, O, C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(bf)
wd = (torch.randn(9, C, O, device=dev) * (1.0 / (9 * O) ** 0.5)).to(bf)
bp = torch.randn(O, device=dev)
z = torch.empty(H * s, W * s, O // 4, device=dev, dtype=bf)
dypad = torch.zeros(H + 2, W + 2, O, device=dev, dtype=bf); dypad[1:-1, 1:-1] = torch.randn(H, W, O, device=dev).to(bf)
zprev = torch.randn(H, W, C, device=dev).to(bf)
dyprev = torch.zeros(H // 2 + 2, W // 2 + 2, C * 4, device=dev, dtype=bf)
st = _lib.stream()
P = lambda t: c_void_p(t.data_ptr())
nwg = 4096
stamps = torch.zeros(nwg, 128, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(_lib.lib_path())
raw.orn_debug_set_stamps.argtypes = [c_void_p]
def run():
    if which == 'fwd':
        _lib.check(lib.orn_conv_nhwc_bf16_fwd(P(xpad), P(wb), _lib.ptr(bp), H, W, C, O, s, P(z), None, st))
    else:
        _lib.check(lib.orn_dgrad_nhwc_bf16(P(dypad), P(wd), H, W, O, C, P(zprev), P(dyprev), 2, st))
for _ in range(50):          # warm clocks
    run()
raw.orn_debug_set_stamps(P(stamps))
if len(sys.argv) > 3:
    lib.orn_debug_set(int(sys.argv[3]))          # ablation flags (needs -DORN_CONV_ABLATE too): 4 = stores dropped
run()
torch.cuda.synchronize()
raw.orn_debug_set_stamps(None)
S = stamps.cpu()
used = S[:, 0] != 0
S = S[used]
print(f'{which} L{layer}: {S.shape[0]} work-groups stamped')
rt = (S[:, 1] - S[:, 0]).double()                    # 100 MHz ticks
ntile = 3 if which == 'fwd' else 1
full = S[:, 2 + 4 * ntile] != 0 if which == 'fwd' else torch.ones(S.shape[0], dtype=torch.bool)
F = S[full]
life = (F[:, 2 + ntile * 4] - F[:, 2]).double()
rtF = (F[:, 1] - F[:, 0]).double()
clk = (life / rtF * 100.0)
print(f'  work-groups with {ntile} N tile(s): {F.shape[0]}; lifetime median {life.median():.0f} cycles, in-kernel clock median {clk.median():.0f} MHz')
for n in range(ntile):
    a, b, c, d = (F[:, 2 + 4 * n + k].double() for k in range(4))
    nxt = F[:, 2 + 4 * (n + 1)].double()
    print(f'  N tile {n}: prologue {(b - a).median():.0f}  main loop {(c - b).median():.0f}  epilogue {(d - c).median():.0f}  to next {(nxt - d).median():.0f}   (cycles, median)')
if which == 'fwd':
    E = S[~full]
    if E.shape[0]:
        a, b, c, d = (E[:, 2 + k].double() for k in range(4))
        print(f'  single-N-tile work-groups: {E.shape[0]}; prologue {(b - a).median():.0f} main {(c - b).median():.0f} epilogue {(d - c).median():.0f}')
# per-tap stamps (taken when the tap's last MFMA has been issued, before the wait + rendezvous): wave 0 and its SIMD partner
nseg = 3 if which == 'fwd' else 4
for base, nm in ((16, 'wave 0 (loader)'), (64, 'SIMD partner  ')):
    T = F[:, base:base + nseg * 9].double()
    if (T == 0).any():
        T = T[(T != 0).all(dim=1)]
    if T.shape[0] == 0:
        continue
    d = (T[:, 1:] - T[:, :-1]).median(dim=0).values
    first = (T[:, 0] - F[:T.shape[0], 3].double()).median() if T.shape[0] == F.shape[0] else float('nan')
    print(f'  {nm} tap-to-tap cycles (median): first tap {first:.0f} | ' + ' '.join(f'{x:.0f}' for x in d.tolist()))
for base, nm in ((100, 'wave 0'), (112, 'partner')):
    B = F[:, base:base + 9].double()
    B = B[(B != 0).all(dim=1)]
    if B.shape[0]:
        for k in range(3):
            a, w, x = B[:, 3 * k], B[:, 3 * k + 1], B[:, 3 * k + 2]
            print(f'  {nm} rendezvous of tap {3 + k}: vmcnt wait {(w - a).median():.0f}  barrier {(x - w).median():.0f} cycles (median); arrival minus tap stamp {(a - F[:B.shape[0], (16 if base == 100 else 64) + 3 + k].double()).median():.0f}')
span = (S[:, 1].max() - S[:, 0].min()).item() / 100.0
print(f'  first start -> last end: {span:.1f} us (realtime)')
