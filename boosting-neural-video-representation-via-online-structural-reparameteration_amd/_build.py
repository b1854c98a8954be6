"""Builds csrc/*.hip into liborn.so (gfx950 only) with hipcc; in-tree so the .so travels to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# ORN_BUILD_TAG=<tag> (+ ORN_EXTRA_DEFS="-DX -DY"): a diagnostic variant next to the product library, build_<tag>/ ->
# liborn_<tag>.so, picked up by tools/probes through ORN_LIB_PATH.  The product build uses neither.
TAG = os.environ.get('ORN_BUILD_TAG', '')
OBJ = os.path.join(HERE, 'build' + ('_' + TAG if TAG else ''))
LIB = os.path.join(HERE, 'liborn' + ('_' + TAG if TAG else '') + '.so')
EXTRA = os.environ.get('ORN_EXTRA_DEFS', '').split() if TAG else []
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent fp32 FMAs / multiplies into v_pk_fma_f32 / v_pk_mul_f32 (+ v_mov's to pair the
# operands), which issue slower than the scalar-per-lane instructions they replace on gfx950 (MI355X_MICROARCH.md: 'an anti-lever');
# measured per kernel in round 3: Fusion6 88 -> 71 us, head backward 85 -> see DESIGN 4.3
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function', '-fvisibility=hidden', '-fno-slp-vectorize']
# Per-file flags (none at present)
FILE_FLAGS = {}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'orn.h'))
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'orn_debug.h'))
    hdrs.append(os.path.join(CSRC, 'orn.map'))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hm = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        variants = [('', [])]
        if src in ('orn_conv_bf16.hip', 'orn_conv_fwd_bf16.hip', 'orn_conv2_bf16.hip'):   # the 16-bit fast path is built for bf16 and for IEEE half
            variants.append(('_f16', ['-DORN_FP16']))
        for suffix, extra in variants:
            o = os.path.join(OBJ, src[:-4] + suffix + '.o')
            objs.append(o)
            if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm):
                jobs.append((s, o, extra))

    def cc(job):
        s, o, extra = job
        # ORN_CONV_ABLATE=1 (with force=True): timing-ablation flags of the conv kernels for tools/probes/*_ablate.py
        abl = ['-DORN_CONV_ABLATE'] if os.environ.get('ORN_CONV_ABLATE') == '1' else []
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + extra + abl + EXTRA + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {s}:\n{r.stdout}\n{r.stderr}')
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(cc, jobs):
                if warn and verbose:
                    print(warn, file=sys.stderr)
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        # dynamic symbol table = the C ABI of include/orn.h (+ orn_debug.h) only: kernel host stubs and internal launchers stay local
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + os.path.join(CSRC, 'orn.map'), '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
