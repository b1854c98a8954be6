"""N>1 path on CPU: world_size-2 gloo run of the sharding / timing / gather plumbing bench.py and the
CLI use on the GPU box over RCCL."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["ORN_ROOT"])
import torch
import orn_amd
from orn_amd import dist_utils as du
rank, local, world = du.env_world()
dist = du.init("gloo")
assert dist is not None and dist.get_world_size() == 2
vids = du.shard_videos(7, world, rank)
assert vids == ([0, 2, 4, 6] if rank == 0 else [1, 3, 5])
dt = du.max_over_ranks(dist, 1.0 + rank)          # rank 1 is slower
assert dt == 2.0, dt
# record layout: [psnr SUM over the rank's videos, videos fitted, frames, seconds, steps]; rank 0 fitted 4 videos, rank 1 three
nv = len(vids)
recs = du.gather_records(dist, [30.0 * nv + rank, float(nv), 100.0 * (rank + 1), 1.0 + rank, 100 * (rank + 1)])
assert len(recs) == 2 and recs[0][0] == 120.0 and recs[1][0] == 91.0 and recs[0][1] == 4.0 and recs[1][1] == 3.0
agg = du.aggregate(recs, dt)
assert abs(agg["frames_per_s"] - 150.0) < 1e-6 and agg["ranks"] == 2
assert agg["videos"] == 7 and abs(agg["mean_psnr"] - 211.0 / 7) < 1e-5, agg      # mean over the VIDEOS of the job
# a rank without a video (8 ranks, 7 videos) contributes nothing to the mean
idle = du.aggregate([[60.0, 2.0, 10.0, 1.0, 10.0], [0.0, 0.0, 0.0, 0.0, 0.0]], 1.0)
assert idle["videos"] == 2 and abs(idle["mean_psnr"] - 30.0) < 1e-6 and idle["ranks"] == 2, idle
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), ORN_ROOT=ROOT, OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f'rank {rank} ok' in out


def test_shard_videos_single_process():
    sys.path.insert(0, ROOT)
    from orn_amd import dist_utils as du
    seen = sorted(v for r in range(8) for v in du.shard_videos(7, 8, r))
    assert seen == list(range(7)) and du.shard_videos(7, 8, 7) == []
    assert du.max_over_ranks(None, 3.5) == 3.5
    assert du.gather_records(None, [1, 2, 3, 4, 5]) == [[1.0, 2.0, 3.0, 4.0, 5.0]]


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no outer launcher: the parent spawns two fresh rank processes before touching any GPU,
    they rendezvous (gloo here, RCCL on the GPU box), time the steps between barriers, reduce MAX over ranks, and rank 0
    prints the one JSON line with the rank count of the process group and the per-rank rates.  --cpu-stub replaces the
    engine by a sleep so the launcher / reduction path runs on a machine without a GPU."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--cpu-stub', '--steps', '6',
                          '--warmup', '2'], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and len(d['per_rank_frames_per_s']) == 2
    assert d['steps'] == 6 and d['warmup'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    # whole-job aggregate = ranks * steps / max-over-ranks time
    assert abs(d['value'] - 2 * 6 / (d['ms_per_step'] * 6 / 1e3)) < 1e-6 * d['value']


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies makes the launcher exit non-zero (WORLD_SIZE mismatch injected through --gpus vs a bad env)."""
    env = dict(os.environ, ORN_BENCH_FAIL_RANK='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--cpu-stub', '--steps', '2',
                          '--warmup', '1'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
