"""BASELINE config 4 ("all 7 UVG 1080p sequences, ERB, one video per GPU") as far as one GPU can take it: the job's seven
independent fits at the 1080p geometry (fc_hw_dim 9_16_48, strides 5 3 2 2 2) through the reference CLI mirror -- as ONE rank
(its single-GPU share: all seven videos, shard_videos(7, 1, 0)) and as TWO fresh processes that share GPU 0 (gloo for the
final gather; RCCL needs one GPU per rank).  SURVEY 8(e); /root/reference has no multi-GPU path to compare with
(main_train.py:156-157 trains in-process on one device), so what is asserted is the sharding contract: every video fitted
exactly once, into its own directory, finite PSNR, one gathered record."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLAGS = ('-e 2 --lower_width 96 --num_blocks 1 --dataset uvg --frame_gap 1 --embed 1.25_40 --stem_dim_num 512_1 '
         '--reduction 2 --fc_hw_dim 9_16_48 --expansion 1 --single_res --loss Fusion6 --warmup 0.2 --lr_type cosine '
         '--strides 5 3 2 2 2 --conv_type conv -b 1 --lr 0.0005 --norm none --act swish --outf uvg7 --branch_type ERB '
         '--synthetic 4 --synthetic_videos 7 --eval_freq 2').split()
FILES = ('model_latest.pth', 'model_latest_deploy.pth', 'model_train_best.pth', 'model_train_best_deploy.pth', 'model_val_best.pth')


def test_config4_single_gpu_share(tmp_path, monkeypatch, capsys):
    import orn_amd
    from orn_amd import checkpoint, dist_utils, main_train
    monkeypatch.chdir(tmp_path)
    assert dist_utils.shard_videos(7, 1, 0) == list(range(7))
    best = main_train.train(main_train.parse_args(FLAGS))
    assert sorted(best) == [f'synthetic{v}' for v in range(7)]
    vals = list(best.values())
    assert all(5.0 < v < 60.0 for v in vals) and len({round(v, 4) for v in vals}) == 7      # seven different videos
    out = tmp_path / 'result' / 'uvg7'
    for v in range(7):
        d = out / f'synthetic{v}'
        for f in FILES + ('rank0.txt',):
            assert (d / f).exists(), (v, f)
        ck = torch.load(d / 'model_latest_deploy.pth', map_location='cpu', weights_only=True)
        assert checkpoint.state_dict_kind(ck['state_dict']) == 'deploy' and ck['epoch'] == 2
        assert ck['state_dict']['layers.1.rbr_reparam.weight'].shape == (864, 48, 3, 3)    # the stride-3 block of the 1080p geometry
        assert abs(float(ck['train_best_psnr']) - best[f'synthetic{v}']) < 1e-4
    txt = capsys.readouterr().out
    assert "'videos': 7" in txt and "'ranks': 1" in txt


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["ORN_ROOT"])
import orn_amd
from orn_amd import main_train
best = main_train.train(main_train.parse_args(sys.argv[1:]))
print("RANK", os.environ["RANK"], "VIDEOS", sorted(best), flush=True)
'''


def test_config4_two_ranks_share_one_gpu(tmp_path):
    """The same CLI under two fresh child processes (spawned, never re-exec'ed), RANK 0 / 1, both on GPU 0
    (LOCAL_RANK % device_count), gloo backend: disjoint video sets, disjoint output directories, one gathered record."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    flags = [f if f != '2' or FLAGS[i - 1] != '-e' else '1' for i, f in enumerate(FLAGS)] + ['--dist_backend', 'gloo', '--precision', 'bf16']
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), ORN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)] + flags, env=env, cwd=str(tmp_path),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
    assert "VIDEOS ['synthetic0', 'synthetic2', 'synthetic4', 'synthetic6']" in outs[0]
    assert "VIDEOS ['synthetic1', 'synthetic3', 'synthetic5']" in outs[1]
    assert "'videos': 7" in outs[0] and "'ranks': 2" in outs[0]
    out = tmp_path / 'result' / 'uvg7'
    for v in range(7):
        d = out / f'synthetic{v}'
        assert (d / 'model_latest.pth').exists() and (d / f'rank{v % 2}.txt').exists() and not (d / f'rank{1 - v % 2}.txt').exists()
