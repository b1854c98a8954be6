"""Multi-GPU plumbing: independent per-video fits, one per rank.  The data path has no
collective; torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests)
is used only for the barrier, the max-over-ranks wall time and the end-of-run result gather."""
import datetime
import os
from typing import Dict, List, Sequence

import torch

# Ranks fit different videos and meet only in the end-of-job gather: a rank with fewer or shorter videos waits there for the
# slowest one, for as long as a whole fit takes.  torch's default collective timeout (10 min for NCCL) would abort it -- and
# with it every rank still fitting -- so the group gets a timeout no fit reaches (ORN_DIST_TIMEOUT_S overrides it).
DEFAULT_TIMEOUT_S = 7 * 24 * 3600


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def shard_videos(n_videos: int, world: int, rank: int) -> List[int]:
    """Video ids fitted by `rank`: round-robin, every video exactly once (7 UVG sequences on 8 GPUs
    leave rank 7 idle; 132-frame replicas for the scaling runs give one per rank)."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world {world}')
    return list(range(rank, n_videos, world))


def init(backend: str = None, timeout_s: float = None):
    """init_process_group from the environment; returns the module or None for a single process."""
    rank, _, world = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if timeout_s is None:
        timeout_s = float(os.environ.get('ORN_DIST_TIMEOUT_S', DEFAULT_TIMEOUT_S))
    if not dist.is_initialized():
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        # nccl (= RCCL): bind the communicator to this rank's GPU at creation instead of relying on the current device
        kw = {'device_id': torch.device('cuda', env_world()[1])} if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return dist


def max_over_ranks(dist, seconds: float, device='cpu') -> float:
    """The job's wall time = the slowest rank's."""
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, record: Sequence[float], device='cpu') -> List[List[float]]:
    """all_gather of one small per-rank record [psnr sum over its videos, videos fitted, frames, seconds, steps] (20 B/rank)."""
    rec = torch.tensor(list(record), dtype=torch.float32, device=device)
    if dist is None:
        return [rec.tolist()]
    out = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return [o.tolist() for o in out]


def aggregate(records: List[List[float]], job_seconds: float) -> Dict[str, float]:
    """Whole-job numbers from the gathered records: total frames/s over the slowest rank's time; the PSNR is the mean over
    the VIDEOS of the job (a rank without a video contributes nothing to it)."""
    frames = sum(r[2] for r in records)
    videos = sum(r[1] for r in records)
    return {'frames_per_s': frames / job_seconds if job_seconds > 0 else 0.0,
            'mean_psnr': sum(r[0] for r in records) / videos if videos > 0 else 0.0, 'videos': int(videos), 'ranks': len(records)}
