"""One process per GPU; images shard embarrassingly; ONE collective at the end of an evaluation epoch.

Reference behaviour being replaced (SURVEY.md sections 2 row 27-28, 8e):
  * `DistributedSequatialSampler` gives rank r the indices r, r+n, r+2n, ... of the evaluation set, padded by
    wrap-around to a multiple of n (utils/distributed_utils.py:150-169, used at tools/eval.py:471);
  * tools/eval.py performs NO cross-rank reduction (each rank prints its own means, :560-562); tools/train.py
    all-gathers (mean*count, count) scalars per class and metric (:725-741).
Here: the same rank-strided shard, but wrap-around duplicates are masked out of the statistics, and the
per-class sums travel in ONE all_reduce(SUM) of a packed fp64 vector (<= 1 KB: latency-bound on xGMI, the
ring/tree choice is irrelevant).  backend "nccl" is RCCL on ROCm; CPU tests use "gloo".
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import torch
import torch.distributed as dist

METRICS = ("add", "add2", "add5", "proj2d", "cmd5")      # utils/eval_metric.py:261-302


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's contract).
    Returns (rank, world_size, local_rank).  No-op (0,1,0) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world <= 1:
        return 0, 1, 0
    if backend is None:
        # RNNPOSE_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- how the multi-rank path of
        # bench.py / eval_epoch is exercised on a single-GPU box
        backend = os.environ.get("RNNPOSE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():      # bind the communicator to this rank's GPU up front
            kw["device_id"] = torch.device("cuda", local % torch.cuda.device_count())
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:                                        # older torch: no device_id argument
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int):
    """Rank-strided shard with wrap-around padding -> (indices, is_unique mask).
    indices follow utils/distributed_utils.py:150-169; is_unique is False for the padded duplicates."""
    if n_items <= 0:
        return [], []
    total = ((n_items + world - 1) // world) * world
    order = list(range(n_items)) + list(range(total - n_items))
    mine = order[rank:total:world]
    positions = list(range(rank, total, world))
    uniq = [p < n_items for p in positions]
    return mine, uniq


@dataclass
class MetricAccumulator:
    """Per-class running sums of the LINEMOD metrics + sample count; reduce() is the epoch-end collective."""
    classes: tuple
    sums: torch.Tensor = field(init=False)

    def __post_init__(self):
        self.sums = torch.zeros(len(self.classes), len(METRICS) + 1, dtype=torch.float64)

    def update(self, cls: str, values: dict, unique: bool = True):
        if not unique:
            return
        r = self.classes.index(cls)
        for k, name in enumerate(METRICS):
            self.sums[r, k] += float(values.get(name, 0.0))
        self.sums[r, -1] += 1.0

    def reduce(self, device=None):
        """ONE all_reduce(SUM) of the packed (n_classes x 6) fp64 vector; returns {cls: {metric: mean, n}}."""
        buf = self.sums.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
            buf = buf.to(device)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            buf = buf.cpu()
        out = {}
        for r, cls in enumerate(self.classes):
            n = float(buf[r, -1])
            out[cls] = {m: (float(buf[r, k]) / n if n > 0 else float("nan")) for k, m in enumerate(METRICS)}
            out[cls]["n"] = int(n)
        return out


def max_over_ranks(seconds: float, device=None) -> float:
    """all_reduce(MAX) of one double -- the throughput clock of bench.py."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def backend_name() -> str:
    """"nccl" (= RCCL on ROCm) / "gloo" / "none" (single process)."""
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"


def _coll_device(device=None):
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if backend_name() == "nccl" else "cpu"


def ranks_seen(device=None) -> int:
    """all_reduce(SUM) of a one per rank: how many ranks the collective actually reached (bench.py prints it next to n_gpus,
    so that a line produced by fewer ranks than it claims cannot pass)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 1
    t = torch.ones(1, dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def gather_values(value: float, device=None) -> list:
    """Every rank's `value` on every rank, in rank order (one all_reduce(SUM) of a one-hot vector: the path has no other
    collective to piggy-back on)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=_coll_device(device))
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.cpu()]


def build_once(build_fn):
    """Rank 0 compiles (or finds the stamp) while the others wait at a barrier, then every rank calls build_fn -- a stamp
    check for all but rank 0.  N ranks racing for the build lock is correct (rnnpose_amd/build.py holds a file lock) but makes
    N-1 processes sit in flock behind a minute of hipcc; this makes the order explicit."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    out = build_fn() if rank == 0 else None
    barrier()
    return out if rank == 0 else build_fn()


def pin_host_threads(local_rank: int, local_world: int, max_threads: int = 16) -> int:
    """Give this rank its own slice of the host cores (sched_setaffinity) and cap torch's intra-op threads: N ranks x
    torch.get_num_threads() = all cores each would oversubscribe the host as soon as any rank does CPU work (the oracle legs of
    bench.py run at world size 1 only, but tensor construction / hashing of synthetic inputs does not).  -> threads in use."""
    n = os.cpu_count() or 1
    per = max(1, n // max(1, local_world))
    lo = (local_rank % max(1, local_world)) * per
    try:
        avail = sorted(os.sched_getaffinity(0))
        if len(avail) >= local_world:                    # slice what this process is actually allowed to use
            per = max(1, len(avail) // local_world)
            mine = avail[(local_rank % local_world) * per:(local_rank % local_world + 1) * per]
        else:
            mine = list(range(lo, min(n, lo + per)))
        if mine:
            os.sched_setaffinity(0, mine)
            per = len(mine)
    except (AttributeError, OSError):
        pass
    t = max(1, min(per, max_threads))
    torch.set_num_threads(t)
    return t
