"""world_size-2 gloo tests of the N>1 path: sharding + the single epoch-end all_reduce + the MAX clock."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rnnpose_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    idx, uniq = D.shard_indices(n_items, r, w)
    acc = D.MetricAccumulator(("cat", "glue"))
    for i, u in zip(idx, uniq):
        cls = "cat" if i % 3 else "glue"
        acc.update(cls, dict(add=float(i % 2), add2=1.0, proj2d=float(i)), unique=u)
    D.barrier()
    res = acc.reduce()
    t = D.max_over_ranks(1.0 + rank)
    # first-contact helpers of bench.py
    assert D.ranks_seen() == world and D.backend_name() == "gloo"
    assert D.gather_values(rank + 0.5) == [r + 0.5 for r in range(world)]
    calls = []
    assert D.build_once(lambda: calls.append(rank) or "lib") == "lib" and calls == [rank]      # every rank ends with the build result
    nthr = D.pin_host_threads(rank, world, max_threads=4)
    assert 1 <= nthr <= 4 and torch.get_num_threads() == nthr
    q.put((rank, res, t, idx))
    torch.distributed.destroy_process_group()


def test_two_rank_metric_reduction():
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process ground truth over the UNIQUE items
    items = list(range(n_items))
    cat = [i for i in items if i % 3]
    glue = [i for i in items if not i % 3]
    for rank, res, t, idx in outs:
        assert t == 2.0                                           # MAX over ranks
        assert res["cat"]["n"] == len(cat) and res["glue"]["n"] == len(glue)
        assert abs(res["cat"]["add"] - sum(i % 2 for i in cat) / len(cat)) < 1e-12
        assert abs(res["glue"]["proj2d"] - sum(glue) / len(glue)) < 1e-12
        assert res["cat"]["add2"] == 1.0
    assert sorted(outs[0][3] + outs[1][3]) == sorted(items + [0])    # one wrap-around duplicate


def _nccl_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from rnnpose_amd import distributed as D
    r, w, local = D.init_from_env()                       # backend "nccl" = RCCL, communicator bound to this rank's device
    assert D.backend_name() == "nccl" and torch.cuda.current_device() == local
    acc = D.MetricAccumulator(("cat",))
    acc.update("cat", dict(add=float(rank)), unique=True)
    res = acc.reduce()
    q.put((rank, D.ranks_seen(), D.gather_values(10.0 + rank), D.max_over_ranks(1.0 + rank), res["cat"]["add"], res["cat"]["n"]))
    D.barrier()
    torch.distributed.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two devices (one rank per GPU): first executed by the driver's multi-GPU run")
def test_two_rank_rccl_on_two_devices():
    """The `nccl` (= RCCL) branch of distributed.init_from_env on real devices: one rank per GPU, device-bound communicator, the
    packed metric all_reduce(SUM), the MAX clock, ranks_seen.  Skipped on single-GPU boxes (every gpurun lease of this build)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, seen, vals, t, add, n in outs:
        assert seen == 2 and vals == [10.0, 11.0] and t == 2.0 and n == 2 and abs(add - 0.5) < 1e-12
