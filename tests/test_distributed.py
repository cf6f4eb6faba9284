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


# ---- r05: readiness for the driver's 8-GPU run (no multi-GPU box is reachable from the build side) ----------
def _lazy_load_worker(rank, world, port, q):
    """What bench.py does on every rank, in bench.py's order: import the package and its ops BEFORE distributed.build_once.  The
    shared library must not be dlopen'ed by that (ranks > 0 would map a file rank 0 is still writing): _lib.load() is lazy."""
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rnnpose_amd import distributed as D
    D.init_from_env(backend="gloo")
    from rnnpose_amd import _lib, build, ops  # noqa: F401  (bench.py:240: `from rnnpose_amd import build, ops`)
    from rnnpose_amd.pose_refiner import PoseRefiner, default_config  # noqa: F401
    mapped = lambda: any("librnnpose_hip" in line for line in open("/proc/self/maps"))
    before = (_lib._lib is None, mapped())
    stamp = {}

    def fake_build():                      # rank 0 "compiles" for a while; the others must still be waiting when it returns
        if rank == 0:
            time.sleep(1.0)
        stamp["t"] = time.time()
        return build.LIB
    t_enter = time.time()
    path = D.build_once(fake_build)
    after = (_lib._lib is None, mapped())
    q.put((rank, before, after, path == build.LIB, stamp["t"] - t_enter))
    torch.distributed.destroy_process_group()


def test_ranks_do_not_dlopen_the_library_before_build_once_returns():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lazy_load_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, before, after, same, waited in outs:
        assert before == (True, False), f"rank {rank} loaded the library while importing the package"
        assert after == (True, False) and same                 # build_once itself does not load it either
        assert waited >= 0.9, f"rank {rank} ran its build step {waited:.2f} s after entering build_once: it did not wait for rank 0"


def test_bench_argument_plumbing_for_the_driver_launch(monkeypatch):
    """`python bench.py --gpus 8 --workload cfg4 ...` outside torchrun re-launches itself as 8 ranks on 127.0.0.1 with the arguments
    passed through; under torchrun (RANK / WORLD_SIZE set) main() must not spawn again.  cfg4 = BASELINE.json configs[4]."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1", "--workload", "cfg4"]
    monkeypatch.setattr(sys, "argv", argv)
    args = bench.parse()
    assert (args.gpus, args.steps, args.warmup) == (8, 5, 1)
    assert (args.batch, args.height, args.width, args.outer, args.inner) == (8, 960, 1280, 3, 8)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(8)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv):] == [os.path.join(root, "bench.py")] + argv[1:]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:                      # fewer devices than ranks: refuse instead of oversubscribing
        bench.spawn_ranks(8)
    assert "only 1 GPU" in str(e.value.code)
