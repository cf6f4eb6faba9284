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
