"""GPU tests of the evaluator rows: nearest neighbour (bit-exact indices, both the
device entry point and the reference's host-pointer cffi symbol) and the LINEMOD pose metrics."""
import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo
from rnnpose_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rnnpose_amd import build, ops as _ops
    build.build()
    return _ops


@pytest.mark.parametrize("n1,n2,dim", [(1, 1, 3), (5, 300, 3), (1500, 1031, 3), (2100, 777, 2), (4096, 4096, 3)])
def test_nn_search_bit_exact(ops, n1, n2, dim):
    B = 2
    ref = syn.uniform("ref", (B, n1, dim), 1, -0.1, 0.1)
    que = syn.uniform("que", (B, n2, dim), 1, -0.1, 0.1)
    if n1 > 10 and n2 > 10:
        que[:, :5] = ref[:, 3:8]                 # exact hits
        ref[:, 9] = ref[:, 4]                    # duplicate reference point: the FIRST index must win
    idx = ops.nn_search(torch.from_numpy(ref).cuda(), torch.from_numpy(que).cuda()).cpu().numpy()
    for b in range(B):
        assert np.array_equal(idx[b], eo.nn_idx(ref[b], que[b]))
    if n1 == n2:
        idx = ops.nn_search(torch.from_numpy(ref).cuda(), torch.from_numpy(ref).cuda(), exclude_self=True).cpu().numpy()
        assert np.array_equal(idx[0], eo.nn_idx(ref[0], ref[0], exclude_self=True))
        assert n1 == 1 or np.all(idx[0] != np.arange(n1))      # a lone point has nobody else: index 0, as the reference


def test_reference_cffi_symbol(ops):
    """findNearestPointIdxLauncher with HOST pointers == nn_utils.find_nearest_point_idx semantics."""
    from rnnpose_amd.evaluator import find_nearest_point_idx
    ref = syn.uniform("r", (3000, 3), 2, -0.2, 0.2)
    que = syn.uniform("q", (2500, 3), 2, -0.2, 0.2)
    assert np.array_equal(find_nearest_point_idx(ref, que), eo.nn_idx(ref, que))
    assert np.array_equal(find_nearest_point_idx(ref[:, :2], que[:, :2]), eo.nn_idx(ref[:, :2], que[:, :2]))


@pytest.mark.parametrize("sym", [False, True])
def test_pose_metrics(ops, sym):
    P, B = 3000, 4
    model = syn.uniform("model", (P, 3), 3, -0.08, 0.08)
    gt = syn.se3_exp_np(syn.normal("gt", (B, 6), 3, std=0.3))[:, :3].astype(np.float32)
    gt[:, 2, 3] += 0.9
    d = syn.normal("dxi", (B, 6), 4, std=1.0) * np.array([[0.002, 0.002, 0.002, 0.01, 0.01, 0.01]]) * np.arange(1, B + 1)[:, None]
    pred = (syn.se3_exp_np(d) @ np.concatenate([gt, np.tile([[[0, 0, 0, 1]]], (B, 1, 1))], 1))[:, :3].astype(np.float32)
    from rnnpose_amd.evaluator import LINEMOD_K
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    got = ops.pose_metrics(t(model), t(pred), t(gt), t(LINEMOD_K), sym).cpu().numpy()
    want = eo.pose_metrics(model, pred, gt, LINEMOD_K, sym)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-9), (got, want)
    assert (got[:, 1] >= 0).all() == sym
    if sym:
        assert np.all(got[:, 1] <= got[:, 0] + 1e-12)        # ADD-S never exceeds ADD


def test_evaluator_accumulates_like_the_reference(ops):
    from rnnpose_amd.evaluator import LineMODEvaluator
    P = 2000
    model = syn.uniform("model", (P, 3), 5, -0.05, 0.05)
    ev = LineMODEvaluator("cat", model, diameter=0.15)
    gt = np.tile(np.eye(4, dtype=np.float32)[None], (3, 1, 1))
    gt[:, 2, 3] = 1.0
    pred = gt.copy()
    pred[1, 0, 3] += 0.001       # 1 mm: inside every threshold
    pred[2, 0, 3] += 0.02        # 2 cm: fails ADD@0.1d (1.5 cm), passes 5cm
    ev.evaluate(torch.from_numpy(pred).cuda()[:, None], torch.from_numpy(gt).cuda()[:, None])
    s = ev.summarize()
    assert s["seq_len"] == 3 and abs(s["add"] - 2 / 3) < 1e-12 and abs(s["add2"] - 2 / 3) < 1e-12
    assert s["cmd5"] == 1.0 and abs(s["proj2d"] - 2 / 3) < 1e-12


@pytest.mark.parametrize("pre,sym,cls", [("cat_", False, "cat"), ("driller_", False, "driller"), ("sym_eggbox_", True, "eggbox")])
def test_pose_metrics_vs_reference_eval_metric(ops, golden, pre, sym, cls):
    """Row f2 pinned to the reference: distances, threshold decisions and summarize() of utils/eval_metric.py:102-192,
    261-302, run on the reference's own module (tests/golden/gen_golden_eval.py)."""
    from test_oracle_golden import EVAL_DIAMETERS, check_eval_against_reference, eval_flags
    from rnnpose_amd.evaluator import LineMODEvaluator
    g = golden("eval_metric")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    K = g["linemod_K"].astype(np.float32)
    m = ops.pose_metrics(t(g[pre + "model"]), t(g[pre + "pred"]), t(g[pre + "gt"]), t(K), sym).cpu().numpy()
    check_eval_against_reference(m, g, pre, sym)
    assert np.array_equal(eval_flags(m, EVAL_DIAMETERS[pre], sym), g[pre + "flags"])
    ev = LineMODEvaluator(cls, g[pre + "model"], diameter=EVAL_DIAMETERS[pre])
    ev.evaluate(t(g[pre + "pred"])[:, None], t(g[pre + "gt"])[:, None])
    s = ev.summarize()
    want = g[pre + "flags"].mean(0)
    assert np.allclose([s["add"], s["add2"], s["add5"], s["proj2d"], s["cmd5"]], want) and s["seq_len"] == g[pre + "flags"].shape[0]
    if not sym:
        r = g[pre + "summary"]
        assert np.allclose([s["proj2d"], s["add"], s["add2"], s["add5"], s["cmd5"], s["seq_len"]], r)
