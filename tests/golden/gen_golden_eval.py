#!/usr/bin/env python3
"""Golden vectors for row f2 (evaluator) from the REFERENCE's own `utils/eval_metric.py` (build container only).

    python tests/golden/gen_golden_eval.py        # writes tests/golden/eval_metric.npz

`utils/eval_metric.py` imports plyfile / open3d / transforms3d / cv2 / the compiled cffi extension at module level; none
of them is touched by the metric arithmetic (`project`, `add*_metric`, `projection_2d`, `cm_degree_5_metric`,
utils/eval_metric.py:28-37,102-192), so empty stand-in MODULES satisfy the imports and the reference's own functions run
unchanged.  `LineMODEvaluator.__init__` (reads a .ply from EXPDATA) is bypassed with `object.__new__`; the fields it would
set (`model`, `diameter`, the result lists) are set here.

The metric methods only append booleans.  To pin the underlying DISTANCES too, the module's `np` global is replaced by a
recording proxy (same numpy underneath): every `np.mean(...)`, `np.linalg.norm(...)` and `np.rad2deg(...)` result the
reference computes is captured as it is produced.

Symmetric classes (eggbox, glue) call the CUDA nearest-neighbour extension (thirdparty/nn).  It cannot run here, so for the
`*_sym` arrays the extension's entry point is served by a serial first-minimum search (the loop of
thirdparty/nn/src/nearest_neighborhood.cu:48-81 in numpy fp32); everything around it is the reference's arithmetic.  Those
arrays are labelled `sym_` and pin the reference arithmetic GIVEN the neighbour indices; the indices themselves stay
pinned only by the brute-force oracle.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path[:0] = [ROOT, REF, os.path.join(REF, "thirdparty")]


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def nn_first_min(ref_pts, que_pts):
    ref, que = np.asarray(ref_pts, np.float32), np.asarray(que_pts, np.float32)
    d = np.zeros((que.shape[0], ref.shape[0]), np.float32)
    for k in range(ref.shape[1]):
        diff = ref[None, :, k] - que[:, None, k]
        d = d + diff * diff if k else diff * diff
    return np.argmin(d, axis=1).astype(np.int32)


# `data/__init__.py` imports the dataset classes (torchvision, cv2 ...): register the PACKAGE without running its __init__
# so that `data.linemod.linemod_config` (plain constants) is the reference's own file
_stub("data").__path__ = [os.path.join(REF, "data")]
_stub("plyfile", PlyData=object)
_stub("open3d")
_stub("cv2")
_stub("transforms3d")
_stub("transforms3d.quaternions", mat2quat=None, quat2mat=None, qmult=None)
_stub("thirdparty.vsd")
_stub("thirdparty.vsd.inout")
_stub("thirdparty.nn")
nn_utils = _stub("thirdparty.nn.nn_utils", find_nearest_point_idx=nn_first_min)
sys.modules["thirdparty.nn"].nn_utils = nn_utils
sys.modules["thirdparty.vsd"].inout = sys.modules["thirdparty.vsd.inout"]
_stub("utils.img_utils", read_depth=None)
_stub("thirdparty.kpconv")
_stub("thirdparty.kpconv.lib")
_stub("thirdparty.kpconv.lib.utils", square_distance=None)

import utils.eval_metric as em  # noqa: E402  (the reference module)
from data.linemod import linemod_config  # noqa: E402

from rnnpose_amd import synthetic as syn  # noqa: E402


class _Rec:
    """numpy proxy that records the results of the calls the metric methods make"""

    def __init__(self, target, names, log, prefix=""):
        self._t, self._names, self._log, self._p = target, names, log, prefix

    def __getattr__(self, k):
        v = getattr(self._t, k)
        if k == "linalg":
            return _Rec(v, self._names, self._log, "linalg.")
        if self._p + k in self._names:
            def f(*a, **kw):
                r = v(*a, **kw)
                self._log.append((self._p + k, r))
                return r
            return f
        return v


def run(class_name, model, pred, gt):
    ev = object.__new__(em.LineMODEvaluator)
    ev.class_name = class_name
    ev.model = model
    ev.diameter = linemod_config.diameters[class_name] / 100            # utils/eval_metric.py:77
    for k in ("proj2d", "add", "adds", "add2", "add5", "cmd5", "icp_proj2d", "icp_add", "icp_cmd5", "mask_ap", "pose_preds"):
        setattr(ev, k, [])
    sym = class_name in ["eggbox", "glue"]                               # :329
    log = []
    em.np = _Rec(np, {"mean", "linalg.norm", "rad2deg"}, log)
    raw = {"add": [], "proj2d": [], "trans_cm": [], "rot_deg": []}
    try:
        for Tp, Tg in zip(pred, gt):
            del log[:]
            ev.add_metric(Tp, Tg, syn=sym)                               # :330-336
            raw["add"].append(float([r for n, r in log if n == "mean"][-1]))
            ev.add2_metric(Tp, Tg, syn=sym)
            ev.add5_metric(Tp, Tg, syn=sym)
            del log[:]
            ev.projection_2d(Tp, Tg, K=linemod_config.linemod_K)         # :338
            raw["proj2d"].append(float([r for n, r in log if n == "mean"][-1]))
            del log[:]
            ev.cm_degree_5_metric(Tp, Tg)                                # :339
            raw["trans_cm"].append(float([r for n, r in log if n == "linalg.norm"][-1]) * 100)
            raw["rot_deg"].append(float([r for n, r in log if n == "rad2deg"][-1]))
    finally:
        em.np = np
    flags = np.stack([np.asarray(getattr(ev, k), bool) for k in ("add", "add2", "add5", "proj2d", "cmd5")], 1)
    summary = None
    if not sym:
        ev.mask_ap = [True]
        ev.icp_refine = False
        s = ev.summarize()                                               # :261-302
        summary = np.array([s[k] for k in ("proj2d", "add", "add2", "add5", "cmd5", "seq_len")], np.float64)
    return flags, {k: np.array(v, np.float64) for k, v in raw.items()}, summary


def poses(n, seed):
    """ground-truth poses in front of the camera and predictions perturbed over 4 decades, so every threshold
    (0.02/0.05/0.1 diameter, 5 px, 5 cm, 5 deg) is crossed inside the set"""
    gt = syn.se3_exp_np(syn.normal("gt", (n, 6), seed, std=0.4)).astype(np.float64)
    gt[:, 2, 3] += 0.9
    mag = np.logspace(-4, -0.3, n)[:, None]
    d = syn.normal("dxi", (n, 6), seed + 1, std=1.0) * mag * np.array([[0.3, 0.3, 0.3, 1.0, 1.0, 1.0]])
    pred = syn.se3_exp_np(d) @ gt
    pred[n - 1] = gt[n - 1]                      # exact hit: trace == 3 (+ rounding) exercises the clamp of :183
    return pred[:, :3].astype(np.float32), gt[:, :3].astype(np.float32)


def main():
    n, P = 48, 2500
    out = {}
    for cls, seed in (("cat", 21), ("driller", 23), ("eggbox", 25)):
        model = syn.uniform("model_" + cls, (P, 3), seed, -0.5, 0.5) * (linemod_config.diameters[cls] / 100) * np.array([[1.0, 0.7, 0.5]])
        model = model.astype(np.float32)
        pred, gt = poses(n, seed)
        flags, raw, summary = run(cls, model, pred, gt)
        pre = ("sym_" if cls == "eggbox" else "") + cls + "_"
        out[pre + "model"], out[pre + "pred"], out[pre + "gt"] = model, pred, gt
        out[pre + "flags"] = flags
        for k, v in raw.items():
            out[pre + k] = v
        if summary is not None:
            out[pre + "summary"] = summary
        print(cls, "flags true per metric:", flags.sum(0), "of", n)
    out["linemod_K"] = np.asarray(linemod_config.linemod_K, np.float64)
    path = os.path.join(HERE, "eval_metric.npz")
    np.savez_compressed(path, **out)
    print(f"  eval_metric.npz  {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
