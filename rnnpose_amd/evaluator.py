"""LINEMOD pose evaluation on device ("next" rows): the per-sample metrics of
utils/eval_metric.py:102-192 (`LineMODEvaluator`) with model points resident on the GPU, plus the mirror of
thirdparty/nn/nn_utils.py:6-24 (`find_nearest_point_idx`) over the drop-in `findNearestPointIdxLauncher`.

Only the metric arithmetic is rebuilt; PLY loading, ICP refinement and visualisation stay outside (out of scope).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from .distributed import METRICS, MetricAccumulator

SYMMETRIC_CLASSES = ("eggbox", "glue")                      # utils/eval_metric.py:329
LINEMOD_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=np.float32)
#                                                           data/linemod/linemod_config.py:23-25


def find_nearest_point_idx(ref_pts: np.ndarray, que_pts: np.ndarray) -> np.ndarray:
    """numpy in / numpy out, same contract as thirdparty/nn/nn_utils.py:6-24 (host buffers, synchronous)."""
    assert ref_pts.shape[1] == que_pts.shape[1] and 1 < que_pts.shape[1] <= 3
    pn1, pn2, dim = ref_pts.shape[0], que_pts.shape[0], ref_pts.shape[1]
    ref = np.ascontiguousarray(ref_pts[None], np.float32)
    que = np.ascontiguousarray(que_pts[None], np.float32)
    idxs = np.zeros([1, pn2], np.int32)
    _lib.load().findNearestPointIdxLauncher(ref.ctypes.data_as(C.c_void_p), que.ctypes.data_as(C.c_void_p),
                                           idxs.ctypes.data_as(C.c_void_p), 1, pn1, pn2, dim, 0)
    return idxs[0]


class LineMODEvaluator:
    """evaluate(pose_pred (B,3,4), pose_gt (B,3,4)) accumulates ADD(-S) @0.1/0.02/0.05 d, proj2d<5px, 5cm5deg."""

    def __init__(self, class_name: str, model_points, diameter: float, K=None, device="cuda"):
        self.class_name = class_name
        self.symmetric = class_name in SYMMETRIC_CLASSES
        self.model = torch.as_tensor(np.asarray(model_points, dtype=np.float32)).to(device)
        self.diameter = float(diameter)
        self.K = torch.as_tensor(LINEMOD_K if K is None else np.asarray(K, dtype=np.float32)).to(device)
        self.acc = MetricAccumulator((class_name,))
        self.last = None

    def evaluate(self, pose_pred, pose_gt, unique=None):
        m = ops.pose_metrics(self.model, pose_pred[..., :3, :].reshape(-1, 3, 4), pose_gt[..., :3, :].reshape(-1, 3, 4),
                             self.K, self.symmetric)
        self.last = m
        dist = m[:, 1] if self.symmetric else m[:, 0]
        flags = torch.stack([dist < 0.1 * self.diameter, dist < 0.02 * self.diameter, dist < 0.05 * self.diameter,
                             m[:, 2] < 5.0, (m[:, 3] < 5.0) & (m[:, 4] < 5.0)], 1).cpu().numpy()   # one D2H per batch
        for i, row in enumerate(flags):
            self.acc.update(self.class_name, dict(zip(METRICS, row.astype(float))),
                            unique=True if unique is None else bool(unique[i]))
        return m

    def summarize(self):
        """Cross-rank means (one RCCL all-reduce) with the reference's key names (utils/eval_metric.py:261-302)."""
        r = self.acc.reduce()[self.class_name]
        return {"proj2d": r["proj2d"], "add": r["add"], "add2": r["add2"], "add5": r["add5"], "cmd5": r["cmd5"],
                "seq_len": r["n"]}
