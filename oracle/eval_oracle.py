"""CPU ORACLE (test infrastructure only) for the evaluator rows f2/f3: numpy restatement of
utils/eval_metric.py:28-37 (project), :102-192 (projection_2d, add*_metric, cm_degree_5_metric) and of the serial
nearest-neighbour loop of thirdparty/nn/src/nearest_neighborhood.cu:48-81.
PINNED (f2): tests/test_oracle_golden.py::test_eval_oracle_matches_reference_eval_metric checks pose_metrics against
distances, threshold decisions and summarize() output produced by the reference's own utils/eval_metric.py
(tests/golden/gen_golden_eval.py imports it with empty stand-in modules for plyfile/open3d/cv2/transforms3d, none of which
the metric arithmetic touches).  The symmetric (ADD-S) vectors use the reference arithmetic around a numpy first-minimum
search: the CUDA extension thirdparty/nn cannot run here, so nn_idx itself (f3) stays PARITY UNPINNED against the .cu."""
import numpy as np


def nn_idx(ref, que, exclude_self=False):
    """first index of the minimum squared distance, fp32 arithmetic in the kernel's operation order"""
    ref, que = np.asarray(ref, np.float32), np.asarray(que, np.float32)
    d = np.zeros((que.shape[0], ref.shape[0]), np.float32)
    for k in range(ref.shape[1]):
        diff = ref[None, :, k] - que[:, None, k]
        d = d + diff * diff if k else diff * diff
    if exclude_self:
        n = min(d.shape)
        d[np.arange(n), np.arange(n)] = np.inf
    return np.argmin(d, axis=1).astype(np.int32)


def project(xyz, K, RT):
    xyz = np.dot(xyz, RT[:, :3].T) + RT[:, 3:].T
    xyz = np.dot(xyz, K.T)
    return xyz[:, :2] / xyz[:, 2:]


def pose_metrics(model, pose_pred, pose_gt, K, symmetric):
    model = np.asarray(model, np.float64)
    out = []
    for Tp, Tg in zip(np.asarray(pose_pred, np.float64), np.asarray(pose_gt, np.float64)):
        mp = model @ Tp[:, :3].T + Tp[:, 3]
        mg = model @ Tg[:, :3].T + Tg[:, 3]
        add = np.mean(np.linalg.norm(mp - mg, axis=-1))
        adds = -1.0
        if symmetric:
            idx = nn_idx(mp.astype(np.float32), mg.astype(np.float32))
            adds = np.mean(np.linalg.norm(mp.astype(np.float32).astype(np.float64)[idx] - mg.astype(np.float32).astype(np.float64), 2, 1))
        proj = np.mean(np.linalg.norm(project(model, np.asarray(K, np.float64), Tp) - project(model, np.asarray(K, np.float64), Tg), axis=-1))
        trans = np.linalg.norm(Tp[:, 3] - Tg[:, 3]) * 100
        tr = min(np.trace(Tp[:, :3] @ Tg[:, :3].T), 3.0)
        out.append([add, adds, proj, trans, np.rad2deg(np.arccos((tr - 1.0) / 2.0))])
    return np.array(out)
