"""Format-level readers for the reference's evaluation inputs.

`EXPDATA/` (LINEMOD / LM-O frames, PoseCNN / PVNet initial poses, trained `.tckpt` files) is not part of this build, so
BASELINE configs 0 / 2 / 3 cannot be reproduced here; these functions read the reference's ON-DISK FORMATS so that
`eval_epoch.run_epoch` runs on real data the day it is supplied.  Tested on files the tests write in those formats
(tests/test_data_io.py).  Image decoding (cv2 / PIL) stays with the caller: `frame_record` returns paths.

    format                                   reference                                    here
    <split>.info  pickle {seq: [frame dict]} data/linemod_dataset.py:41-58,143-164        load_info, frame_record
    frame dict: rgb_observed_path, depth_gt_observed_path, K (3,3), gt_pose (3,4), [pose_noisy_rendered, ...]   :311-343
    linemod_posecnn_results.pkl {cls: [{'pose': (7,) quaternion wxyz + t}]}  :177-181,339   InitPoses("POSECNN_LINEMOD")
    pvnet_linemod[occ]_test.npy  dict {cls: [(3,4)]} + blender2bop_RT.npy {cls: (3,4)}  :182-196,330-338   InitPoses("PVNET_*")
    *.tckpt  = torch.save(state_dict)        torchplus/train/checkpoint.py:92, tools/eval.py:386-413   render_adapter.load_motion_net_checkpoint
    crop + intrinsics update                 data/preprocess.py:181-255                   patch_crop_window
    camera / diameters                       data/linemod/linemod_config.py:2-25          LINEMOD_K, DIAMETERS_CM
"""
from __future__ import annotations

import os
import pickle

import numpy as np

LINEMOD_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=np.float32)
BLENDER_K = np.array([[700.0, 0.0, 320.0], [0.0, 700.0, 240.0], [0.0, 0.0, 1.0]], dtype=np.float32)
DIAMETERS_CM = {"cat": 15.2633, "ape": 9.74298, "benchvise": 28.6908, "bowl": 17.1185, "cam": 17.1593, "camera": 17.1593,
                "can": 19.3416, "cup": 12.5961, "driller": 25.9425, "duck": 10.7131, "eggbox": 17.6364, "glue": 16.4857,
                "holepuncher": 14.8204, "iron": 30.3153, "lamp": 28.5155, "phone": 20.8394}
LINEMOD_CLASSES = ["ape", "cam", "cat", "duck", "glue", "iron", "phone", "benchvise", "can", "driller", "eggbox", "holepuncher",
                   "lamp"]


def diameter_m(class_name: str) -> float:
    """utils/eval_metric.py uses the table in metres (diameters[cls] / 100)."""
    return DIAMETERS_CM[class_name] / 100.0


def load_info(info_paths, seq_names=None) -> dict:
    """One or several `.info` pickles ({sequence name: [frame dict, ...]}) merged the way LinemodDataset does
    (data/linemod_dataset.py:143-164): -> dict(seqs, seq_lengths, data, dataset_idx)."""
    if isinstance(info_paths, (str, bytes, os.PathLike)):
        info_paths = [info_paths]
    merged = None
    for di, path in enumerate(info_paths):
        with open(path, "rb") as f:
            info = pickle.load(f)
        if seq_names is not None:
            info = {k: v for k, v in info.items() if k in seq_names}
        conv = {"seqs": list(info.keys()), "seq_lengths": [len(info[k]) for k in info], "data": [info[k] for k in info]}
        conv["dataset_idx"] = [di] * len(conv["seqs"])
        if merged is None:
            merged = conv
        else:
            for k in merged:
                merged[k].extend(conv[k])
    return merged


def dataset_len(infos: dict) -> int:
    return int(sum(infos["seq_lengths"]))


def frame_record(infos: dict, idx: int, root_paths=("",)) -> dict:
    """Flat sample index -> (sequence, frame) (data/linemod_dataset.py:296-328) with absolute paths, K and the ground-truth
    pose as 4x4: dict(class_name, frame_idx, rgb_path, depth_path, K (3,3) f32, pose_gt (4,4) f32, pose_noisy_rendered|None)."""
    cum = np.concatenate([[0], np.cumsum(infos["seq_lengths"])])
    if not 0 <= idx < cum[-1]:
        raise IndexError(idx)
    s = int(np.searchsorted(cum, idx, side="right") - 1)
    fi = int(idx - cum[s])
    fr = infos["data"][s][fi]
    root = root_paths[infos["dataset_idx"][s]] if infos.get("dataset_idx") else root_paths[0]
    rec = dict(class_name=infos["seqs"][s], frame_idx=fi, rgb_path=os.path.join(root, fr["rgb_observed_path"]),
               depth_path=os.path.join(root, fr["depth_gt_observed_path"]) if fr.get("depth_gt_observed_path") else None,
               K=np.asarray(fr["K"], np.float32), pose_gt=to44(fr["gt_pose"]))
    nz = fr.get("pose_noisy_rendered")
    rec["pose_noisy_rendered"] = to44(nz) if nz is not None else None
    return rec


def to44(RT) -> np.ndarray:
    RT = np.asarray(RT, np.float32)
    out = np.eye(4, dtype=np.float32)
    out[:RT.shape[0], :4] = RT[:, :4]
    out[3] = [0, 0, 0, 1]
    return out


def quat2mat(q) -> np.ndarray:
    """Quaternion (w, x, y, z) -> 3x3 rotation: the transforms3d.quaternions.quat2mat the reference calls (se3_q2m,
    data/linemod_dataset.py:31-39); a (near-)zero quaternion gives the identity as transforms3d does."""
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ, xX, xY, xZ, yY, yZ, zZ = w * X, w * Y, w * Z, x * X, x * Y, x * Z, y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def se3_q2m(se3_q) -> np.ndarray:
    """(7,) [quaternion wxyz | translation] -> (3,4) [R|t] (data/linemod_dataset.py:31-39)."""
    se3_q = np.asarray(se3_q, np.float64).reshape(-1)
    assert se3_q.size == 7
    m = np.zeros((3, 4))
    m[:, :3] = quat2mat(se3_q[:4])
    m[:, 3] = se3_q[4:]
    return m


class InitPoses:
    """Initial poses of the evaluation split (data/linemod_dataset.py:177-196 loading, :330-343 selection)."""

    def __init__(self, kind="POSECNN_LINEMOD", posecnn_pkl=None, pvnet_npy=None, blender2bop_npy=None):
        self.kind = kind
        self.posecnn = None
        if posecnn_pkl is not None:
            with open(posecnn_pkl, "rb") as f:
                self.posecnn = pickle.load(f)
        self.pvnet = np.load(pvnet_npy, allow_pickle=True).flat[0] if pvnet_npy is not None else None
        self.blender2bop = np.load(blender2bop_npy, allow_pickle=True).flat[0] if blender2bop_npy is not None else None
        if kind == "POSECNN_LINEMOD" and self.posecnn is None:
            raise ValueError("POSECNN_LINEMOD needs linemod_posecnn_results.pkl")
        if kind.startswith("PVNET") and (self.pvnet is None or self.blender2bop is None):
            raise ValueError(f"{kind} needs the pvnet .npy file and the blender -> BOP frame file (data/linemod_dataset.py:186-196)")
        if kind not in ("POSECNN_LINEMOD", "PVNET_LINEMOD", "PVNET_LINEMOD_OCC"):
            raise NotImplementedError(kind)

    def __call__(self, class_name: str, frame_idx: int) -> np.ndarray:
        """-> (4,4) float32 initial pose of frame `frame_idx` of sequence `class_name` (data/linemod_dataset.py:330-367):
        PVNet poses of BOTH PVNet kinds are converted from PVNet's (Blender) object frame to BOP's; PVNET_LINEMOD falls back to
        the PoseCNN pose when a frame is missing, PVNET_LINEMOD_OCC re-raises (:346-354); the rotation block is then projected
        onto the nearest rotation, R (R^T R)^(-1/2) (:367)."""
        if self.kind == "POSECNN_LINEMOD":
            RT = se3_q2m(self.posecnn[class_name][frame_idx]["pose"])
        else:
            try:
                RT = np.array(self.pvnet[class_name][frame_idx], np.float64)[:3, :4].copy()
                C = np.asarray(self.blender2bop[class_name], np.float64)                     # PVNet's object frame -> BOP's (:333-335, :349-351)
                RT[:3, :3] = RT[:3, :3] @ C[:3, :3].T
                RT[:3, 3:] = -RT[:3, :3] @ C[:3, 3:] + RT[:3, 3:]
            except Exception:
                if self.kind == "PVNET_LINEMOD_OCC" or self.posecnn is None:
                    raise                                                                      # (:352-354: no fallback for LM-O)
                RT = se3_q2m(self.posecnn[class_name][frame_idx]["pose"])                      # the reference's fallback (:337-338)
        RT = np.asarray(RT, np.float64)
        RT[:3, :3] = nearest_rotation(RT[:3, :3])
        return to44(RT)


def nearest_rotation(R) -> np.ndarray:
    """R (R^T R)^(-1/2): the orthogonal polar factor of R, as data/linemod_dataset.py:367 computes it with scipy.linalg.sqrtm
    (here through the symmetric eigendecomposition of R^T R: the same matrix for a non-singular R)."""
    R = np.asarray(R, np.float64)
    w, V = np.linalg.eigh(R.T @ R)
    return R @ (V * (1.0 / np.sqrt(w))) @ V.T


def patch_crop_window(bbox_xywh, K_old, margin_ratio=0.2, output_size=128, offset_ratio=(0.0, 0.0)):
    """The window and intrinsics update of data/preprocess.py:181-255 (`patch_crop`), without the image resampling:
    bbox (x, y, w, h) of the mask -> (x0, y0, L, K_new) with L = int(max(w, h) (1 + 2 margin)), the window's top-left corner
    clamped at 0 and K_new = K scaled by output_size / L about that corner."""
    _x, _y, _w, _h = (float(v) for v in bbox_xywh)
    cx, cy = _x + _w / 2 + offset_ratio[1] * _w, _y + _h / 2 + offset_ratio[0] * _h
    L = int(max(_w, _h) * (1 + 2 * margin_ratio))
    if L <= 0:
        L = 128
    x0, y0 = max(0, int(cx - L / 2)), max(0, int(cy - L / 2))
    scale = output_size / L
    K_old = np.asarray(K_old)
    K_new = np.zeros_like(K_old, dtype=np.float64)
    K_new[0, 2], K_new[1, 2] = (K_old[0, 2] - x0) * scale, (K_old[1, 2] - y0) * scale
    K_new[0, 0], K_new[1, 1], K_new[2, 2] = K_old[0, 0] * scale, K_old[1, 1] * scale, 1
    return x0, y0, L, K_new.astype(K_old.dtype)
