"""Format-level readers (rnnpose_amd/data_io.py) on files written HERE in the reference's on-disk formats
(data/linemod_dataset.py:143-196,296-343; data/preprocess.py:181-255) -- EXPDATA itself is not part of the build."""
import pickle

import numpy as np
import pytest

from rnnpose_amd import data_io as io


def _frames(n, cls, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        R, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        R *= np.sign(np.linalg.det(R))
        out.append({"rgb_observed_path": f"data/observed/{cls}/{i:06d}-color.png", "depth_gt_observed_path": f"data/gt_observed/{cls}/{i:06d}-depth.png",
                    "K": io.LINEMOD_K.astype(np.float64), "gt_pose": np.concatenate([R, rng.normal(size=(3, 1))], 1)})
    return out


def test_info_pickles_and_flat_index(tmp_path):
    a = {"cat": _frames(5, "cat", 0), "ape": _frames(3, "ape", 1)}
    b = {"glue": _frames(4, "glue", 2)}
    pa, pb = tmp_path / "linemod_orig_test.info", tmp_path / "lmo_test.info"
    pickle.dump(a, open(pa, "wb"))
    pickle.dump(b, open(pb, "wb"))
    infos = io.load_info([pa, pb])
    assert infos["seqs"] == ["cat", "ape", "glue"] and infos["seq_lengths"] == [5, 3, 4] and infos["dataset_idx"] == [0, 0, 1]
    assert io.dataset_len(infos) == 12
    r = io.frame_record(infos, 6, root_paths=("/d0", "/d1"))        # 5 cat frames, then ape[1]
    assert (r["class_name"], r["frame_idx"]) == ("ape", 1) and r["rgb_path"] == "/d0/data/observed/ape/000001-color.png"
    assert r["pose_gt"].shape == (4, 4) and np.allclose(r["pose_gt"][:3], a["ape"][1]["gt_pose"], atol=1e-6) and r["pose_gt"][3, 3] == 1
    r = io.frame_record(infos, 11, root_paths=("/d0", "/d1"))
    assert (r["class_name"], r["frame_idx"]) == ("glue", 3) and r["rgb_path"].startswith("/d1/")
    with pytest.raises(IndexError):
        io.frame_record(infos, 12)
    only = io.load_info([pa], seq_names=["ape"])
    assert only["seqs"] == ["ape"] and io.dataset_len(only) == 3


def test_init_pose_files(tmp_path):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    q = rng.normal(size=(4, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(4, 3))
    posecnn = {"cat": [{"pose": np.concatenate([q[i], t[i]])} for i in range(4)]}
    pickle.dump(posecnn, open(tmp_path / "linemod_posecnn_results.pkl", "wb"))
    ip = io.InitPoses("POSECNN_LINEMOD", posecnn_pkl=tmp_path / "linemod_posecnn_results.pkl")
    for i in range(4):
        T = ip("cat", i)
        want = Rotation.from_quat([q[i, 1], q[i, 2], q[i, 3], q[i, 0]]).as_matrix()      # scipy is (x, y, z, w)
        assert np.allclose(T[:3, :3], want, atol=1e-6) and np.allclose(T[:3, 3], t[i], atol=1e-6) and np.allclose(T[3], [0, 0, 0, 1])
    assert np.allclose(io.quat2mat([0, 0, 0, 0]), np.eye(3))
    # PVNet poses + the blender -> BOP object-frame conversion (data/linemod_dataset.py:333-335)
    pv = {"cat": [np.concatenate([Rotation.random(random_state=i).as_matrix(), rng.normal(size=(3, 1))], 1) for i in range(3)]}
    conv = {"cat": np.concatenate([Rotation.random(random_state=9).as_matrix(), rng.normal(size=(3, 1)) * 0.01], 1)}
    np.save(tmp_path / "pvnet_linemod_test.npy", pv, allow_pickle=True)
    np.save(tmp_path / "blender2bop_RT.npy", conv, allow_pickle=True)
    ip = io.InitPoses("PVNET_LINEMOD", posecnn_pkl=tmp_path / "linemod_posecnn_results.pkl", pvnet_npy=tmp_path / "pvnet_linemod_test.npy",
                      blender2bop_npy=tmp_path / "blender2bop_RT.npy")
    T = ip("cat", 1)
    R = pv["cat"][1][:, :3] @ conv["cat"][:, :3].T
    assert np.allclose(T[:3, :3], R, atol=1e-6) and np.allclose(T[:3, 3:], -R @ conv["cat"][:, 3:] + pv["cat"][1][:, 3:], atol=1e-6)
    assert np.allclose(ip("cat", 3), io.to44(io.se3_q2m(posecnn["cat"][3]["pose"])), atol=1e-6)       # out of PVNet's range: PoseCNN fallback
    # LM-O (data/linemod_dataset.py:346-354): the SAME frame conversion, and a missing frame is an error, not a fallback
    ipo = io.InitPoses("PVNET_LINEMOD_OCC", posecnn_pkl=tmp_path / "linemod_posecnn_results.pkl", pvnet_npy=tmp_path / "pvnet_linemod_test.npy",
                       blender2bop_npy=tmp_path / "blender2bop_RT.npy")
    assert np.allclose(ipo("cat", 1), T, atol=1e-7)
    with pytest.raises((IndexError, KeyError)):
        ipo("cat", 3)
    with pytest.raises(ValueError):
        io.InitPoses("PVNET_LINEMOD_OCC", pvnet_npy=tmp_path / "pvnet_linemod_test.npy")        # the frame file is required
    # the rotation block is regularised to the nearest rotation, R (R^T R)^(-1/2) (:367), as scipy.linalg.sqrtm gives it
    import scipy.linalg
    Rn = Rotation.random(random_state=5).as_matrix() + rng.normal(size=(3, 3)) * 0.02
    want = Rn @ np.linalg.inv(scipy.linalg.sqrtm(Rn.T @ Rn))
    got = io.nearest_rotation(Rn)
    assert np.allclose(got, np.real(want), atol=1e-10) and np.allclose(got.T @ got, np.eye(3), atol=1e-12)
    pv2 = {"cat": [np.concatenate([Rn, rng.normal(size=(3, 1))], 1)]}
    np.save(tmp_path / "pvnet_noisy.npy", pv2, allow_pickle=True)
    Tn = io.InitPoses("PVNET_LINEMOD_OCC", pvnet_npy=tmp_path / "pvnet_noisy.npy", blender2bop_npy=tmp_path / "blender2bop_RT.npy")("cat", 0)
    assert np.allclose(Tn[:3, :3].T @ Tn[:3, :3], np.eye(3), atol=1e-6)
    assert np.allclose(Tn[:3, :3], io.nearest_rotation(Rn @ conv["cat"][:, :3].T), atol=1e-6)


def test_patch_crop_window_matches_the_reference_arithmetic():
    K = io.LINEMOD_K
    x0, y0, L, Kn = io.patch_crop_window((300, 200, 80, 50), K, margin_ratio=0.2, output_size=128)
    assert L == int(80 * 1.4) and (x0, y0) == (int(340 - L / 2), int(225 - L / 2))
    s = 128 / L
    assert np.allclose(Kn, [[K[0, 0] * s, 0, (K[0, 2] - x0) * s], [0, K[1, 1] * s, (K[1, 2] - y0) * s], [0, 0, 1]])
    x0, y0, L, _ = io.patch_crop_window((2, 1, 40, 60), K)                                 # window clamped at the image corner
    assert (x0, y0) == (0, 0) and L == int(60 * 1.4)
    assert abs(io.diameter_m("cat") - 0.152633) < 1e-9 and len(io.LINEMOD_CLASSES) == 13


def test_tckpt_subtree_selection(tmp_path):
    """`.tckpt` = torch.save(state_dict) (torchplus/train/checkpoint.py:92); include / exclude / shape rule of tools/eval.py:386-413."""
    import torch
    from rnnpose_amd.render_adapter import filter_param_dict
    sd = {"motion_net.sigma.0": torch.ones(1), "motion_net.cf_net.update_block.gru.convz1.weight": torch.zeros(128, 384, 1, 5),
          "descriptor_net.conv.weight": torch.zeros(3, 3), "global_step": torch.zeros(1)}
    torch.save(sd, tmp_path / "voxelnet-100.tckpt")
    got = torch.load(tmp_path / "voxelnet-100.tckpt", map_location="cpu")
    kept = filter_param_dict(got, include="motion_net", exclude=".*sigma")
    assert sorted(kept) == ["motion_net.cf_net.update_block.gru.convz1.weight"]
