"""Closed-form synthetic inputs for the refinement hot path.

Every tensor is a pure function of (name, seed, shape): a 32-bit integer hash of the flat element
index, mapped to fp32 without any transcendental function, so this container, the GPU box and the
golden-vector generator (tests/golden/gen_golden.py) all produce bit-identical inputs and only the
*outputs* of the reference need to be stored as fixtures.

Shapes and value ranges follow what the reference feeds the loop:
  * fmap1/fmap2: RAFT encoder outputs (B,256,H/8,W/8)           model/PoseRefiner.py:311
  * ctx:         rendered 3-D context features x0.1 (B,256,H,W)  model/PoseRefiner.py:283
  * g1/g2:       L2-normalised 32-ch descriptors (B,32,H,W)      model/PoseRefiner.py:342-345
  * depth:       rendered depth, 0 = background (B,1,H,W)        model/PoseRefiner.py:313
  * K:           LINEMOD focal lengths, principal point = crop centre  data/linemod/linemod_config.py:23-25
"""
from __future__ import annotations

import zlib

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """murmur3 fmix32 on uint64 arrays holding 32-bit values (exact integer arithmetic)."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _stream_key(name: str, seed: int) -> np.uint64:
    return np.uint64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B9)) & 0xFFFFFFFF)


def uniform(name: str, shape, seed: int = 0, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """U[lo,hi) fp32, 24 random mantissa bits per element."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    h = _mix32(idx * np.uint64(0x9E3779B1) + _stream_key(name, seed))
    h = _mix32(h ^ (idx >> np.uint64(32)) ^ np.uint64(0x68E31DA4))
    u = (h >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(name: str, shape, seed: int = 0, std: float = 1.0) -> np.ndarray:
    """Approximately N(0,std^2): Irwin-Hall sum of 4 uniforms (variance 1/3), no transcendentals."""
    acc = np.zeros(int(np.prod(shape)), dtype=np.float64)
    for k in range(4):
        acc += uniform(f"{name}#{k}", (acc.size,), seed).astype(np.float64)
    return ((acc - 2.0) * (np.sqrt(3.0) * std)).astype(np.float32).reshape(shape)


def intrinsics(B: int, H: int, W: int) -> np.ndarray:
    K = np.zeros((B, 3, 3), dtype=np.float32)
    K[:, 0, 0] = 572.4114
    K[:, 1, 1] = 573.57043
    K[:, 0, 2] = W / 2.0
    K[:, 1, 2] = H / 2.0
    K[:, 2, 2] = 1.0
    return K


def se3_exp_np(xi: np.ndarray) -> np.ndarray:
    """fp64 closed-form SE(3) exponential (used only to build perturbed start poses)."""
    xi = np.asarray(xi, dtype=np.float64).reshape(-1, 6)
    out = np.tile(np.eye(4), (xi.shape[0], 1, 1))
    for n, (v, w) in enumerate(zip(xi[:, :3], xi[:, 3:])):
        th = np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if th < 1e-8:
            A, Bc, C = 1.0, 0.5, 1.0 / 6.0
        else:
            A, Bc, C = np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3
        R = np.eye(3) + A * Wx + Bc * Wx @ Wx
        V = np.eye(3) + Bc * Wx + C * Wx @ Wx
        out[n, :3, :3] = R
        out[n, :3, 3] = V @ v
    return out


def make_inputs(B: int, H: int, W: int, seed: int = 0, C: int = 256, D: int = 32,
                pose_sigma: float = 0.02, with_images: bool = False) -> dict:
    """Synthetic per-batch inputs of one refinement (numpy, fp32). See module docstring."""
    h, w = H // 8, W // 8
    out = {
        "fmap1": normal("fmap1", (B, C, h, w), seed),
        "fmap2": normal("fmap2", (B, C, h, w), seed),
        "ctx": normal("ctx", (B, 256, H, W), seed, std=0.1),
        "K": intrinsics(B, H, W),
        "sigma": np.ones((1,), dtype=np.float32),
    }
    for nm in ("g1", "g2"):
        g = normal(nm, (B, D, H, W), seed).astype(np.float64)
        g /= np.sqrt((g * g).sum(axis=1, keepdims=True)) + 1e-12
        out[nm] = g.astype(np.float32)
    depth = uniform("depth", (B, 1, H, W), seed, 0.9, 1.2)
    depth[:, :, : H // 4] = 0.0  # background band
    out["depth"] = depth
    xi = normal("xi0", (B, 6), seed, std=pose_sigma)
    out["G0"] = se3_exp_np(xi).astype(np.float32).reshape(B, 1, 4, 4)
    if with_images:
        out["img_render"] = uniform("img_render", (B, 3, H, W), seed)
        out["img_target"] = uniform("img_target", (B, 3, H, W), seed)
    return out


def make_module_weights(shapes: dict, seed: int = 0, gain: float = 1.0) -> dict:
    """Hash-generated weights for a {name: shape} dict, kaiming-like std = gain*sqrt(2/fan_in);
    biases (1-D) are small uniform.  Used for the update block and the encoder in tests/bench."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if len(shape) == 1:
            out[name] = uniform("w:" + name, shape, seed, -0.05, 0.05)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = normal("w:" + name, shape, seed, std=gain * float(np.sqrt(2.0 / fan_in)))
    return out


# ---- the same generator on torch tensors (any device) -------------------------------------------------------------
# Bit-identical to the numpy functions above: exact 64-bit integer arithmetic (two's-complement wrap == the uint64 wrap in
# the low 32 bits that survive the mask) and IEEE fp64 -> fp32 conversions.  Used for config-5-sized test inputs, which the
# single-threaded numpy path needs minutes for; tests/test_host_logic.py checks the two against each other.
_CHUNK = 1 << 25


def _mix32_t(x):
    x = x & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x


def _uniform01_t(name, n0, n1, seed, device):
    import torch
    idx = torch.arange(n0, n1, dtype=torch.int64, device=device)
    h = _mix32_t(idx * 0x9E3779B1 + int(_stream_key(name, seed)))
    h = _mix32_t(h ^ (idx >> 32) ^ 0x68E31DA4)
    return (h >> 8).to(torch.float64) * (1.0 / 16777216.0)


def uniform_t(name, shape, seed=0, lo=0.0, hi=1.0, device="cpu"):
    import torch
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.float32, device=device)
    for n0 in range(0, n, _CHUNK):
        n1 = min(n, n0 + _CHUNK)
        out[n0:n1] = (lo + (hi - lo) * _uniform01_t(name, n0, n1, seed, device)).to(torch.float32)
    return out.reshape(shape)


def normal_t(name, shape, seed=0, std=1.0, device="cpu"):
    import torch
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.float32, device=device)
    scale = float(np.sqrt(3.0) * std)
    for n0 in range(0, n, _CHUNK):
        n1 = min(n, n0 + _CHUNK)
        acc = torch.zeros(n1 - n0, dtype=torch.float64, device=device)
        for k in range(4):
            # (the numpy path rounds every uniform to fp32 before summing: 0.0 + 1.0 * u in fp64, then fp32)
            acc += (0.0 + 1.0 * _uniform01_t(f"{name}#{k}", n0, n1, seed, device)).to(torch.float32).to(torch.float64)
        out[n0:n1] = ((acc - 2.0) * scale).to(torch.float32)
    return out.reshape(shape)


def make_inputs_t(B, H, W, seed=0, C=256, D=32, pose_sigma=0.02, device="cpu", with_images=False):
    """make_inputs() as torch tensors generated on `device` (bit-identical values)."""
    import torch
    h, w = H // 8, W // 8
    out = {
        "fmap1": normal_t("fmap1", (B, C, h, w), seed, device=device),
        "fmap2": normal_t("fmap2", (B, C, h, w), seed, device=device),
        "ctx": normal_t("ctx", (B, 256, H, W), seed, std=0.1, device=device),
        "K": torch.from_numpy(intrinsics(B, H, W)).to(device),
        "sigma": torch.ones(1, dtype=torch.float32, device=device),
    }
    for nm in ("g1", "g2"):
        g = normal_t(nm, (B, D, H, W), seed, device=device).to(torch.float64)
        g /= torch.sqrt((g * g).sum(dim=1, keepdim=True)) + 1e-12
        out[nm] = g.to(torch.float32)
    depth = uniform_t("depth", (B, 1, H, W), seed, 0.9, 1.2, device=device)
    depth[:, :, : H // 4] = 0.0
    out["depth"] = depth
    xi = normal("xi0", (B, 6), seed, std=pose_sigma)
    out["G0"] = torch.from_numpy(se3_exp_np(xi).astype(np.float32).reshape(B, 1, 4, 4)).to(device)
    if with_images:
        out["img_render"] = uniform_t("img_render", (B, 3, H, W), seed, device=device)
        out["img_target"] = uniform_t("img_target", (B, 3, H, W), seed, device=device)
    return out
