"""The per-pixel HIP kernels of csrc/pointwise.hip EXECUTED ON THE HOST (tests/host_exec/harness.hpp), against the CPU oracle and the
reference-generated golden vectors -- in the `-m "not gpu"` tier.

The source file is compiled as plain C++ and its C-ABI entry points are called through the product's own ctypes prototypes
(rnnpose_amd/_lib.py) on numpy arrays: argument checks, launch geometry and kernel code are the shipped ones; one OS thread stands in
for each GPU thread of a workgroup, `__shared__` is a static, `__syncthreads()` a barrier.  This is test infrastructure: nothing here
is reachable from the product path (ops.py refuses CPU tensors; the library itself has no host build).  Kernels built on wave-level
hardware operations (MFMA convolutions, volume build, LM reduction) cannot run this way and are covered by the `-m gpu` tests only.

What it adds to the GPU parity tests: the a5 / a6 / a7 / a8 arithmetic of the product SOURCE is pinned to the reference's vectors
before a GPU box is involved -- geometry bit for bit (same fp32 operation order as geometry/projective_ops.py:68-114, no contraction),
the rest to the ulp-level difference between the host's and the device's expf / tanhf."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import rnnpose_oracle as orc
from rnnpose_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    from rnnpose_amd import _lib, build
    clang = os.path.join(os.path.dirname(os.path.realpath(build.hipcc())), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
    out = str(tmp_path_factory.mktemp("host_exec") / "pointwise_host.so")
    cmd = [clang, "-x", "c++", "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "rnnpose_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "host_exec"),
           os.path.join(ROOT, "tests", "host_exec", "pointwise_host.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(out)
    lib.host_last_error.restype = C.c_char_p
    for name in ("rnnpose_context_prep_f32", "rnnpose_flow_to_coords_f32", "rnnpose_convex_upsample_f32", "rnnpose_induced_flow_f32",
                 "rnnpose_induced_coords_lowres_f32", "rnnpose_corr_weight_f32", "rnnpose_gru_gate_f32", "rnnpose_gru_update_f32"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.PROTOTYPES[name]          # the product's own prototypes

    def call(name, *args):
        rc = getattr(lib, name)(*args)
        assert rc == 0, (name, lib.host_last_error())
    lib.call = call
    return lib


def A(x):
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def test_induced_flow_is_the_reference_geometry_bit_for_bit(host):
    """a7 (geometry/transformation.py:184-198, projective_ops.py:68-114, PoseRefiner.py:324-328): flow and validity mask of the shipped
    kernel == the oracle's fp32 statements, exactly, and within the oracle test's own tolerance of the reference's output (golden)."""
    d = syn.make_inputs(2, 64, 80, seed=5)
    depth, K, G = A(d["depth"]), A(d["K"]), A(d["G0"])
    B, _, H, W = depth.shape
    flow, vm = np.full((B, 2, H, W), 7.0, np.float32), np.full((B, H, W), 7.0, np.float32)
    host.call("rnnpose_induced_flow_f32", P(depth), P(K), P(G), B, H, W, 1e-5, 0, P(flow), P(vm), None)
    want, wvm = orc.induced_flow(depth, K, G)
    assert np.array_equal(flow, want.numpy()) and np.array_equal(vm, wvm.numpy())
    g = np.load(os.path.join(GOLD, "geometry.npz"))                 # the reference's own output (inputs: tests/golden/gen_golden.py)
    depth, K, G = A(g["depth"]), A(syn.make_inputs(2, 64, 96, seed=7, pose_sigma=0.03)["K"]), A(g["G"])
    B, _, H, W = depth.shape
    flow, vm = np.zeros((B, 2, H, W), np.float32), np.zeros((B, H, W), np.float32)
    host.call("rnnpose_induced_flow_f32", P(depth), P(K), P(G), B, H, W, 1e-5, 0, P(flow), P(vm), None)
    ref = g["flow_init"][:, 0].astype(np.float64)
    assert float(np.max(np.abs(flow - ref) - 1e-4 - 1e-6 * np.abs(ref))) <= 0        # (the oracle test's own mixed tolerance)
    assert np.array_equal(vm, g["vmask"].reshape(vm.shape).astype(np.float32))
    # argument validation is the shipped one
    assert host.rnnpose_induced_flow_f32(P(depth), P(K), P(G), B, H, W, 1e-5, 2, P(flow), P(vm), None) == 1
    assert b"mode must be 0 or 1" in host.host_last_error()


def test_induced_coords_lowres_is_flow_init_resized(host):
    """a7 + a5 fused (CFNet.py:136-144): coords1 = grid + resize_ac(flow_init / 8) evaluated at the four taps only == the two-step
    path of the same library (induced_flow -> flow_to_coords) bit for bit, and the oracle."""
    d = syn.make_inputs(2, 64, 80, seed=6)
    depth, K, G = A(d["depth"]), A(d["K"]), A(d["G0"])
    B, _, H, W = depth.shape
    h, w = H // 8, W // 8
    c1 = np.zeros((B, 2, h, w), np.float32)
    host.call("rnnpose_induced_coords_lowres_f32", P(depth), P(K), P(G), B, H, W, h, w, 1e-5, P(c1), None)
    flow = np.zeros((B, 2, H, W), np.float32)
    host.call("rnnpose_induced_flow_f32", P(depth), P(K), P(G), B, H, W, 1e-5, 0, P(flow), None, None)
    c2 = np.zeros_like(c1)
    host.call("rnnpose_flow_to_coords_f32", P(flow), B, H, W, h, w, P(c2), None)
    assert np.array_equal(c1, c2)
    want = orc.flow_init_to_coords1(orc.induced_flow(depth, K, G)[0]).numpy()
    assert np.abs(c1 - want).max() <= 1e-5


def test_corr_weight_vs_oracle_and_reference_vector(host):
    """a8 (PoseRefiner.py:342-345): exp(-|1 - <g1, bilinear(g2, target)>| / sigma) * [depth > 0], the align_corners mismatch of the
    reference included; both target forms of the entry point (absolute coordinates; planar flow + pixel grid)."""
    d = syn.make_inputs(2, 40, 56, seed=7)
    g1, g2, depth = A(d["g1"]), A(d["g2"]), A(d["depth"])
    B, D, H, W = g1.shape
    flow = A(syn.uniform("hostexec.flow", (B, 2, H, W), 7, -6.0, 6.0))
    flow[0, :, :3] = 200.0                          # taps far outside the map: zero padding
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    target = A(np.stack([flow[:, 0] + xs, flow[:, 1] + ys], -1))
    sigma = A([0.7])
    want = orc.corr_weight(g1, g2, target, depth, sigma).numpy()
    for mode, t in ((0, target), (1, flow)):
        wt = np.full((B, H, W), 7.0, np.float32)
        host.call("rnnpose_corr_weight_f32", P(g1), P(g2), P(t), mode, P(depth), P(sigma), B, D, H, W, P(wt), None)
        assert np.abs(wt - want).max() <= 2e-6, mode
        assert np.array_equal(wt == 0, want == 0)
    g = np.load(os.path.join(GOLD, "geometry.npz"))                 # the reference's own weight map for its own correspondence field
    d = syn.make_inputs(2, 64, 96, seed=7, pose_sigma=0.03)
    g1, g2, depth, t = A(d["g1"]), A(d["g2"]), A(g["depth"]), A(g["target"][:, 0])
    B, D, H, W = g1.shape
    wt = np.zeros((B, H, W), np.float32)
    host.call("rnnpose_corr_weight_f32", P(g1), P(g2), P(t), 0, P(depth), P(A(g["sigma"])), B, D, H, W, P(wt), None)
    assert np.abs(wt - g["weight"].reshape(wt.shape)).max() < 1e-4


def test_context_prep_and_convex_upsample_vs_reference_vectors(host):
    """a5 (CFNet.py:124-133: tanh / relu halves of the resized context) and a6 (CFNet.py:95-106: softmax over 9 taps of 8 x flow, the
    kernel with shared-memory staging and a workgroup barrier) on the reference-generated `upsample_ctx` vectors."""
    g = np.load(os.path.join(GOLD, "upsample_ctx.npz"))             # reference outputs; inputs regenerated as tests/test_oracle_golden.py does
    ctx = A(syn.normal("ctx", (1, 256, 64, 96), 5, std=0.1))
    B, Cc, H, W = ctx.shape
    h, w = g["net"].shape[-2:]
    net, inp = np.zeros(g["net"].shape, np.float32), np.zeros(g["inp"].shape, np.float32)
    host.call("rnnpose_context_prep_f32", P(ctx), B, Cc, H, W, h, w, g["net"].shape[1], P(net), P(inp), None)
    assert np.abs(net - g["net"]).max() < 1e-6 and np.abs(inp - g["inp"]).max() < 1e-6
    finit = A(syn.normal("finit", (2, 2, 64, 96), 5, std=4.0))
    c1 = np.zeros(g["coords1"].shape, np.float32)
    host.call("rnnpose_flow_to_coords_f32", P(finit), 2, 64, 96, 8, 12, P(c1), None)
    assert np.abs(c1 - g["coords1"]).max() < 1e-5
    flow, mask = A(syn.normal("up_flow", (2, 2, 16, 12), 5, std=3.0)), A(syn.normal("up_mask", (2, 576, 16, 12), 5, std=2.0))
    B, _, h, w = flow.shape
    up = np.zeros((B, 2, 8 * h, 8 * w), np.float32)
    host.call("rnnpose_convex_upsample_f32", P(flow), P(mask), B, h, w, 8, P(up), None)
    assert np.abs(up - g["flow_up"]).max() < 1e-4
    want = orc.convex_upsample(flow, mask).numpy()
    assert np.abs(up - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_gru_pointwise_stages(host):
    """a4 (thirdparty/raft/update.py:45-60): z = sigmoid(.), r * h written into the concatenated buffer, h' = (1 - z) h + z tanh(q)."""
    B, Cn, hw, Ctot = 2, 16, 35, 40
    zr = A(syn.normal("hostexec.zr", (B, 2 * Cn, hw), 3, std=2.0))
    hcat = A(syn.normal("hostexec.h", (B, Ctot, hw), 3))
    q = A(syn.normal("hostexec.q", (B, Cn, hw), 3, std=2.0))
    z, rhx = np.zeros((B, Cn, hw), np.float32), hcat.copy()
    host.call("rnnpose_gru_gate_f32", P(zr), P(hcat), B, Cn, Ctot, hw, P(z), P(rhx), None)
    tz, tr, th = torch.sigmoid(torch.from_numpy(zr[:, :Cn])), torch.sigmoid(torch.from_numpy(zr[:, Cn:])), torch.from_numpy(hcat[:, :Cn])
    assert np.abs(z - tz.numpy()).max() <= 1e-6 and np.abs(rhx[:, :Cn] - (tr * th).numpy()).max() <= 1e-6
    assert np.array_equal(rhx[:, Cn:], hcat[:, Cn:])
    hout = np.zeros((B, Ctot, hw), np.float32)
    host.call("rnnpose_gru_update_f32", P(z), P(q), P(hcat), B, Cn, Ctot, hw, P(hout), Ctot, None)
    want = (1 - torch.from_numpy(z)) * th + torch.from_numpy(z) * torch.tanh(torch.from_numpy(q))
    assert np.abs(hout[:, :Cn] - want.numpy()).max() <= 1e-6
