"""Host-side logic that needs no GPU: module trees / state_dict keys, synthetic generator determinism,
sharding rule, refusal to run on CPU tensors (there is no CPU product path)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import rnnpose_oracle as orc
from rnnpose_amd import synthetic as syn


def test_update_block_state_dict_matches_reference_shapes():
    from rnnpose_amd.cfnet import GRU_CFUpdator
    net = GRU_CFUpdator(dict(pretrained_model=None, mixed_precision=True, fea_net="default"))
    sd = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    want = {"update_block." + k: v for k, v in orc.UPDATE_BLOCK_SHAPES.items()}
    assert sd == want                                             # 30 tensors
    assert sum(int(np.prod(s)) for s in sd.values()) == 3_120_960


def test_encoder_and_refiner_state_dict_keys():
    from rnnpose_amd.cfnet import ImageFeaEncoder
    from rnnpose_amd.pose_refiner import PoseRefiner
    enc = ImageFeaEncoder()
    sd = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert sd == {"fnet." + k: v for k, v in orc.encoder_shapes().items()}
    ref = PoseRefiner()
    keys = list(ref.state_dict())
    assert keys[0] == "sigma.0"
    assert sum(k.startswith("image_fea_enc.fnet.") for k in keys) == len(sd)
    assert sum(k.startswith("cf_net.update_block.") for k in keys) == 30
    # reference weights load by name
    w = syn.make_module_weights(orc.UPDATE_BLOCK_SHAPES, seed=0)
    ref.cf_net.update_block.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)


def test_engine_param_key_tracks_updates():
    """The identity captured hipGraphs depend on changes on load_state_dict and on in-place parameter updates."""
    from rnnpose_amd.cfnet import GRU_CFUpdator
    net = GRU_CFUpdator(dict(pretrained_model=None))
    eng = net.engine()
    k0 = eng.param_key()
    assert eng.param_key() == k0
    net.update_block.load_state_dict({k: v.clone() for k, v in net.update_block.state_dict().items()})
    k1 = eng.param_key()
    assert k1 != k0
    with torch.no_grad():
        net.update_block.gru.convz1.bias.add_(1.0)
    assert eng.param_key() != k1


def test_motion_net_checkpoint_loader(tmp_path):
    """A reference checkpoint is torch.save(RNNPose.state_dict()) (torchplus/train/checkpoint.py:92): the `motion_net.*`
    sub-tree must load into PoseRefiner with the include / exclude / same-shape rule of tools/eval.py:386-413."""
    from rnnpose_amd.pose_refiner import PoseRefiner
    from rnnpose_amd.render_adapter import filter_param_dict, load_motion_net_checkpoint
    src = PoseRefiner()
    g = torch.Generator().manual_seed(3)
    ckpt = {"motion_net." + k: torch.randn(v.shape, generator=g) for k, v in src.state_dict().items()}
    ckpt["descriptor_net.encoder.weight"] = torch.randn(4, 4)                 # other sub-trees of RNNPose: ignored
    ckpt["global_step"] = torch.zeros(1)
    bad = "motion_net.cf_net.update_block.encoder.convc1.weight"
    good_shape = ckpt[bad].shape
    ckpt[bad] = torch.randn(7, 7)                                             # wrong shape: "Fail to load", kept at init
    path = tmp_path / "voxelnet-1000.tckpt"
    torch.save(ckpt, path)
    dst = PoseRefiner()
    before = dst.state_dict()[bad[len("motion_net."):]].clone()
    loaded, skipped = load_motion_net_checkpoint(dst, str(path))
    assert skipped == [bad] and len(loaded) == len(src.state_dict()) - 1
    sd = dst.state_dict()
    for k in loaded:
        assert torch.equal(sd[k], ckpt["motion_net." + k]), k
    assert torch.equal(sd[bad[len("motion_net."):]], before) and before.shape == good_shape
    # include / exclude are regular expressions matched at the start of the FULL key (re.match, eval.py:110-127)
    only = filter_param_dict(ckpt, include="motion_net\\.cf_net", exclude=".*bias")
    assert only and all(k.startswith("motion_net.cf_net") and not k.endswith("bias") for k in only)
    dst2 = PoseRefiner()
    l2, _ = load_motion_net_checkpoint(dst2, ckpt, include="motion_net\\.image_fea_enc")
    assert l2 and all(k.startswith("image_fea_enc.") for k in l2)
    with pytest.raises(RuntimeError):
        load_motion_net_checkpoint(PoseRefiner(), ckpt, strict=True)


def test_synthetic_inputs_are_bit_reproducible():
    d = syn.make_inputs(2, 32, 48, seed=5)
    h = hashlib.sha256()
    for k in sorted(d):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    assert h.hexdigest() == syn_digest()
    assert d["depth"][:, :, :8].max() == 0 and d["depth"][:, :, 8:].min() >= 0.9
    n = np.sqrt((d["g1"].astype(np.float64) ** 2).sum(1))
    assert np.allclose(n, 1.0, atol=1e-6)
    u = syn.uniform("u", (100000,), 1)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3
    g = syn.normal("n", (100000,), 1)
    assert abs(g.mean()) < 1e-2 and abs(g.std() - 1.0) < 1e-2


def syn_digest():
    # recorded from the build container; any change of the generator invalidates tests/golden/*.npz
    return "0baa51015be23deba3ab0ca1b76b004dced002a7442ceb5c7affcb82cb54795b"


def test_torch_generator_is_bit_identical_to_numpy_generator():
    """make_inputs_t (used for config-5-sized GPU tests) == make_inputs, element for element."""
    a = syn.make_inputs(2, 64, 96, seed=7, with_images=True)
    b = syn.make_inputs_t(2, 64, 96, seed=7, with_images=True)
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k].numpy()), k
    assert np.array_equal(syn.uniform("u", (5, 7), 3, -2.0, 3.0), syn.uniform_t("u", (5, 7), 3, -2.0, 3.0).numpy())


def test_ops_refuse_cpu_tensors():
    from rnnpose_amd import ops
    x = torch.zeros(1, 256, 16, 16)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.corr_pyramid(x, x)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.lm_solve_update(torch.eye(6, dtype=torch.float64)[None], torch.zeros(1, 6, dtype=torch.float64),
                            torch.eye(4)[None])
    from rnnpose_amd.update import BasicUpdateBlock
    from rnnpose_amd.cfnet import AttrDict
    blk = BasicUpdateBlock(AttrDict(corr_levels=4, corr_radius=4))
    with pytest.raises(RuntimeError, match="GPU only"):
        blk(torch.zeros(1, 128, 16, 16), torch.zeros(1, 128, 16, 16), torch.zeros(1, 324, 16, 16),
            torch.zeros(1, 2, 16, 16))


def test_shard_indices_match_reference_rule():
    from rnnpose_amd.distributed import shard_indices
    # utils/distributed_utils.py:150-169: rank-strided, wrap-around padded
    for n, world in ((10, 4), (8, 4), (1, 2), (13, 8)):
        seen = []
        for r in range(world):
            idx, uniq = shard_indices(n, r, world)
            total = -(-n // world) * world
            ref = (list(range(n)) + list(range(total - n)))[r:total:world]
            assert idx == ref
            seen += [i for i, u in zip(idx, uniq) if u]
        assert sorted(seen) == list(range(n))                   # duplicates are masked exactly once each


def test_metric_accumulator_single_process():
    from rnnpose_amd.distributed import MetricAccumulator
    acc = MetricAccumulator(("cat", "ape"))
    acc.update("cat", dict(add=1.0, proj2d=1.0))
    acc.update("cat", dict(add=0.0, proj2d=1.0))
    acc.update("cat", dict(add=1.0), unique=False)              # padded duplicate: ignored
    out = acc.reduce()
    assert out["cat"]["n"] == 2 and out["cat"]["add"] == 0.5 and out["cat"]["proj2d"] == 1.0
    assert out["ape"]["n"] == 0 and np.isnan(out["ape"]["add"])


def test_concurrent_build_calls_are_serialised(tmp_path):
    """One process per GPU: every rank calls build.build() at start-up.  Two processes forcing / checking a build at the same time
    must both succeed and leave a loadable library (the file lock makes the loser wait and re-check the stamp).  Run on a scratch
    library directory with two small sources (the locking is what is tested; a forced rebuild of all 17 files took this test 70 s
    and re-wrote the in-tree library under the other tests)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r); from rnnpose_amd import build; d = %r; "
            "build.LIBDIR = d; build.LIB = os.path.join(d, 'librnnpose_hip.so'); build.STAMP = os.path.join(d, 'librnnpose_hip.stamp'); "
            "build.sources = lambda: [os.path.join(build.CSRC, f) for f in ('runtime.hip', 'raster.hip')]; "
            "p = build.build(force=(sys.argv[1] == 'force')); print(p)") % (root, str(tmp_path / "lib"))
    procs = [subprocess.Popen([sys.executable, "-c", code, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for mode in ("force", "check", "check")]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        assert out.strip().endswith("librnnpose_hip.so")
    assert os.path.exists(str(tmp_path / "lib" / "librnnpose_hip.stamp"))
    import ctypes
    lib = ctypes.CDLL(outs[0][0].strip())
    assert lib.rnnpose_abi_version() >= 3


def test_torch_library_ops_registered_with_fake_kernels():
    """the operator set is visible as torch.ops.rnnpose.* with the listed schemas; fake kernels give shapes
    and dtypes under FakeTensorMode (no GPU touched); a CPU tensor fails loudly in the dispatcher (no CPU path)."""
    import rnnpose_amd.torch_ops as to
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name, schema in to._SCHEMAS.items():
        op = getattr(torch.ops.rnnpose, name).default
        assert str(op._schema).startswith("rnnpose::" + name + "(Tensor") and len(op._schema.arguments) == schema.split(" -> ")[0].count(",") + 1
    with FakeTensorMode():
        dev = "cuda"
        f = torch.empty(2, 256, 60, 80, device=dev)
        pyr = torch.ops.rnnpose.corr_pyramid(f, f, 4)
        assert [tuple(p.shape) for p in pyr] == [(9600, 1, 60, 80), (9600, 1, 30, 40), (9600, 1, 15, 20), (9600, 1, 7, 10)]
        c = torch.empty(2, 2, 60, 80, device=dev)
        assert torch.ops.rnnpose.corr_lookup(pyr, c, 4).shape == (2, 324, 60, 80)
        assert torch.ops.rnnpose.convex_upsample(c, torch.empty(2, 576, 60, 80, device=dev), 8).shape == (2, 2, 480, 640)
        d = torch.empty(2, 1, 480, 640, device=dev)
        K, G = torch.empty(2, 3, 3, device=dev), torch.empty(2, 1, 4, 4, device=dev)
        fl, vm = torch.ops.rnnpose.induced_flow(d, K, G, 1e-5)
        assert fl.shape == (2, 2, 480, 640) and vm.shape == (2, 480, 640)
        g = torch.empty(2, 32, 480, 640, device=dev)
        w = torch.ops.rnnpose.corr_weight(g, g, fl, d, torch.empty(1, device=dev))
        H, b = torch.ops.rnnpose.lm_normal_eq(fl, w, d, K, G)
        assert w.shape == (2, 480, 640) and H.shape == (2, 6, 6) and H.dtype == torch.float64 and b.shape == (2, 6)
        Gn, xi = torch.ops.rnnpose.lm_solve_update(H, b, G)
        assert Gn.shape == G.shape and xi.shape == (2, 6) and xi.dtype == torch.float32
    with pytest.raises(NotImplementedError):
        torch.ops.rnnpose.convex_upsample(torch.zeros(1, 2, 4, 4), torch.zeros(1, 576, 4, 4))


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE launches N ranks itself (the driver's invocation measured
    one GPU): the command is the torchrun contract of the task (one node, 127.0.0.1 rendezvous), arguments passed through."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4)
    assert "only 1 GPU" in str(e.value.code)


def test_traffic_record_belongs_to_the_kernel_sources_in_the_tree():
    """bench.py quotes profiles/traffic.json (HBM counters per launch) only when the record carries the digest of the kernel sources
    in the tree, and prints `traffic: null` otherwise.  The record is either measured on these sources or carried over from the commit
    it was measured on by tools/isa_equal.py (byte-identical ISA of every measured kernel + identical tiling of every headline launch):
    an edit of csrc/ without either makes this fail here, not silently blank the roofline line on the driver's box."""
    import json
    from rnnpose_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert t["csrc_digest"] == build.source_digest()
    for k in ("conv", "corr_pyramid_h3"):
        assert t[k]["bytes_per_launch"] > 0 and t[k]["launches"] > 0
    co = t.get("carried_over")
    if co is not None:
        assert co["measured_on_digest"] != t["csrc_digest"] and "isa_equal" in co["check"] and co["measured_at"]


def test_division_by_255_by_correction_step_is_the_correctly_rounded_quotient():
    """csrc/stem.hip forms x / 255 (model/CFNet.py:42) as q = x * y, r = fma(-255, q, x), q' = fma(r, y, q) with y = fl(1 / 255).  The
    exhaustive check over all finite non-negative floats is tools/probes/div255_markstein.c (0 differences, profiles/r06_stem.txt);
    this samples every 257th float and all integers / half-integers an 8-bit image can hold."""
    import numpy as np
    bits = np.arange(0, 0x7F800000, 257, dtype=np.uint32)
    x = np.concatenate([bits.view(np.float32), np.arange(0, 256.5, 0.5, dtype=np.float32)])
    y = np.float32(1.0) / np.float32(255.0)
    xd, yd = x.astype(np.float64), np.float64(y)
    q = (xd * yd).astype(np.float32)                                   # one rounding: the product of two floats is exact in fp64
    r = (xd - 255.0 * q.astype(np.float64)).astype(np.float32)         # fma(-255, q, x): exact difference, one rounding
    q1 = (r.astype(np.float64) * yd + q.astype(np.float64)).astype(np.float32)   # fma(r, y, q): |r y| << |q|, the fp64 sum rounds once more -- double
    want = x / np.float32(255.0)                                       # rounding can differ from a true fma only in 2^-29 of the cases: none here
    assert np.array_equal(q1, want)
