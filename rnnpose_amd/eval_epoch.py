"""One evaluation epoch: what tools/eval.py does between `mp.spawn` and the printed per-class table.

    reference                                                         here
    tools/eval.py:224-225   one process per GPU (mp.spawn)            torchrun / bench-style spawn; rank, world from the env
    tools/eval.py:305-316   init_process_group("nccl")                distributed.init_from_env (RCCL; gloo on CPU)
    tools/eval.py:471       DistributedSequatialSampler               distributed.shard_indices (rank-strided, wrap-around
                            (utils/distributed_utils.py:150-169)      duplicates flagged so that they can be masked)
    model/RNNPose.py:157-222  one batch = ONE object class: views     batches of one class -> PoseRefiner(image, Ts, K, fea_3d,
                            from the class model, PoseRefiner(...)    Tj_gt, obj_cls, geofea_3d, geofea_2d) through its renderer
    utils/eval_metric.py:306-356  LineMODEvaluator.evaluate per       evaluator.LineMODEvaluator (csrc/eval_metrics.hip), batched
                            sample (ADD / ADD-S / proj2d / 5cm5deg)
    tools/train.py:725-741  two scalar all_gathers per metric         ONE all_reduce(SUM) of a packed fp64 vector at the end
    (tools/eval.py itself prints per-rank means, :560-562)            of the epoch (MetricAccumulator.reduce)

Datasets (`EXPDATA`) and trained weights are absent from this build: `synthetic_dataset` makes a closed-form stand-in
(ellipsoid meshes per class, ground-truth poses, perturbed initial poses, images rendered by the same HIP rasteriser at the
ground-truth pose) so that the whole epoch -- sharding, per-class batching, refinement, metrics, reduction -- runs end to end
on any number of ranks; `data_io` reads the reference's on-disk formats the day real data is supplied.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import distributed as D
from .distributed import METRICS, MetricAccumulator


@dataclass
class ClassModel:
    """What model/RNNPose.py:160-189 looks up per object class: mesh, per-vertex context / geometric features, diameter."""
    name: str
    verts: np.ndarray            # (P,3) float32, metres
    faces: np.ndarray            # (F,3) int32
    colors: np.ndarray           # (P,3) in [0,1]
    fea_3d: torch.Tensor         # (1,P,256) context features sampled on the vertices
    geofea_3d: torch.Tensor      # (1,P,32) geometric descriptors of the vertices
    diameter: float
    eval_points: np.ndarray = None   # model points of the metric (defaults to verts)


@dataclass
class EvalItem:
    """One evaluation sample (data/linemod_dataset.py:311-343 after preprocessing)."""
    class_name: str
    image: torch.Tensor          # (3,H,W) observed image
    K: np.ndarray                # (3,3)
    pose_init: np.ndarray        # (4,4) initial pose (PoseCNN / PVNet in the reference)
    pose_gt: np.ndarray          # (4,4)
    geofea_2d: torch.Tensor      # (32,H,W) descriptors of the observed image


class PackedEpochMetrics:
    """Initial-pose and refined-pose statistics of every class in ONE packed fp64 buffer -> one all_reduce per epoch."""

    def __init__(self, classes):
        self.classes = tuple(classes)
        self.init = MetricAccumulator(self.classes)
        self.refined = MetricAccumulator(self.classes)

    def reduce(self, device=None):
        both = MetricAccumulator(tuple(f"{k}/{c}" for k in ("init", "refined") for c in self.classes))
        both.sums = torch.cat([self.init.sums, self.refined.sums], 0)
        r = both.reduce(device=device)
        return {k: {c: r[f"{k}/{c}"] for c in self.classes} for k in ("init", "refined")}


def class_batches(items, indices, unique, batch_size):
    """The shard's samples as batches of ONE class each (model/RNNPose.py:158 asserts a single class per batch), in shard
    order: -> [(class, [item index], [unique flag])]."""
    out = []
    for i, u in zip(indices, unique):
        c = items[i].class_name
        if out and out[-1][0] == c and len(out[-1][1]) < batch_size:
            out[-1][1].append(i)
            out[-1][2].append(u)
        else:
            out.append((c, [i], [u]))
    return out


def flags_from_metrics(m, diameter, symmetric):
    """(B,5) [ADD, ADD-S, proj2d px, translation cm, rotation deg] -> (B,5) 0/1 flags in METRICS order
    (utils/eval_metric.py:102-192: ADD(-S) < 10 % / 2 % / 5 % of the diameter, proj2d < 5 px, 5 cm 5 deg)."""
    m = np.asarray(m, dtype=np.float64)
    dist = m[:, 1] if symmetric else m[:, 0]
    return np.stack([dist < 0.1 * diameter, dist < 0.02 * diameter, dist < 0.05 * diameter, m[:, 2] < 5.0,
                     (m[:, 3] < 5.0) & (m[:, 4] < 5.0)], 1).astype(np.float64)


def run_epoch(items, models, refine_fn, metric_fn, rank=0, world=1, batch_size=8, symmetric=(), reduce_device=None):
    """items: list[EvalItem]; models: {class: ClassModel};
    refine_fn(class_name, [EvalItem]) -> (B,4,4) refined poses (numpy or tensor): one PoseRefiner call per batch;
    metric_fn(class_name, pose_pred (B,4,4), pose_gt (B,4,4)) -> (B,5) [ADD, ADD-S, proj2d, t cm, r deg].
    -> {"init": {cls: {metric: mean, "n": count}}, "refined": {...}} identical on every rank (wrap-around duplicates of the
    sampler are excluded from the sums)."""
    classes = sorted(models)
    idx, uniq = D.shard_indices(len(items), rank, world)
    acc = PackedEpochMetrics(classes)
    for cls, ids, us in class_batches(items, idx, uniq, batch_size):
        batch = [items[i] for i in ids]
        gt = np.stack([it.pose_gt for it in batch]).astype(np.float32)
        init = np.stack([it.pose_init for it in batch]).astype(np.float32)
        pred = refine_fn(cls, batch)
        pred = pred.detach().cpu().numpy() if torch.is_tensor(pred) else np.asarray(pred)
        sym = cls in symmetric
        for which, poses in ((acc.init, init), (acc.refined, pred.reshape(-1, 4, 4))):
            fl = flags_from_metrics(metric_fn(cls, poses, gt), models[cls].diameter, sym)
            for row, u in zip(fl, us):
                which.update(cls, dict(zip(METRICS, row)), unique=u)
    return acc.reduce(device=reduce_device)


# ---- the GPU pieces behind refine_fn / metric_fn ---------------------------------------------------------------------
class HipEpoch:
    """PoseRefiner on the HIP mesh rasteriser + device metrics: the refine_fn / metric_fn pair of run_epoch."""

    def __init__(self, models, cfg=None, device="cuda", refiner=None, symmetric=("eggbox", "glue")):
        from .evaluator import LineMODEvaluator
        from .pose_refiner import PoseRefiner, default_config
        from .rasterizer import MeshRenderer
        self.models = models
        self.device = torch.device(device)
        self.renderer = MeshRenderer({n: dict(verts=m.verts, faces=m.faces, colors=m.colors) for n, m in models.items()},
                                     device=device)
        self.cfg = cfg if cfg is not None else default_config()
        self.refiner = refiner if refiner is not None else PoseRefiner(self.cfg, renderer=self.renderer).to(self.device).eval()
        self.symmetric = tuple(symmetric)
        self.evaluators = {n: LineMODEvaluator(n, m.eval_points if m.eval_points is not None else m.verts, m.diameter,
                                               device=device) for n, m in models.items()}
        for n, e in self.evaluators.items():
            e.symmetric = n in self.symmetric

    def refine(self, cls, batch):
        from .transformation import SE3Sequence
        dev, m = self.device, self.models[cls]
        image = torch.stack([it.image for it in batch]).to(dev)
        g2 = torch.stack([it.geofea_2d for it in batch]).to(dev)
        K = torch.as_tensor(np.stack([it.K for it in batch]).astype(np.float32)).to(dev)
        T0 = torch.as_tensor(np.stack([it.pose_init for it in batch]).astype(np.float32)).to(dev)
        Tg = torch.as_tensor(np.stack([it.pose_gt for it in batch]).astype(np.float32)).to(dev)
        out = self.refiner(image, SE3Sequence(matrix=T0[:, None]), K, fea_3d=m.fea_3d.to(dev), Tj_gt=SE3Sequence(matrix=Tg[:, None]),
                           obj_cls=[cls] * len(batch), geofea_3d=m.geofea_3d.to(dev), geofea_2d=g2)
        return out["Ti_pred"].G.reshape(-1, 4, 4)

    def metrics(self, cls, pose_pred, pose_gt):
        ev = self.evaluators[cls]
        T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
        return ev.evaluate(T(pose_pred), T(pose_gt)).cpu().numpy()


# ---- synthetic stand-in for EXPDATA ----------------------------------------------------------------------------------
def _ellipsoid(sub, scale):
    """Closed triangle mesh: subdivided icosahedron scaled to an ellipsoid (consistent outward winding)."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * np.array(scale)).astype(np.float32), np.array(f, np.int32)


def synthetic_models(class_names=("ape", "cat", "glue"), sub=3, seed=0):
    """Ellipsoid 'objects' of LINEMOD-like size (5-10 cm half axes) with hash-generated vertex features."""
    from . import synthetic as syn
    models = {}
    for k, name in enumerate(class_names):
        scale = (0.05 + 0.015 * k, 0.04 + 0.01 * ((k + 1) % 3), 0.035 + 0.01 * ((k + 2) % 3))
        verts, faces = _ellipsoid(sub, scale)
        P = verts.shape[0]
        g3 = syn.normal(f"g3:{name}", (1, P, 32), seed)
        g3 /= np.linalg.norm(g3, axis=-1, keepdims=True) + 1e-12
        d = verts[:, None, :] - verts[None, :, :]
        models[name] = ClassModel(name=name, verts=verts, faces=faces, colors=syn.uniform(f"col:{name}", (P, 3), seed),
                                  fea_3d=torch.from_numpy(syn.normal(f"f3:{name}", (1, P, 256), seed, std=0.5)),
                                  geofea_3d=torch.from_numpy(g3.astype(np.float32)),
                                  diameter=float(np.sqrt((d * d).sum(-1)).max()))
    return models


def synthetic_dataset(models, n_items, image_size=(480, 640), seed=0, pose_sigma=(0.05, 0.01), renderer=None, device="cuda"):
    """n_items samples cycling through the classes in blocks (as a per-class LINEMOD sequence does): ground-truth pose in front
    of the camera, initial pose = exp(xi) * gt with xi ~ N(0, pose_sigma (rotation rad, translation m)); the observed image
    and its descriptors are RENDERED at the ground-truth pose by `renderer` (MeshRenderer) -- or hash noise when None (CPU)."""
    from . import synthetic as syn
    from .evaluator import LINEMOD_K
    H, W = image_size
    names = sorted(models)
    K = LINEMOD_K.copy()
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    items = []
    block = max(1, -(-n_items // len(names)))
    for i in range(n_items):
        cls = names[min(i // block, len(names) - 1)]
        g = syn.se3_exp_np(syn.normal(f"gt{i}", (1, 6), seed, std=0.6))[0]
        g[:3, 3] = syn.uniform(f"t{i}", (3,), seed, -0.03, 0.03) + np.array([0.0, 0.0, 0.8])
        xi = syn.normal(f"xi{i}", (1, 6), seed)[0] * np.array([pose_sigma[1]] * 3 + [pose_sigma[0]] * 3)
        init = syn.se3_exp_np(xi[None])[0] @ g
        items.append(EvalItem(cls, None, K.copy(), init.astype(np.float32), g.astype(np.float32), None))
    if renderer is None:
        for i, it in enumerate(items):
            it.image = torch.from_numpy(syn.uniform(f"img{i}", (3, H, W), seed) * 255.0)
            it.geofea_2d = torch.from_numpy(syn.normal(f"g2{i}", (32, H, W), seed))
        return items
    dev = torch.device(device)
    for cls in names:
        ids = [i for i, it in enumerate(items) if it.class_name == cls]
        m = models[cls]
        for a in range(0, len(ids), 8):
            sub_ids = ids[a:a + 8]
            Tg = torch.as_tensor(np.stack([items[i].pose_gt for i in sub_ids])).to(dev)
            Kt = torch.as_tensor(np.stack([items[i].K for i in sub_ids])).to(dev)
            out, _ = renderer([cls] * len(sub_ids), m.geofea_3d.to(dev), T=Tg, K=Kt, render_image_size=(H, W), render_tex=True)
            for j, i in enumerate(sub_ids):
                items[i].image = (out[j, :3] * 255.0).cpu()
                items[i].geofea_2d = out[j, 3:].cpu()
    return items
