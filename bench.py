#!/usr/bin/env python3
"""bench.py -- RNNPose recurrent pose-refinement throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one full refinement of one batch: 3 outer x 8 inner iterations of the loop body
(model/PoseRefiner.py:239-365) on synthetic 640x480 render+target pairs, batch 8 per GPU
(BASELINE.json configs[1]); per outer iteration: RAFT encoder + correlation volume/pyramid + context prep;
per inner iteration: induced flow -> pyramid lookup -> update block -> convex upsample -> descriptor
weight -> LM step.  The renderer is outside the path: views are synthetic and fixed.
Inputs are resident in HBM before the timed region.  value = refinement iterations/s (one iteration = the
loop body for one rank's batch of 8), summed over ranks (weak scaling: every rank refines its own batch).

Rank 0 prints ONE JSON line (see the task contract) carrying `roofline` for the correlation-volume kernel
(measured live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle, timed on this
box's host cores on a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--outer", type=int, default=3)
    ap.add_argument("--inner", type=int, default=8)
    ap.add_argument("--optim-iters", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoder", action="store_true", help="feed synthetic feature maps (kernel-only runs)")
    ap.add_argument("--unfused", action="store_true", help="literal reference call sequence through the facade")
    ap.add_argument("--no-graph", action="store_true", help="eager launches only (no hipGraph replay; for counter passes)")
    ap.add_argument("--conv-backend", choices=["hip", "miopen"], default="hip",
                    help="update-block convolutions: hand-written fp16x3 implicit GEMM (default) or torch/MIOpen fp32")
    return ap.parse_args()


def synth_views(B, H, W, device, seed, with_encoder):
    """Synthetic, fixed, already-cropped views in HBM (shapes/ranges of SURVEY.md section 8d)."""
    from rnnpose_amd import ops
    from rnnpose_amd.pose_refiner import SyntheticRenderer
    from rnnpose_amd.synthetic import intrinsics
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    rnd = lambda *s: torch.randn(*s, device=device, generator=g)
    uni = lambda *s: torch.rand(*s, device=device, generator=g)
    syn_img, image_crop = uni(B, 3, H, W), uni(B, 3, H, W)
    cfea = rnd(B, 256, H, W).mul_(0.1)
    g1, g2 = rnd(B, 32, H, W), rnd(B, 32, H, W)
    g1 /= g1.norm(dim=1, keepdim=True)
    g2 /= g2.norm(dim=1, keepdim=True)
    depth = uni(B, 1, H, W) * 0.3 + 0.9
    depth[:, :, : H // 4] = 0
    K = torch.from_numpy(intrinsics(B, H, W)).to(device)
    xi = rnd(B, 6) * 0.02
    G0 = ops.se3_exp(xi).reshape(B, 1, 4, 4)
    fm = (None, None) if with_encoder else (rnd(B, 256, H // 8, W // 8), rnd(B, 256, H // 8, W // 8))
    rend = SyntheticRenderer(syn_img=syn_img, image_crop=image_crop, cfea=cfea, geofea1=g1, geofea2_crop=g2,
                             syn_depth=depth, intrinsics_crop=K, fmap1=fm[0], fmap2=fm[1])
    return rend, K, G0


def cpu_baseline(refiner, rend, K, G0, args):
    """CPU oracle (oracle/rnnpose_oracle.py, kind 'port') on a bounded sample of the same workload:
    the full batch, 1 outer x 2 inner iterations; per-stage timers extrapolate to the 3x8 schedule."""
    from oracle import rnnpose_oracle as orc
    v = rend.views
    ncpu = os.cpu_count() or 1
    cores = int(os.environ.get("RNNPOSE_CPU_THREADS", min(ncpu, 64)))   # torch-CPU stops scaling well before 256 SMT threads
    torch.set_num_threads(cores)
    nb = min(args.batch, 4)                                             # bounded sample: 4 images of the batch
    sl = lambda t: t[:nb]
    inp = {"ctx": sl(v["cfea"]), "g1": sl(v["geofea1"]), "g2": sl(v["geofea2_crop"]), "depth": sl(v["syn_depth"]),
           "K": sl(K), "G0": sl(G0), "sigma": refiner.sigma[0].detach()}
    W = {"upd": {k: p.detach().cpu().numpy() for k, p in refiner.cf_net.update_block.state_dict().items()}}
    if v["fmap1"] is None:
        inp["img_render"], inp["img_target"] = sl(v["syn_img"]), sl(v["image_crop"])
        W["enc"] = {k: p.detach().cpu().numpy() for k, p in refiner.image_fea_enc.fnet.state_dict().items()}
    else:
        inp["fmap1"], inp["fmap2"] = sl(v["fmap1"]), sl(v["fmap2"])
    inp = {k: t.detach().cpu().numpy() for k, t in inp.items()}
    orc.refine(inp, W, outer=1, inner=1, optim_iters=args.optim_iters, fast=True)          # warm-up (thread pools, oneDNN)
    tm = {}
    t0 = time.perf_counter()
    orc.refine(inp, W, outer=1, inner=3, optim_iters=args.optim_iters, stage_timer=tm, fast=True)
    wall = time.perf_counter() - t0
    scale = args.batch / nb                                             # per-image cost is batch-independent
    t_outer = (tm.get("encoder", 0.0) + tm.get("corr_build_ctx", 0.0)) * scale
    t_inner = (wall * scale - t_outer) / 3.0
    sched = args.outer * t_outer + args.outer * args.inner * t_inner
    return {
        "value": args.outer * args.inner / sched, "unit": "iters/s", "cores": cores, "kind": "port",
        "sample": (f"{nb} of the {args.batch} images ({args.height}x{args.width}), 1 outer x 3 inner iterations of the CPU "
                   f"oracle in its library-call form (grid_sample/unfold as the reference uses on CPU), {wall:.1f} s wall, "
                   f"torch {torch.__version__} CPU with {cores} threads on {ncpu} logical CPUs; scaled x{scale:g} to the "
                   f"batch: per-outer {t_outer:.2f} s, per-inner {t_inner:.2f} s, extrapolated to {args.outer}x{args.inner}"),
        "stages_s": {k: round(x, 3) for k, x in tm.items()},
    }


def main():
    args = parse()
    from rnnpose_amd import distributed as D
    rank, world, local = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: rnnpose_amd has no CPU product path")
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    from rnnpose_amd import build, ops
    build.build()
    from rnnpose_amd.pose_refiner import PoseRefiner, default_config
    from rnnpose_amd.transformation import SE3Sequence

    B, H, W = args.batch, args.height, args.width
    torch.manual_seed(0)
    rend, K, G0 = synth_views(B, H, W, device, seed=rank, with_encoder=not args.no_encoder)
    cfg = default_config(RENDER_ITER_COUNT=args.outer, ITER_COUNT=args.inner, OPTIM_ITER_COUNT=args.optim_iters)
    cfg.raft.conv_backend = args.conv_backend
    refiner = PoseRefiner(cfg, renderer=rend, fused=not args.unfused, use_graph=not args.no_graph).to(device).eval()

    def step():
        return refiner(rend.views["image_crop"], SE3Sequence(matrix=G0.clone()), K)

    # setup (not a warm-up step): one eager pass under the event recorder -- the event-instrumented timed step below runs
    # eagerly too and would otherwise pay first-use allocations (1 GB volume buffer, ...) inside the timed region --
    # and one pass that captures the hipGraphs the remaining steps replay.
    with ops.profile(None):
        step()
    step()
    for _ in range(args.warmup):
        step()
    hip_ops = None          # HIP events around EVERY C-ABI launch (~25 per iteration; <1 % of the step)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # HIP events bracket every C-ABI launch of the FIRST OUTER ITERATION of the first timed step only (~450 launches,
    # eager): the rest of that step and the other K-1 steps replay hipGraphs, so that a small K does not dilute `value`.
    rec = refiner.profile_first_outer() if (args.outer > 1 and not args.unfused) else ops.profile_begin(None)
    out = step()
    if refiner.profile_rec is not None or ops.profiling():      # single outer iteration / unfused: the whole step was recorded
        ops.profile_end(rec)
        refiner.profile_rec = None
    for _ in range(args.steps - 1):
        out = step()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    prof = ops.summarize(rec)
    prof_steps = (1.0 / args.outer) if (args.outer > 1 and not args.unfused) else 1.0     # fraction of one step that was instrumented

    if rank != 0:
        return
    iters = args.outer * args.inner
    value = world * args.steps * iters / dt
    h, w = H // 8, W // 8
    N = h * w
    C = 256
    # algorithmic work per launch (SURVEY.md section 8d formulas x batch)
    lvl = sum((h >> l) * (w >> l) for l in range(4))
    alg = {
        "rnnpose_corr_pyramid_f32": dict(bytes=4 * (2 * N * C + N * lvl) * B, flops=2 * N * N * C * B),
        "rnnpose_corr_lookup_f32": dict(bytes=4 * N * (4 * 100 + 4 * 81 + 2) * B),
        "rnnpose_convex_upsample_f32": dict(bytes=(4 * N * (576 + 2) + 8 * H * W) * B),
        "rnnpose_corr_weight_f32": dict(bytes=(4 * 32 * 2 + 8 + 4 + 4) * H * W * B),
        "rnnpose_lm_step_f32": dict(bytes=16 * H * W * B * args.optim_iters),
    }
    alg["rnnpose_corr_pyramid_f16x3"] = alg["rnnpose_corr_pyramid_f32"]
    alg["rnnpose_corr_lookup_nhwc_f32"] = alg["rnnpose_corr_lookup_f32"]
    alg["rnnpose_convex_upsample_nhwc_f32"] = alg["rnnpose_convex_upsample_f32"]
    kernels = {}
    for name, (n, mean_ms, tot_ms, work) in prof.items():
        e = {"launches": n, "mean_ms": round(mean_ms, 4), "share_of_step": round(tot_ms / prof_steps / (dt / args.steps * 1e3), 4)}
        if name in alg:
            e["GBps"] = round(alg[name]["bytes"] / (mean_ms * 1e-3) / 1e9, 1)
            e["hbm_frac"] = round(e["GBps"] / PEAK_HBM_GBS, 4)
            if "flops" in alg[name]:
                e["TFLOPps"] = round(alg[name]["flops"] / (mean_ms * 1e-3) / 1e12, 2)
                e["mfma_frac"] = round(e["TFLOPps"] / PEAK_F32_MFMA_TFLOPS, 4)
        if work:
            e["TFLOPps_fp32_equivalent"] = round(work / (tot_ms * 1e-3) / 1e12, 2)
        kernels[name.replace("rnnpose_", "")] = e
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = {}
    # roofline object = the DOMINANT hand-written kernel of the timed region
    dom = max(prof.items(), key=lambda kv: kv[1][2])[0] if prof else None
    roofline = None
    if dom == "rnnpose_conv2d_nhwc_f16x3":
        n, mean_ms, tot_ms, work = prof[dom]
        eq = work / (tot_ms * 1e-3) / 1e12
        roofline = {"kernel": "conv_igemm_f16x3_kernel (NHWC implicit-GEMM convolution, fp16x3-split MFMA = fp32-class accuracy; "
                              "all update-block and stride-1 encoder convolutions)",
                    "bound": "mfma", "achieved": round(3 * eq, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(3 * eq / PEAK_F16_MFMA_TFLOPS, 4),
                    "note": "achieved = EXECUTED fp16 MFMA flops (3 products per algorithmic multiply-add) / summed HIP-event time of "
                            "all launches in the timed region; fp32-equivalent algorithmic rate = achieved/3.  The update step runs "
                            "the two halves of the batch as two concurrent convolution chains (two streams): every launch is timed "
                            "while it SHARES the chip with its twin, so the per-launch rate understates the aggregate matrix-pipe use "
                            "(RNNPOSE_SPLIT_BATCH=0, one full-batch chain: 650 TF per launch at 5 % lower iters/s)",
                    "fp32_equivalent_TFLOPps": round(eq, 1), "launches_timed": n, "mean_ms": round(mean_ms, 4),
                    "share_of_step": round(tot_ms / prof_steps / (dt / args.steps * 1e3), 4), "algorithmic_flops_timed": work,
                    "traffic": traffic.get("conv_igemm_bytes_per_launch")}
    elif dom == "rnnpose_corr_pyramid_f32" or (dom and "rnnpose_corr_pyramid_f32" in prof):
        cp = prof["rnnpose_corr_pyramid_f32"]
        ach = alg["rnnpose_corr_pyramid_f32"]["flops"] / (cp[1] * 1e-3) / 1e12
        roofline = {"kernel": "corr_pyramid_kernel (fp32 MFMA all-pairs correlation + fused 4-level pyramid)",
                    "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic.get("corr_pyramid_bytes_per_launch"),
                    "launches_timed": cp[0], "mean_ms": round(cp[1], 4)}
    # north_star's named kernel is always reported beside it
    corr_vol = None
    if "rnnpose_corr_pyramid_f16x3" in prof:
        cp = prof["rnnpose_corr_pyramid_f16x3"]
        a = alg["rnnpose_corr_pyramid_f16x3"]
        gbs = a["bytes"] / (cp[1] * 1e-3) / 1e9
        corr_vol = {"kernel": "corr_pyramid_h3_kernel (+ 2 split_features_kernel pre-passes inside the same C-ABI call)",
                    "bound": "hbm (volume + pooled levels written once; fp16x3-split MFMA: 3 fp16 products per multiply-add)",
                    "achieved_GBps": round(gbs, 1), "peak_GBps": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "executed_fp16_TFLOPps": round(3 * a["flops"] / (cp[1] * 1e-3) / 1e12, 1),
                    "fp32_equivalent_TFLOPps": round(a["flops"] / (cp[1] * 1e-3) / 1e12, 1),
                    "algorithmic_bytes_per_launch": a["bytes"], "traffic": traffic.get("corr_pyramid_h3_bytes_per_launch"),
                    "mean_ms": round(cp[1], 4)}
    elif "rnnpose_corr_pyramid_f32" in prof:
        cp = prof["rnnpose_corr_pyramid_f32"]
        ach = alg["rnnpose_corr_pyramid_f32"]["flops"] / (cp[1] * 1e-3) / 1e12
        corr_vol = {"kernel": "corr_pyramid_kernel", "bound": "mfma (fp32, 96 flop/B)", "achieved_TFLOPps": round(ach, 2),
                    "peak_TFLOPps": PEAK_F32_MFMA_TFLOPS, "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    "hbm_GBps": kernels["corr_pyramid_f32"]["GBps"], "hbm_frac": kernels["corr_pyramid_f32"]["hbm_frac"],
                    "algorithmic_bytes_per_launch": alg["rnnpose_corr_pyramid_f32"]["bytes"],
                    "traffic": traffic.get("corr_pyramid_bytes_per_launch"), "mean_ms": round(cp[1], 4)}
    res = {
        "metric": "pose-refine iters/sec (640x480, B=8, 3x8 recurrent)", "value": round(value, 3), "unit": "iters/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.conv_backend == "miopen" else "f32 (convolutions: fp16x3-split MFMA, fp32-class accuracy; LM: f64)",
        "data": "synthetic", "image_iters_per_sec": round(value * B, 2),
        "config": {"workload": f"synthetic {W}x{H} render+target pairs, batch {B}/GPU, {args.outer} outer x "
                               f"{args.inner} inner refinement (BASELINE.json configs[1]); 1 step = 1 refinement = "
                               f"{iters} iterations", "batch_per_gpu": B, "height": H, "width": W,
                   "outer": args.outer, "inner": args.inner, "optim_iters": args.optim_iters,
                   "encoder_in_timed_region": not args.no_encoder, "fused_schedule": not args.unfused,
                   "conv_backend": args.conv_backend,
                   "hip_graphs": "encoder+volume build and the inner-iteration body replay as hipGraphs (except in the "
                                 "event-instrumented first timed step)" if refiner.use_graph and not args.unfused else "off",
                   "weights": "random init", "lm_accumulation": "f64", "sharding": f"dp{world} (independent images, no collective in the path)"},
        "roofline": roofline, "correlation_volume_kernel": corr_vol, "kernels": kernels,
        "kernels_note": "HIP events around every C-ABI launch of the first outer iteration of the first timed step (eager); "
                        "share_of_step = summed launch durations x outer iterations / step time -- launches of the two batch "
                        "halves overlap on two streams, so the shares add up to more than 1",
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(refiner, rend, K, G0, args)
        res["speedup_vs_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res))


if __name__ == "__main__":
    main()
