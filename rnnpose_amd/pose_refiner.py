"""PoseRefiner: the recurrent refinement loop with the reference's module API (model/PoseRefiner.py:60-376).

    refiner = PoseRefiner(cfg, renderer=renderer)
    out = refiner(image, Ts, intrinsics, fea_3d, Tj_gt, obj_cls, geofea_3d, geofea_2d)   # dict, same keys

state_dict keys match the reference (`sigma.0`, `image_fea_enc.fnet.*`, `cf_net.update_block.*`) so a
checkpoint's `motion_net.*` sub-tree loads unchanged.

What is different, on purpose (DESIGN.md section "Batched semantics"):
  * B > 1 is defined as "the B=1 computation per sample" (the reference broadcasts masks wrongly for B>1,
    model/PoseRefiner.py:328,336-338);
  * rendering / zoom-cropping (PyTorch3D, cv2) is NOT part of the hot path: a `renderer` object hands over the
    per-outer-iteration views (protocol below); `SyntheticRenderer` serves fixed synthetic views;
  * fused=True (default) runs the MI355X schedule: the induced flow is evaluated only at the 4 taps every
    1/8-res pixel needs (no full-res flow_init), the correspondence target is never materialised (kernels add
    the pixel grid to the planar flow), and normal equations + solve + SE(3) update stay on the device.
    fused=False replays the reference's literal call sequence through the facade classes (same kernels
    underneath); tests compare both.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .cfnet import AttrDict, GRU_CFUpdator, ImageFeaEncoder
from .corr import coords_grid as coords_grid_lowres
from .transformation import EP_LMBDA, LM_LMBDA, SE3Sequence, coords_grid

EPS = 1e-5          # model/PoseRefiner.py:21


def default_config(**over):
    """Iteration counts of config/linemod/template_fw0.5.yml:76-81,92 unless overridden."""
    cfg = AttrDict(RENDER_ITER_COUNT=3, ITER_COUNT=4, OPTIM_ITER_COUNT=1, FLOW_NET="raft", ONLINE_CROP=True,
                   IS_CALIBRATED=True, RESCALE_IMAGES=False, with_corr_weight=True,
                   raft=AttrDict(pretrained_model=None, mixed_precision=False, fea_net="default"),
                   render_image_size=(480, 640), zoom_crop_size=(240, 240),       # BASIC.* of the reference config
                   LM_LMBDA=LM_LMBDA, EP_LMBDA=EP_LMBDA)
    cfg.update(over)
    return cfg


class SyntheticRenderer:
    """Hands the loop fixed, already-cropped synthetic views.  A real integration
    implements the same method on top of its rasteriser (reference: model/PoseRefiner.py:253-304)."""

    def __init__(self, syn_img, image_crop, cfea, geofea1, geofea2_crop, syn_depth, intrinsics_crop,
                 fmap1=None, fmap2=None):
        self.views = dict(syn_img=syn_img, image_crop=image_crop, cfea=cfea, geofea1=geofea1,
                          geofea2_crop=geofea2_crop, syn_depth=syn_depth, intrinsics_crop=intrinsics_crop,
                          fmap1=fmap1, fmap2=fmap2)

    def render_views(self, Ti, intrinsics, obj_cls=None, image=None, fea_3d=None, geofea_3d=None, geofea_2d=None):
        return self.views


class PoseRefiner(nn.Module):
    def __init__(self, cfg=None, reuse=False, schedule=None, use_regressor=True, is_calibrated=True,
                 bn_is_training=False, is_training=True, renderer=None, fused=True,
                 img_fea_enc_weights=None, use_graph=True, literal_legacy_pose=None):
        """literal_legacy_pose (or cfg["literal_legacy_pose"]; default True since r06): start every outer iteration from the reference's
        legacy product Tij = Ti * Ti.inv() (model/PoseRefiner.py:243-244) -- what the reference computes.  False: the exact identity
        that product stands for (the r03-r05 default; on the reference-generated fixtures it is FURTHER from the reference on three of
        four, and crosses 1e-4 px at 960 x 1280: profiles/r05_fixture_distances.txt)."""
        super().__init__()
        self.legacy = True
        self.cfg = cfg = cfg if cfg is not None else default_config()
        self.literal_legacy_pose = bool(cfg.get("literal_legacy_pose", True) if literal_legacy_pose is None else literal_legacy_pose)
        self.reuse = reuse
        self.sigma = nn.ParameterList([nn.Parameter(torch.ones(1) * 1)])
        self.with_corr_weight = cfg.get("with_corr_weight", True)
        self.is_calibrated = cfg.get("IS_CALIBRATED", True) and is_calibrated
        self.is_training = is_training
        self.use_regressor = use_regressor
        if cfg.get("FLOW_NET", "raft") != "raft":
            raise NotImplementedError
        self.image_fea_enc = ImageFeaEncoder(pretrained=img_fea_enc_weights)
        self.cf_net = GRU_CFUpdator(cfg.get("raft", None))
        if renderer is not None and not hasattr(renderer, "render_views"):
            # an object with the reference renderer's call shape (DiffRendererWrapper, geometry/diff_render_optim.py:404-494;
            # constructed as PoseRefiner(opt.motion_net, ..., renderer=diff_renderer) at model/RNNPose.py:76-79)
            from .render_adapter import RendererAdapter
            renderer = RendererAdapter(renderer, render_image_size=cfg.get("render_image_size", (480, 640)),
                                       zoom_crop_size=cfg.get("zoom_crop_size", (240, 240)), legacy=True)
        self.renderer = renderer
        self.fused = fused
        # hipGraph replay of the inner-iteration body (~25 launches): at the reference's own working size (B=1,
        # 240x240) the loop is launch-bound, not GPU-bound.  Falls back to eager launches if capture is refused.
        self.use_graph = use_graph
        self.split_fmaps = os.environ.get("RNNPOSE_SPLIT_FMAPS", "1") != "0"
        self.mixed_precision = bool(cfg.get("raft", {}).get("mixed_precision", False)) or os.environ.get("RNNPOSE_MIXED_PRECISION", "0") != "0"
        if self.mixed_precision:
            # the reference's YAMLs set raft.mixed_precision: True (config/linemod/template_fw0.5.yml:88): loading one changes the arithmetic
            import warnings
            warnings.warn("cfg.raft.mixed_precision: the 160-row strip convolutions run ONE fp16 product per multiply-add (the reference's GPU "
                          "autocast arithmetic, ~2^-11 per product).  The parity tolerances (1e-4 flow / 1e-5 pose against the fp32 CPU path) "
                          "are stated for mixed_precision = False; this mode is not pinned against the reference's autocast run.")
        self.profile_rec = None           # set by profile_first_outer(): record of the instrumented first outer iteration
        # inner-iteration graphs, one record PER INPUT SHAPE (a partial last evaluation batch followed by a full one must not
        # evict each other: ADVICE r02): {"gr": graph captured on the caller's tensors (keyed by their addresses), "captures":
        # how often this shape was re-captured because the addresses moved, "static": the graph over persistent input copies
        # that takes over once they keep moving, "sbuf": those copies}
        self._shapes = {}
        self._outer_graphs = {}
        self._outer_captures = 0
        self.loop_timing = None           # set to [] to collect HIP events around the per-half graph replays
        self._wkey = None                 # identity of the live parameters + engine buffers every captured graph depends on
        self._clear()

    def _clear(self):
        self.residual_pose_history = []
        self.Ti_history = []
        self.flow_history = []
        self.intrinsics_history = []

    def __len__(self):
        return len(self.residual_pose_history)

    def profile_first_outer(self):
        """bench.py: HIP events around every C-ABI launch of the FIRST outer iteration of the next forward() only (one
        third of a 3x8 refinement runs eagerly; instrumenting the whole step cost ~10 ms of a 46-ms step)."""
        self.profile_rec = ops.profile_begin(None)
        return self.profile_rec

    # ---- per-outer-iteration unit: RAFT encoder (PoseRefiner.py:311) + CorrBlock build + context prep (CFNet.py:115-133)
    def _outer_body(self, views):
        if views.get("fmap1") is not None:
            feats1, feats2 = views["fmap1"], views["fmap2"]
        else:
            # the encoder's output convolution writes the volume build's operand format directly (fp16 hi|lo, pixel-major):
            # no NCHW transposition, no split pre-pass.  RNNPOSE_SPLIT_FMAPS=0: fp32 NCHW maps as the reference hands over.
            if self.split_fmaps and self.cf_net.corr_precision == "f16x3":
                feats1, feats2 = self.image_fea_enc.forward_split(views["syn_img"], views["image_crop"])
            else:
                feats1, feats2 = self.image_fea_enc(views["syn_img"], views["image_crop"])
        self.cf_net.prepare(feats1, feats2, views["cfea"])
        return feats1, feats2

    def _drop_graphs(self):
        self._shapes = {}
        self._outer_graphs = {}
        self._outer_captures = 0

    # (compatibility views for tests / tools: the record of the most recently used shape)
    @property
    def _ptr_captures(self):
        return self._last_shape["captures"] if getattr(self, "_last_shape", None) else 0

    @property
    def _graph_static(self):
        return self._last_shape["static"] if getattr(self, "_last_shape", None) else None

    @property
    def _graph(self):
        return self._last_shape["gr"] if getattr(self, "_last_shape", None) else None

    def _refresh(self):
        """Once per forward(): re-pack weights whose parameters changed (load_state_dict, in-place updates, .to()) and
        return the identity every captured graph depends on -- replay never runs the packing code itself."""
        eng = self.cf_net.engine()
        # cfg.raft.mixed_precision (the reference's GPU arithmetic, model/CFNet.py:47,126,152): single-product fp16 convolutions in the
        # strip kernels; RNNPOSE_MIXED_PRECISION=1 forces it for measurements.  Part of the graph key (graphs bake the kernels in).
        # The mode lives on THIS refiner's engines, not in a process-wide default: direct calls of ops / another refiner's modules are
        # not affected by what this one ran last.
        enc = self.image_fea_enc.engine()
        eng.single_product = enc.single_product = self.mixed_precision
        key = (eng.refresh(), enc.refresh(), self.sigma[0].data_ptr(), eng.epoch, ops.range_guard_state(),
               self.mixed_precision)
        if key != self._wkey:
            if self._wkey is not None:
                self._drop_graphs()          # graphs hold pointers to the old packed weights / freed activation buffers
            self._wkey = key
        return key

    def _outer(self, views):
        if not self.use_graph or ops.profiling():
            return self._outer_body(views)
        eng = self.cf_net.engine()
        ins = [views[k] for k in ("syn_img", "image_crop", "cfea", "fmap1", "fmap2") if views.get(k) is not None]
        key = tuple((t.data_ptr(), tuple(t.shape)) for t in ins) + (self._wkey,)
        gr = self._outer_graphs.get(key)
        if gr is None:
            if self._outer_captures >= 8:      # views keep moving (a renderer allocating fresh tensors): stay eager
                return self._outer_body(views)
            self._outer_captures += 1
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._outer_body(views)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._outer_body(views)
                gr = dict(graph=graph, out=out, corr_fn=self.cf_net.corr_fn, net=self.cf_net._net, inp=self.cf_net.inp,
                          buf_key=eng.buffer_key())
                if len(self._outer_graphs) >= 4:
                    self._outer_graphs.pop(next(iter(self._outer_graphs)))
                self._outer_graphs[key] = gr
            except Exception as e:             # noqa: BLE001
                import warnings
                warnings.warn(f"hipGraph capture of encoder+volume build failed ({e!r}); running eager launches")
                self._outer_captures = 8
                return self._outer_body(views)
        gr["graph"].replay()
        # the Python-side state prepare() sets is the one recorded at capture time (same device buffers)
        eng.select(gr["buf_key"])
        self.cf_net.corr_fn = gr["corr_fn"]
        self.cf_net._net, self.cf_net.inp = gr["net"], gr["inp"]
        self.cf_net._net_in_engine = True
        self.cf_net.fmap1, self.cf_net.fmap2 = gr["out"]
        return gr["out"]

    # ---- one inner iteration of the fused schedule, eager or as a replayed hipGraph ---------------------------
    @staticmethod
    def _out_views(big, small, B, H, W):
        """Typed views of the two per-iteration output buffers (one clone each hands a replayed graph's outputs over):
        big = [flow_up (B,2,H,W) | weight (B,H,W)] fp32; small = [Hm (B,6,6) f64 | bv (B,6) f64 | G (B,4,4) | xi (B,6) | info (B,) i32]."""
        P = H * W
        flow_up = big[:B * 2 * P].view(B, 2, H, W)
        wmap = big[B * 2 * P:].view(B, H, W)
        o = [0, B * 288, B * 336, B * 400, B * 424, B * 428]
        Hm = small[o[0]:o[1]].view(torch.float64).view(B, 6, 6)
        bv = small[o[1]:o[2]].view(torch.float64).view(B, 6)
        Gn = small[o[2]:o[3]].view(torch.float32).view(B, 4, 4)
        xi = small[o[3]:o[4]].view(torch.float32).view(B, 6)
        info = small[o[4]:o[5]].view(torch.int32).view(B)
        return flow_up, wmap, Gn, Hm, bv, xi, info

    def _loop_buffers(self, B, H, W, h, w, n, dev):
        """Output / scratch buffers of the n inner iterations of one outer iteration: big (n, B*3*H*W) and small (n, .)
        rows = iteration i (see _out_views); coords (n,B,2,h,w) the re-projected low-res coordinates."""
        big = torch.empty(n, B * 3 * H * W, device=dev, dtype=torch.float32)
        small = torch.empty(n, -(-B * 428 // 16) * 16, device=dev, dtype=torch.uint8)     # rows 16-byte aligned (fp64 views)
        return dict(big=big, small=small, coords=torch.empty(n, B, 2, h, w, device=dev, dtype=torch.float32),
                    views=[self._out_views(big[i], small[i], B, H, W) for i in range(n)])

    def _half_loop(self, bufs, b0, b1, st, depth, K, g1, g2, G3, h, w, ep_l, lm_l, n, single):
        """Generator: the COMPLETE inner loop of images [b0, b1) on stream `st` -- re-projection of their poses -> window
        lookup -> update block -> up-sampling -> descriptor weight -> LM step, n times (PoseRefiner.py:315-362).  Images are
        independent, so nothing synchronises the batch halves per iteration: each half runs its own loop on its own stream
        (r02 timeline: a per-iteration fork cost ~115 us of every 1.3-ms iteration -- cross-queue dependencies are that
        slow on this GPU), and the halves drift out of phase, so one half's memory-bound tail runs under the other half's
        convolutions."""
        eng = self.cf_net.engine()
        B = depth.shape[0]
        opt = self.cfg.OPTIM_ITER_COUNT
        Gc = G3[b0:b1]
        for i in range(n):
            flow_up, wmap, Gn, Hm, bv, xi, info = (t[b0:b1] for t in bufs["views"][i])
            if eng.fused_induced and not eng.fused_lookup:      # r06: :324-328 + CFNet.py:136-144 evaluated inside the lookup / flow-feature launches
                coords1 = bufs["coords"][i, b0:b1]
                ic = ops.InducedCoords(depth[b0:b1], K[b0:b1], Gc, h, w, EPS, coords1)
            else:
                coords1, ic = ops.induced_coords_lowres(depth[b0:b1], K[b0:b1], Gc, h, w, EPS, out=bufs["coords"][i, b0:b1]), None   # :324-328, CFNet.py:136-144
                yield
            yield from eng.half_gen(self.cf_net.corr_fn, coords1, B, b0, b1, st, flow_up, single=single, induced=ic)
            ops.corr_weight(g1[b0:b1], g2[b0:b1], flow_up, depth[b0:b1], self.sigma[0], out=wmap)                      # :342-345
            yield
            ops.lm_step(flow_up, wmap, depth[b0:b1], K[b0:b1], Gc, num_iters=opt, ep_lambda=ep_l, lm_lambda=lm_l,
                        max_update=1.0, eps=EPS, out=(Gn, Hm, bv, xi, info), slot=b0)                                   # :349-356
            yield
            Gc = Gn

    def _loop(self, depth, K, g1, g2, G, h, w, ep_l, lm_l, n):
        """Eager launches of the n inner iterations -> (big, small); the batch halves' launches are issued alternately."""
        B, dev = depth.shape[0], depth.device
        eng = self.cf_net.engine()
        bufs = self._loop_buffers(B, depth.shape[-2], depth.shape[-1], h, w, n, dev)
        G3 = G.reshape(-1, 4, 4)
        hv = eng.halves(B)
        main = torch.cuda.current_stream()
        jobs = []
        for k, (b0, b1) in enumerate(hv):
            st = main if k == 0 else eng._stream(dev, k)
            jobs.append((self._half_loop(bufs, b0, b1, st, depth, K, g1, g2, G3, h, w, ep_l, lm_l, n, len(hv) == 1), st))
        eng.run_interleaved(jobs, main)
        return bufs["big"], bufs["small"]

    def _inner_loop(self, depth, K, g1, g2, G, h, w, ep_l, lm_l, n):
        """-> list of n tuples (flow_up, wmap, G, Hm, bv, xi, info): eager launches, or one replayed hipGraph PER BATCH HALF
        (each a linear chain on its own stream) per outer iteration."""
        B, H, W = depth.shape[0], depth.shape[-2], depth.shape[-1]
        unpack = lambda big, small: [self._out_views(big[i], small[i], B, H, W) for i in range(n)]
        if not self.use_graph or ops.profiling():
            return unpack(*self._loop(depth, K, g1, g2, G, h, w, ep_l, lm_l, n))
        eng = self.cf_net.engine()
        key = (depth.data_ptr(), K.data_ptr(), g1.data_ptr(), g2.data_ptr(), self.cf_net.corr_fn._buf.data_ptr(),
               tuple(depth.shape), n, self.cfg.OPTIM_ITER_COUNT, float(ep_l), float(lm_l), eng.buffer_key(), self._wkey)
        skey = key[5:]                          # everything but the addresses: one record per shape
        rec = self._shapes.get(skey)
        if rec is None:
            if len(self._shapes) >= eng.MAX_SETS:      # as many shapes as the engine keeps activation buffer sets for
                self._shapes.pop(next(iter(self._shapes)))
            rec = self._shapes[skey] = dict(gr=None, captures=0, static=None, sbuf=None)
        self._last_shape = rec
        gr = rec["gr"]
        if gr is None or gr["key"] != key:
            if rec["captures"] >= 2:
                # the views keep moving (a renderer that allocates fresh tensors every outer iteration): re-capturing per
                # pointer set would cost more than it saves.  Switch to ONE graph set over persistent input buffers and pay
                # a device copy of depth / K / descriptors (~0.65 GB at 480x640, B=8: ~0.25 ms) per outer iteration.
                sb = self._static_inputs(rec, depth, K, g1, g2)
                stkey = ("static",) + key[4:]
                gr = rec["static"]
                if gr is None or gr["key"] != stkey:
                    gr = rec["static"] = self._capture(stkey, sb["depth"], sb["K"], sb["g1"], sb["g2"], G, h, w, ep_l, lm_l, n)
            else:
                rec["captures"] += 1
                gr = rec["gr"] = self._capture(key, depth, K, g1, g2, G, h, w, ep_l, lm_l, n)
        if gr is None:                         # capture refused: eager launches from now on (use_graph is off)
            return unpack(*self._loop(depth, K, g1, g2, G, h, w, ep_l, lm_l, n))
        gr["G"].copy_(G.reshape(-1, 4, 4))
        main = torch.cuda.current_stream()
        timing = self.loop_timing is not None          # measurement hook (tools/loop_overlap.py): when does each half run?
        fork = torch.cuda.Event(enable_timing=timing)
        fork.record(main)
        marks = [fork]
        for graph, st in gr["graphs"]:          # one linear graph per batch half, replayed on its own stream
            if st is not main:
                st.wait_event(fork)
            with torch.cuda.stream(st):
                if timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    marks.append(e0)
                graph.replay()
                j = torch.cuda.Event(enable_timing=timing)
                j.record(st)
                marks.append(j)
            if st is not main:
                main.wait_event(j)
        if timing:
            self.loop_timing.append(marks)
        # the output buffers belong to the graph record and are overwritten by the next replay: the caller gets its own
        # copies (the reference returns distinct tensors per iteration): two device copies per OUTER iteration
        return unpack(gr["bufs"]["big"].clone(), gr["bufs"]["small"].clone())

    def _static_inputs(self, rec, depth, K, g1, g2):
        """Persistent copies (per shape record) of the per-outer-iteration inputs of the inner graph; refreshed when the sources change."""
        shapes = (tuple(depth.shape), tuple(K.shape), tuple(g1.shape), tuple(g2.shape))
        sb = rec["sbuf"]
        if sb is None or sb["shapes"] != shapes:
            sb = rec["sbuf"] = dict(shapes=shapes, depth=torch.empty_like(depth), K=torch.empty_like(K),
                                    g1=torch.empty_like(g1), g2=torch.empty_like(g2), src=None)
            rec["static"] = None
        src = (depth.data_ptr(), K.data_ptr(), g1.data_ptr(), g2.data_ptr(), depth._version, K._version, g1._version, g2._version)
        if sb["src"] != src:
            sb["depth"].copy_(depth); sb["K"].copy_(K); sb["g1"].copy_(g1); sb["g2"].copy_(g2)
            sb["src"] = src
        return sb

    def _capture(self, key, depth, K, g1, g2, G, h, w, ep_l, lm_l, n):
        """-> graph record, or None when capture is refused (the caller then runs eager launches).  One hipGraph per batch
        half, each a LINEAR chain captured on the stream it will be replayed on: concurrency between the halves comes from
        the two streams, not from branches inside a graph (a two-branch graph of this length replayed almost serially on
        ROCm 7.2: r02 timeline, 92 % of the time one kernel in flight).  The warm-up launches advance the GRU hidden
        state; it is restored whatever happens, so an eager retry starts from the right state."""
        eng = self.cf_net.engine()
        snap = eng.state_snapshot()
        B, dev = depth.shape[0], depth.device
        try:
            Gs = G.reshape(-1, 4, 4).clone()
            main = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):                  # warm-up on a side stream: weight packing, allocator, caches
                self._loop(depth, K, g1, g2, Gs, h, w, ep_l, lm_l, n)
            main.wait_stream(side)
            eng.state_restore(snap)
            bufs = self._loop_buffers(B, depth.shape[-2], depth.shape[-1], h, w, n, dev)
            hv = eng.halves(B)
            graphs = []
            torch.cuda.synchronize()
            for k, (b0, b1) in enumerate(hv):
                # both halves replay on pool streams, never on the caller's (default) stream: a graph launched into the
                # default stream did not overlap with the other half's graph at all (r02: HIP-event timing of the replays)
                st = eng._stream(dev, k) if len(hv) > 1 else main
                graph = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream()                  # (torch captures on a side stream of its own and replays on the current one)
                with torch.cuda.graph(graph, stream=cap):
                    for _ in self._half_loop(bufs, b0, b1, cap, depth, K, g1, g2, Gs, h, w, ep_l, lm_l, n, len(hv) == 1):
                        pass
                graphs.append((graph, st))
            return dict(key=key, graphs=graphs, G=Gs, bufs=bufs)
        except Exception as e:                             # noqa: BLE001 -- any capture failure means "run eagerly"
            import warnings
            warnings.warn(f"hipGraph capture of the refinement iterations failed ({e!r}); running eager launches")
            self.use_graph = False
            self._shapes = {}
            return None
        finally:
            eng.state_restore(snap)

    @torch.no_grad()
    def forward(self, image, Ts, intrinsics, fea_3d=None, Tj_gt=None, obj_cls=None, geofea_3d=None, geofea_2d=None):
        """image (B,3,H0,W0); Ts SE3Sequence (B,1,4,4); intrinsics (B,3,3) -> dict (PoseRefiner.py:366-376)."""
        self._clear()
        if image is not None and image.is_cuda or intrinsics.is_cuda:
            from .streams import reserve
            reserve(intrinsics.device)        # bind the concurrent streams to distinct hardware queues before anything else
            ops.range_guard_arm(intrinsics.device)
        self._refresh()
        cfg = self.cfg
        lm_l, ep_l = cfg.get("LM_LMBDA", LM_LMBDA), cfg.get("EP_LMBDA", EP_LMBDA)
        Tij_gt, syn_imgs, syn_depths = [], [], []
        Ti = Ts
        Tij = Ti.copy().identity()
        corr_weight = flow_up = None
        views = None
        for ren_iter in range(cfg.RENDER_ITER_COUNT):
            if ren_iter == 1 and self.profile_rec is not None:          # measurement hook: only the FIRST outer iteration is
                ops.profile_end(self.profile_rec)                       # event-instrumented (eager), the others replay graphs
                self.profile_rec = None
            fused_pose = self.fused and Ti.G.is_cuda and self.legacy and Ti.G.shape == Tij.G.shape and os.environ.get("RNNPOSE_FUSED_POSE", "1") != "0"
            if fused_pose:       # r06: :241-244 as ONE launch (ops.se3_outer_update: bit-identical to the three it replaces)
                Ti_G, Tij_G = ops.se3_outer_update(Tij.G, Ti.G, self.literal_legacy_pose)
                Ti = Ti.__class__(matrix=Ti_G, internal=Ti.internal)
            else:
                Ti = Tij * Ti                                           # accumulate (PoseRefiner.py:241)
            # the reference calls Tij.identity_() here, which also resets the object it stored in
            # residual_pose_history one line of bookkeeping earlier (same Python object); a fresh object keeps
            # the history intact
            Tij = Tij.__class__(matrix=Tij_G, internal=Tij.internal, eq=Tij.eq) if fused_pose else Tij.identity()
            # The reference's legacy branch forms Tij = Ti * Ti.inv() here (:243-244): the identity up to fp32 rounding of ITS
            # inverse / product (~1e-7, BLAS-dependent).  That noise is not reproducible between implementations, and it is not
            # harmless: it moves the first lookup ~1e-6..1e-5 px off the integer grid, which a correlation surface of
            # un-normalised features (|corr| ~ 900, gradients of hundreds per pixel at the bench shape) turns into 1e-4-level
            # flow differences -- r03 found it to be the whole first-iteration distance of the timed configuration to the oracle
            # (tools/bench_parity_probe.py; the distance did not move by one bit under any change of the GPU arithmetic).  r03-r05
            # therefore shipped the EXACT identity.  r06: the reference-generated fixtures (encoder in the loop, 480 x 640 at three
            # gains, 960 x 1280) show the literal product CLOSER to the reference's own outputs than the identity on three of four
            # (profiles/r05_fixture_distances.txt), so the product is the default (two 4 x 4 kernels per outer iteration) and the
            # oracle's default follows; `literal_legacy_pose=False` keeps the identity as an option.
            if self.legacy and self.literal_legacy_pose and not fused_pose:
                Tij = Ti * Ti.inv()
            views = self.renderer.render_views(Ti.matrix().squeeze(1), intrinsics, obj_cls=obj_cls, image=image,
                                               fea_3d=fea_3d, geofea_3d=geofea_3d, geofea_2d=geofea_2d)
            syn_depth = views["syn_depth"]
            intrinsics_crop = views["intrinsics_crop"]
            cfea_crop = views["cfea"]
            geofea1_crop, geofea2_crop = views["geofea1"], views["geofea2_crop"]
            syn_imgs += [views["syn_img"], views["image_crop"]]
            if self.fused:       # encoder + volume / pyramid + context prep: one (replayable) unit per outer iteration
                feats1, feats2 = self._outer(views)
            elif views.get("fmap1") is not None:
                feats1, feats2 = views["fmap1"], views["fmap2"]
            else:
                feats1, feats2 = self.image_fea_enc(views["syn_img"], views["image_crop"])   # (:311)
            B, _, H, W = syn_depth.shape
            h, w = feats1.shape[-2:]
            if (H, W) != (8 * h, 8 * w):
                raise ValueError(f"view size {H}x{W} must be 8x the feature-map size {h}x{w} (multiples of 8, CFNet.py:86-93)")
            use_w = self.with_corr_weight and geofea1_crop is not None and geofea2_crop is not None
            if not use_w:
                raise NotImplementedError("with_corr_weight=False has no defined weight in the reference (:347)")
            coords0 = coords_grid_lowres(B, h, w, device=syn_depth.device)

            loop = None
            if self.fused:      # all ITER_COUNT inner iterations of this outer iteration: one (replayable) unit
                loop = self._inner_loop(syn_depth, intrinsics_crop, geofea1_crop, geofea2_crop, Tij.G, h, w, ep_l, lm_l,
                                        cfg.ITER_COUNT)
            for i in range(cfg.ITER_COUNT):
                self.intrinsics_history.append(intrinsics_crop)
                syn_depths.append(syn_depth)
                Tij = Tij.copy(stop_gradients=True)
                if self.fused:
                    flow_up, wmap, G, Hm, bv, xi, info = loop[i]
                    flow = [flow_up]
                    Tij = SE3Sequence(matrix=G.reshape(B, 1, 4, 4))
                    Tij.last_info, Tij.last_system = info, (Hm, bv, xi)
                    corr_weight = wmap[:, None, :, :, None]
                else:
                    depths = syn_depth + EPS                                                      # (:313)
                    reproj, vmask = Tij.transform(depths, intrinsics_crop, valid_mask=True)       # (:324)
                    grids = coords_grid(depths)
                    flow_init = (reproj - grids[..., :2]).permute(0, 1, 4, 2, 3) * (depths > EPS)[:, :, None]
                    flow = self.cf_net(feats1, feats2, flow_init=flow_init.squeeze(1), context_fea=cfea_crop,
                                       update_corr_fn=(i == 0))                                  # (:329)
                    flow_up = flow[-1]
                    target = flow_up.permute(0, 2, 3, 1)[:, None] + grids[..., :2]                # (:336)
                    wmap = ops.corr_weight(geofea1_crop, geofea2_crop, target.squeeze(1), syn_depth, self.sigma[0])
                    corr_weight = wmap[:, None, :, :, None]
                    Tij = Tij.reprojction_optim(target, corr_weight, depths, intrinsics_crop,
                                                num_iters=cfg.OPTIM_ITER_COUNT, lm_lmbda=lm_l, ep_lmbda=ep_l)
                self.flow_history.append(flow)
                self.residual_pose_history.append(Tij)
                self.Ti_history.append(Ti.copy(stop_gradients=True))
                if Tj_gt is not None:
                    Tij_gt.append((Tj_gt * Ti.inv()).copy(stop_gradients=True))

        Ti = Tij * Ti                                                   # final update (:365)
        return {
            "Tij": Tij,
            "Ti_pred": Ti,
            "intrinsics": intrinsics,
            "flow": self.flow_history[0],
            "flow_last": flow_up,
            "vmask": views["syn_depth"] > 0,
            "weight": corr_weight.permute(0, 1, 4, 2, 3),
            "syn_depth": syn_depths,
            "syn_img": syn_imgs,
            "Tij_gt": Tij_gt,
            # sticky count of activation quads the fp16x3 split had to clamp so far (device tensor, no host sync): nonzero
            # means some value left the +-8188 range the fp32 reference would have handled
            "f16x3_range_events": ops.saturation_events(),
        }
