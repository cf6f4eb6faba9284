"""Fused NHWC execution of one GRU correspondence update (model/CFNet.py:147-168 around
thirdparty/raft/update.py:178-188): every convolution of BasicUpdateBlock runs in the hand-written implicit-GEMM
kernel (csrc/conv_igemm.hip) with bias / ReLU / GRU gates fused into its epilogue, reading the concatenations
[h | inp | motion] and [cor | flo] as virtual concats and writing straight into channel slices.

Per step and batch part: lookup (NHWC) -> convc1 -> convc2 | flow features (coords - grid + 7x7 convf1 in one direct
kernel) -> convf2 -> conv -> z|r (1x5) -> q (1x5) -> z|r (5x1) -> q (5x1)  [the context input's share of the GRU gates is
hoisted: computed once per outer iteration and added in the epilogues] -> flow/mask heads (one 128->512 conv) ->
flow_head.conv2 + coords update -> mask.2 + convex up-sampling in ONE kernel (csrc/mask_upsample.hip; the 576-channel
mask only exists for the BasicUpdateBlock facade): 14 launches, no ATen elementwise / cat / clone kernels.  By default the whole
batch is cut into two half-batch chains on two streams when every chain still fills the chip (RNNPOSE_SPLIT_BATCH=0: ONE chain on
the caller's stream; see __init__).  The module's parameters stay the single source of truth: packed fp16 hi/lo copies are rebuilt whenever a parameter's
version or storage changes.
"""
from __future__ import annotations

import torch

from . import ops


class UpdateEngine:
    def __init__(self, update_block):
        self.blk = update_block
        self._key = None
        self._w = None
        self._buf_key = None
        self._b = None
        self._sets = {}           # (B,h,w,device) -> activation buffer set; kept alive because captured hipGraphs hold raw
        self.epoch = 0            # pointers into them.  Bumped whenever a set is freed: graph caches keyed on it are dropped
        import os
        # Two half-batch chains on two streams (RNNPOSE_SPLIT_BATCH, default 1 again since r05; 0: ONE full-batch chain on the caller's
        # stream).  Images are independent, so each half of the batch runs its complete inner loop on its own stream: one chain's
        # latency-bound tail kernels and kernel boundaries run under the other chain's convolutions (+2 % on the headline step next to
        # two encoder streams, +4-5 % for both together: profiles/r05_schedules.txt).  r04 had switched this off because fresh instances
        # differed in 64-byte runs of a weight map whenever two streams were active.  r05 found the cause (profiles/r05_determinism.txt):
        # corr_weight, compiled into PACKED fp32 instructions by plain -O3, computed other values for groups of 16 lanes while a
        # v_mfma_f32_16x16x32_f16 kernel (mask_upsample, conv1x1_resident) of the other stream shared its SIMDs -- a property of the chip
        # that a plain-HIP probe reproduces (tools/probes/pk_f32_vs_mfma.hip), not a visibility problem.  The library is built without
        # packed fp32 since (build.py, tests/test_isa_guard.py) and every schedule is bit-reproducible and bit-identical to the others
        # (tests/test_gpu_reproducibility.py).
        self.split_batch = os.environ.get("RNNPOSE_SPLIT_BATCH", "1") != "0"
        # flow-feature / flow-head side chain of a lone SMALL chain on a helper stream: off by default since r05 (B = 1, 240 x 240:
        # 4.25 ms per refinement without it, 4.49 with it -- hipGraph replay runs the two-branch graph no faster than the linear one)
        self.side_stream = os.environ.get("RNNPOSE_SIDE_STREAM", "0") != "0"
        self.parts = int(os.environ.get("RNNPOSE_PARTS", "2"))
        self.parts_forced = "RNNPOSE_PARTS" in os.environ                         # explicit part count: no small-batch merging
        self.fused_mask = os.environ.get("RNNPOSE_FUSED_MASK", "1") != "0"       # mask.2 inside the up-sampling kernel
        self.resident_1x1 = os.environ.get("RNNPOSE_RESIDENT_1X1", "1") != "0"    # convc1 through csrc/conv1x1_resident.hip
        # r06: window lookup + convc1 as ONE launch (csrc/corr_convc1.hip: the 324-channel corr tensor is never written).  Built, parity-
        # tested and NOT faster: alone 39.8 us against 23.1 + 18.7 us for the two kernels at B = 4 (67 vs 70 at B = 8), on the step 754 vs 756
        # iters/s (profiles/r06_lookup_convc1_fusion.txt) -- the lookup is bound by the sectors its gather touches (81 MB per launch), and a
        # one-round launch runs its footprint phases and its MFMA phases in lock step, so the fusion saves the 25-MB round trip of the
        # tensor and nothing else.  OFF by default; RNNPOSE_FUSED_LOOKUP=1 switches it on.
        self.fused_lookup = os.environ.get("RNNPOSE_FUSED_LOOKUP", "0") != "0"
        # r06: RNNPOSE_FUSED_INDUCED=1 forms the pose-induced coordinates inside their first consumers (window lookup, flow-feature convolution:
        # ops.InducedCoords) instead of by a 5-us launch of their own in front of every iteration -- bit-identical values (csrc/induced.cuh), one
        # launch fewer per iteration, and measured EQUAL on the step (750.6 vs 749.5 iters/s, S1 4.08 vs 4.08 ms: what the launch cost, the two
        # consumers now spend on 4 depth taps per pixel and level / per patch entry; profiles/r06_fused_induced.txt): an option, off by default
        self.fused_induced = os.environ.get("RNNPOSE_FUSED_INDUCED", "0") != "0"
        self.ksplit = os.environ.get("RNNPOSE_KSPLIT", "1") != "0"                # small launches (B = 1 crops) split their K loop
        # SPLIT TENSORS (include/rnnpose_hip.h): every activation that only feeds further convolutions is written once, by its
        # producer's epilogue, as fp16 hi|lo pairs and staged by the consumers with a plain 16-byte copy (no per-tile re-split).
        # r03 (128-row kernels, which re-stage through registers anyway): -1.3 % on the step, off.  r04: the strip kernels take split
        # tensors by LDS-DMA (no registers, no split instructions in the loop) and their specialised epilogue writes the split form
        # at 4-6 us per workgroup: +2-3 % on the step (profiles/r04_ab_split_tensors.txt, same box: 724 vs 690-713 iters/s).  ON by
        # default; RNNPOSE_SPLIT_TENSORS=0 restores fp32 activations everywhere (tests cover both).
        self.hl = os.environ.get("RNNPOSE_SPLIT_TENSORS", "1") != "0"
        # cfg.raft.mixed_precision: one fp16 product per multiply-add in the 160-row strip kernels.  Carried by the ENGINE its PoseRefiner
        # configures (r04 used a process-wide default that direct callers of ops / cf_net inherited from the last refiner: ADVICE r04)
        self.single_product = False
        # tile-shape override per layer for measurements: RNNPOSE_CONV_TILE="zr=3,q=1,heads=2" (0 auto, see conv2d_nhwc)
        self.tile = {}
        for kv in filter(None, os.environ.get("RNNPOSE_CONV_TILE", "").split(",")):
            k, v = kv.split("=")
            self.tile[k] = int(v)

    # ---- weights ---------------------------------------------------------------------------------------------
    def _params(self):
        b = self.blk
        e, g = b.encoder, b.gru
        return [e.convc1, e.convc2, e.convf1, e.convf2, e.conv, g.convz1, g.convr1, g.convq1, g.convz2, g.convr2,
                g.convq2, b.flow_head.conv1, b.flow_head.conv2, b.mask[0], b.mask[2]]

    def param_key(self):
        """Identity of the live parameters: changes on load_state_dict / in-place updates / .to()."""
        return tuple((m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()) for m in self._params())

    def refresh(self):
        """Re-pack the fp16 hi/lo weights if the module's parameters changed; -> the key captured graphs depend on."""
        self._weights()
        return self._key

    def _weights(self):
        key = self.param_key()
        if key == self._key:
            return self._w
        b = self.blk
        e, g = b.encoder, b.gru
        cat = lambda *t: torch.cat([x.detach() for x in t], 0)
        hm = lambda wt: torch.cat([wt[:, :128], wt[:, 256:]], 1).contiguous()
        zero = lambda n: torch.zeros(n, device=g.convz1.weight.device, dtype=torch.float32)
        P = ops.PackedConv
        w = dict(
            convc1=P(e.convc1.weight, e.convc1.bias, [e.convc1.weight.shape[1]]),
            convc1r=(ops.PackedConv1x1(e.convc1.weight, e.convc1.bias)
                     if (e.convc1.weight.shape[0] == 256 and e.convc1.weight.shape[1] <= 352 and e.convc1.weight.shape[1] % 4 == 0) else None),
            convc2=P(e.convc2.weight, e.convc2.bias, [256]),
            convf1_wt=e.convf1.weight.detach().float().reshape(e.convf1.weight.shape[0], 98).t().contiguous(),   # (98, 128)
            convf1_b=e.convf1.bias.detach().float().contiguous(),
            convf2=P(e.convf2.weight, e.convf2.bias, [128]),
            conv=P(e.conv.weight, e.conv.bias, [256]),
            # GRU input = [h | inp | motion] (update.py:181, :47-58).  `inp` (the context half) is constant over the inner
            # iterations, so its share of every GRU convolution is computed ONCE per outer iteration (load_state) into
            # per-pixel bias maps: conv([h|inp|m]) = conv([h|m]) + conv(inp) -- one third of the GRU's multiply-adds leave
            # the loop.  hm(): the weights on [h | motion]; inp1 / inp2: z|r|q weights on `inp` for the 1x5 / 5x1 halves.
            zr=P(hm(cat(g.convz1.weight, g.convr1.weight)), cat(g.convz1.bias, g.convr1.bias), [128, 128]),
            q=P(hm(g.convq1.weight), g.convq1.bias, [128, 128]),
            zr2=P(hm(cat(g.convz2.weight, g.convr2.weight)), cat(g.convz2.bias, g.convr2.bias), [128, 128]),
            q2=P(hm(g.convq2.weight), g.convq2.bias, [128, 128]),
            inp1=P(cat(g.convz1.weight, g.convr1.weight, g.convq1.weight)[:, 128:256].contiguous(), zero(384), [128]),
            inp2=P(cat(g.convz2.weight, g.convr2.weight, g.convq2.weight)[:, 128:256].contiguous(), zero(384), [128]),
            heads=P(cat(b.flow_head.conv1.weight, b.mask[0].weight), cat(b.flow_head.conv1.bias, b.mask[0].bias), [128]),
            mask2=P(b.mask[2].weight, b.mask[2].bias, [256], post_scale=0.25),            # update.py:187
            mask2u=ops.PackedMaskHead(b.mask[2].weight, b.mask[2].bias, post_scale=0.25),  # the same layer inside the up-sampling kernel
            flow2_w=b.flow_head.conv2.weight.detach().float().contiguous(),
            flow2_b=b.flow_head.conv2.bias.detach().float().contiguous(),
        )
        self._key, self._w = key, w
        return w

    # ---- activations -----------------------------------------------------------------------------------------
    MAX_SETS = 4

    def _buffers(self, B, h, w, device):
        """Select (allocating on first use) the activation buffers of shape (B,h,w).  One set per shape stays alive --
        a partial last eval batch followed by a full one must find the full batch's buffers (and the graphs captured
        over them) intact; beyond MAX_SETS shapes the oldest set is dropped and `epoch` invalidates every graph."""
        key = (B, h, w, str(device))
        if key != self._buf_key:
            st = self._sets.pop(key, None)
            if st is None:
                z = lambda c: torch.zeros(B, h, w, c, device=device, dtype=torch.float32)
                st = dict(corr=z(324), cor1=z(256), corflo=z(256), flow4=z(4), flo1=z(128), motion=z(128), hA=z(128),
                          hB=z(128), inp=z(128), inp1=z(384), inp2=z(384), z=z(128), rh=z(128), heads=z(512), delta=z(2), flow_lr=z(2),
                          mask=z(576), coords1=torch.zeros(B, 2, h, w, device=device))
                if self.hl:       # split-form copies of the tensors that are needed in fp32 as well (GRU epilogues read h)
                    st.update(hA_s=z(128), hB_s=z(128), inp_s=z(128))
                if len(self._sets) >= self.MAX_SETS:
                    self._sets.pop(next(iter(self._sets)))
                    self.epoch += 1
            self._sets[key] = st          # (re-)insert as most recent
            self._b, self._buf_key = st, key
        return self._b

    def buffer_key(self):
        return (self._buf_key, self.epoch)

    def _ksplit_ws(self, b0, b1, which):
        """K-split workspace (ops.conv_ksplit_workspace) of the chain of images [b0, b1) -- `which` 0: its main stream, 1: its
        helper stream (launches that can run concurrently must not share one).  None when the chain has too many pixels for
        any of its convolutions to split (the library splits launches of <= 96 tiles).  Allocated on first use, i.e. in the
        eager warm-up runs before a hipGraph capture."""
        B, h, w, dev = self._buf_key
        if not self.ksplit or (b1 - b0) * h * w > 96 * 128:
            return None
        reg = self.__dict__.setdefault("_ksws", {})
        key = (self._buf_key, self.epoch, b0, which)
        ws = reg.get(key)
        if ws is None:
            for k in [k for k in reg if k[1] != self.epoch]:
                del reg[k]
            ws = reg[key] = ops.conv_ksplit_workspace(self._b["hA"].device)
        return ws

    def select(self, buf_key):
        """Make the buffer set a captured graph was recorded on the current one (graph replay path)."""
        (B, h, w, dev), _ = buf_key
        return self._buffers(B, h, w, dev)

    def _stream(self, device, i):
        """Helper stream i (0, 1: the two batch halves; 2: the flow-feature / flow-head side chain of an unsplit step) --
        from the per-device set that is bound to distinct hardware queues (rnnpose_amd/streams.py)."""
        from .streams import reserve
        if ops.profiling():
            # per-launch timing (ops.profile): everything on the caller's stream, so that each launch is timed ALONE on the
            # chip -- the duration a roofline wants (and what rocprofv3's kernel trace, which serialises kernels, reports).
            # Timed next to its twin on the other stream, a launch's duration includes the sharing.
            return torch.cuda.current_stream()
        ss = reserve(device)
        return (ss.chain + [ss.aux] + ss.extra)[i]

    def load_state(self, net, inp):
        """net, inp: (B,128,h,w) NCHW (tanh / relu of the context features, model/CFNet.py:131-133)."""
        B, _, h, w = net.shape
        b = self._buffers(B, h, w, net.device)
        ops.nchw_to_nhwc(net, b["hA"])
        ops.nchw_to_nhwc(inp, b["inp"])
        self._hoist_inp(self._weights(), b)

    def _hoist_inp(self, W, b):
        """The `inp` share of the six GRU convolutions (z|r|q x two halves), once per hidden-state load (+ the split forms of
        the freshly loaded hidden state and context input)."""
        src = b["inp"]
        if self.hl:
            ops.split_hl(b["hA"], b["hA_s"])
            src = ops.split_hl(b["inp"], b["inp_s"])
        ops.conv2d_nhwc(W["inp1"], [(src, 0)], (b["inp1"], 0), ops.EPI_LINEAR, src_hl=self.hl, tile=self.tile.get("inp", 0), single_product=self.single_product)
        ops.conv2d_nhwc(W["inp2"], [(src, 0)], (b["inp2"], 0), ops.EPI_LINEAR, src_hl=self.hl, tile=self.tile.get("inp", 0), single_product=self.single_product)

    def hidden_nchw(self):
        return ops.nhwc_to_nchw(self._b["hA"])

    def state_snapshot(self):
        """Copies of the recurrent state (graph capture warms up by running the loop, then restores it)."""
        return {k: self._b[k].clone() for k in ("hA", "hA_s") if k in self._b}

    def state_restore(self, snap):
        for k, v in snap.items():
            self._b[k].copy_(v)

    def forward_nchw(self, net, inp, corr, flow):
        """The reference boundary `BasicUpdateBlock.forward(net, inp, corr, flow) -> (net, mask, delta_flow)`
        (thirdparty/raft/update.py:178-188), all NCHW: layout transposes around the fused NHWC chain."""
        B, _, h, w = net.shape
        b = self._buffers(B, h, w, net.device)
        W = self._weights()
        ops.nchw_to_nhwc(net, b["hA"])
        ops.nchw_to_nhwc(inp, b["inp"])
        ops.nchw_to_nhwc(corr, b["corr"])
        self._hoist_inp(W, b)
        flow = flow.float().contiguous()
        self._chain(W, b, flow, torch.cuda.current_stream(), None, flow_is_delta=True, want_mask=True)
        return (ops.nhwc_to_nchw(b["hA"]), ops.nhwc_to_nchw(b["mask"]), ops.nhwc_to_nchw(b["delta"]))

    MIN_CHAIN_PIXELS = 8192     # 1/8-resolution pixels per chain below which the batch stays ONE chain

    def halves(self, B):
        """Image ranges of the concurrent chains: `parts` parts of the batch (one stream each), or the whole batch.
        A chain needs enough pixels to fill the chip on its own launches: B=16 at 240x240 (14400 pixels at 1/8 resolution) runs
        3 % faster as one chain than as two of 7200 (1009 vs 978 iters/s), B=32 (two of 14400) 3 % faster as two, the headline
        shape (two of 19200) 5 % faster as two (r02) -- on the 128-row kernels; with the strip kernels (r04-r05): 2-3.5 % at the
        headline.  RNNPOSE_SPLIT_BATCH=0: always one chain; RNNPOSE_PARTS=n: exactly n (see __init__)."""
        n = 1 if (B < 2 or not (self.split_batch or self.parts_forced)) else min(self.parts, B)
        if n > 1 and not self.parts_forced and self._buf_key is not None:
            _, h, w, _ = self._buf_key
            while n > 1 and B * h * w < n * self.MIN_CHAIN_PIXELS:
                n -= 1
        cuts = [B * i // n for i in range(n + 1)]
        return list(zip(cuts[:-1], cuts[1:]))

    def half_gen(self, corr_fn, coords1_part, B, b0, b1, st, flow_up_part, single=False, induced=None):
        """One GRU step of images [b0, b1) on stream `st` (current): window lookup -> update block -> convex up-sampling.
        coords1_part (b1-b0,2,h,w); flow_up_part (b1-b0,2,8h,8w) is written.  Generator: yields after every launch so
        that a caller can issue several chains alternately.  single: this is the only chain (B = 1): the flow-feature /
        flow-head side chain then runs on a helper stream (the only concurrency there is; nested forks inside two
        concurrent chains segfault hipStreamEndCapture on ROCm 7.2).
        induced (ops.InducedCoords with out = coords1_part): coords1_part is not filled yet -- the lookup launch forms the coordinates and
        writes them there, the flow-feature launch forms them again for its patch (r06)."""
        W = self._weights()
        view = {k: v[b0:b1] for k, v in self._b.items()}
        view["_ksws"], view["_ksws_side"] = self._ksplit_ws(b0, b1, 0), self._ksplit_ws(b0, b1, 1)
        fused = (self.fused_lookup and self.resident_1x1 and W["convc1r"] is not None and W["convc1r"].c_in == 324
                 and corr_fn.num_levels == 4 and corr_fn.radius == 4)
        if fused:           # update.py:89 on corr.py:36-57 in one launch; the chain below skips its convc1
            assert induced is None
            ops.corr_lookup_convc1(W["convc1r"], corr_fn._buf, coords1_part, (view["cor1"], 0), B, b0, b1, relu=True, dst_split=self.hl)
        elif induced is not None:
            ops.corr_lookup_induced_nhwc_part(corr_fn._buf, induced, view["corr"], B, b0, b1, corr_fn.num_levels, corr_fn.radius)
        else:
            ops.corr_lookup_nhwc_part(corr_fn._buf, coords1_part, view["corr"], B, b0, b1, corr_fn.num_levels, corr_fn.radius)
        yield
        # helper stream for the flow-feature / flow-head side chain: opt-in (RNNPOSE_SIDE_STREAM=1), and only for a lone SMALL chain; it
        # measured equal at the headline (695-699 iters/s either way) and slower at B = 1 (see __init__)
        small = coords1_part.shape[0] * coords1_part.shape[2] * coords1_part.shape[3] < self.MIN_CHAIN_PIXELS
        yield from self._chain_gen(W, view, coords1_part, st, self._stream(coords1_part.device, 2) if (single and small and self.side_stream) else None,
                                   have_cor1=fused, induced=induced)
        if self.fused_mask:         # mask.2 + up-sampling in one kernel (the chain skipped its mask.2 launch)
            ops.mask_upsample(W["mask2u"], view["heads"], 256, view["flow_lr"], out=flow_up_part)
        else:
            ops.convex_upsample_nhwc(view["flow_lr"], view["mask"], out=flow_up_part)
        yield

    @staticmethod
    def run_interleaved(jobs, main):
        """jobs: [(generator, stream)].  Every generator's launches go to its stream; the generators are advanced in turn.
        Streams other than `main` start after everything already on `main` and `main` waits for them at the end: ONE fork
        and ONE join, however long the generators are -- a cross-queue dependency costs ~100 us on this GPU (r02 timeline:
        the second chain of every inner iteration started 115 us after the event it waited for), an in-queue one ~1 us."""
        fork = torch.cuda.Event()
        fork.record(main)
        for _, st in jobs:
            if st is not main:
                st.wait_event(fork)
        active = list(jobs)
        while active:
            for item in list(active):
                g, st = item
                with torch.cuda.stream(st):
                    try:
                        next(g)
                    except StopIteration:
                        active.remove(item)
        for _, st in jobs:
            if st is not main:
                j = torch.cuda.Event()
                j.record(st)
                main.wait_event(j)

    def step(self, corr_fn, coords1, tail=None, flow_up=None):
        """coords1 (B,2,h,w) -> (coords1 + delta_flow (B,2,h,w), flow_up (B,2,8h,8w)).
        tail(b0, b1, flow_up[b0:b1]) is called on the stream of each batch half right after its up-sampling.
        The two halves of the batch run as two concurrent chains: every convolution of this workload is a ONE-round kernel
        (300-1200 tiles on 768 resident slots) whose ramp-up, lock-step prologue and epilogue burst cost ~30 % against the
        multi-round steady state; two kernels in flight fill those gaps.  Bit-identical (images are independent)."""
        B, _, h, w = coords1.shape
        main = torch.cuda.current_stream()
        if flow_up is None:
            flow_up = torch.empty(B, 2, 8 * h, 8 * w, device=coords1.device, dtype=torch.float32)
        hv = self.halves(B)

        def half(b0, b1, st):
            yield from self.half_gen(corr_fn, coords1[b0:b1], B, b0, b1, st, flow_up[b0:b1], single=len(hv) == 1)
            if tail is not None:
                tail(b0, b1, flow_up[b0:b1])

        jobs = []
        for i, (b0, b1) in enumerate(hv):
            st = main if i == 0 else self._stream(coords1.device, i)
            jobs.append((half(b0, b1, st), st))
        self.run_interleaved(jobs, main)
        return self._b["coords1"], flow_up

    def _chain(self, W, b, coords1, main, side, flow_is_delta=False, want_mask=False):
        for _ in self._chain_gen(W, b, coords1, main, side, flow_is_delta, want_mask):
            pass

    def _chain_gen(self, W, b, coords1, main, side, flow_is_delta=False, want_mask=False, have_cor1=False, induced=None):
        """The update block on the (sub-)batch views `b`, issued on stream `main` (current) with `side` as helper; yields
        after every launch so that the caller can interleave two chains.
        flow_is_delta: `coords1` holds the flow itself (facade call) instead of absolute coordinates.
        want_mask: write the (B,h,w,576) mask tensor (the facade returns it); the loop's up-sampling kernel computes mask.2
        itself (ops.mask_upsample) and the chain then ends after the flow head."""
        mask_here = want_mask or not self.fused_mask
        hl = self.hl
        tl = self.tile.get
        R = ops.EPI_RELU

        ks_main = b.get("_ksws")
        ks_side = b.get("_ksws_side") if side is not None else ks_main

        def c(name, srcs, dst, epi, hl_out=False, ks=None, **kw):
            """One implicit-GEMM launch; with split tensors every source is in split form and hl_out says the result only
            feeds further convolutions (written split).  ks: K-split workspace of the stream the launch goes to."""
            ops.conv2d_nhwc(W[name], srcs, dst, epi, src_hl=hl, dst_hl=hl and hl_out, tile=tl(name, 0),
                            ksplit_ws=ks_main if ks is None else ks, single_product=self.single_product, **kw)
        # Two independent chains feed the motion encoder's last convolution (update.py:89-92): correlation features
        # (convc1 -> convc2) and flow features (flow_prep -> convf1 -> convf2).  With a helper stream the second one runs
        # there (a parallel branch when the step is captured into a hipGraph).  Same for flow_head.conv2 next to mask.2.
        def flow_chain():
            # flow = coords1 - grid -> motion[126:128] (:97) and relu(convf1(flow)) (:91; direct fp32 kernel, K = 98): one launch
            if induced is not None:       # (r06: coords1 formed inside the launch; the lookup launch in front of the chain wrote the tensor)
                ops.flow_features_induced(induced, W["convf1_wt"], W["convf1_b"], b["flo1"], b["motion"], 126, out_split=hl, motion_split=hl,
                                          a_scale=ops.A_SCALE)
            else:
                ops.flow_features(coords1, W["convf1_wt"], W["convf1_b"], b["flo1"], b["motion"], 126, subtract_grid=not flow_is_delta,
                                  out_split=hl, motion_split=hl, a_scale=ops.A_SCALE)
            yield
            c("convf2", [(b["flo1"], 0)], (b["corflo"], 192), R, hl_out=True, ks=ks_side)       # :92
            yield

        join = None
        if side is None:
            yield from flow_chain()
        else:
            fork = torch.cuda.Event()
            fork.record(main)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                for _ in flow_chain():
                    pass
                join = torch.cuda.Event()
                join.record(side)
        if have_cor1:                                                               # (r06: convc1 ran inside the lookup launch)
            pass
        elif self.resident_1x1 and W["convc1r"] is not None:
            ops.conv1x1_resident(W["convc1r"], (b["corr"], 0), (b["cor1"], 0), relu=True, dst_split=hl)   # update.py:89 (LDS-resident tile)
            yield
        else:                                                                       # (the looked-up correlation features are fp32)
            ops.conv2d_nhwc(W["convc1"], [(b["corr"], 0)], (b["cor1"], 0), R, dst_hl=hl, single_product=self.single_product)   # update.py:89
            yield
        c("convc2", [(b["cor1"], 0)], (b["corflo"], 0), R, hl_out=True)             # :90
        yield
        if join is not None:
            main.wait_event(join)
        c("conv", [(b["corflo"], 0)], (b["motion"], 0), R, hl_out=True)             # :95-96 (126 ch; flow already at 126)
        yield
        # [h | motion]; the `inp` share comes from the hoisted maps.  Split tensors: the convolutions read h from its split copy
        # (hA_s / hB_s), the gate epilogues read the fp32 one; r*h and the motion features exist in split form only.
        hs = (lambda k: b[k + "_s"]) if hl else (lambda k: b[k])
        hx = lambda hbuf: [(hbuf, 0), (b["motion"], 0)]
        sp = (lambda k: dict(dst_split=(b[k + "_s"], 0))) if hl else (lambda k: {})
        c("zr", hx(hs("hA")), (b["z"], 0), ops.EPI_GRU_ZR, aux0=(b["hA"], 0), dst2=(b["rh"], 0), dst2_hl=hl, gru_c=128, add_map=(b["inp1"], 0))
        yield
        c("q", hx(b["rh"]), (b["hB"], 0), ops.EPI_GRU_Q, aux0=(b["hA"], 0), aux1=(b["z"], 0), add_map=(b["inp1"], 256), **sp("hB"))
        yield
        c("zr2", hx(hs("hB")), (b["z"], 0), ops.EPI_GRU_ZR, aux0=(b["hB"], 0), dst2=(b["rh"], 0), dst2_hl=hl, gru_c=128, add_map=(b["inp2"], 0))
        yield
        c("q2", hx(b["rh"]), (b["hA"], 0), ops.EPI_GRU_Q, aux0=(b["hB"], 0), aux1=(b["z"], 0), add_map=(b["inp2"], 256), **sp("hA"))
        yield
        c("heads", [(hs("hA"), 0)], (b["heads"], 0), R)                             # flow_head.conv1 | mask.0
        yield
        head = lambda: ops.flow_head_out(b["heads"], 0, 256, W["flow2_w"], W["flow2_b"], coords1, b["delta"], b["coords1"],
                                         b["flow_lr"])
        if side is None:
            head()
            yield
            if mask_here:
                ops.conv2d_nhwc(W["mask2"], [(b["heads"], 256)], (b["mask"], 0), ops.EPI_LINEAR, single_product=self.single_product)  # 0.25 * mask.2(relu(mask.0(h)))
                yield
            return
        fork2 = torch.cuda.Event()
        fork2.record(main)
        side.wait_event(fork2)
        with torch.cuda.stream(side):
            head()
            join2 = torch.cuda.Event()
            join2.record(side)
        if mask_here:
            ops.conv2d_nhwc(W["mask2"], [(b["heads"], 256)], (b["mask"], 0), ops.EPI_LINEAR, single_product=self.single_product)      # 0.25 * mask.2(relu(mask.0(h)))
        main.wait_event(join2)
        yield


class EncoderEngine:
    """RAFT BasicEncoder (instance-norm variant, thirdparty/raft/extractor.py:118-232) executed NHWC: the stem (input
    normalisation + 7x7 stride-2 convolution on 3 channels) is one im2col-in-LDS MFMA kernel (csrc/stem.hip), all 3x3
    convolutions and the 1x1 output convolution run in the implicit-GEMM kernel, instance norm / ReLU / residual adds in
    one fused HIP pass each; the stride-2 3x3 / 1x1 convolutions use the kernel's strided mode.  No MIOpen, no ATen math."""

    def __init__(self, fnet):
        import os
        self.fnet = fnet
        self._key = None
        self._w = None
        self._side = None
        self.resident_1x1 = os.environ.get("RNNPOSE_RESIDENT_1X1", "1") != "0"
        self.ksplit = os.environ.get("RNNPOSE_KSPLIT", "1") != "0"
        self.single_product = False                   # cfg.raft.mixed_precision, set by the PoseRefiner that owns the encoder (see UpdateEngine)
        self.split_batch = os.environ.get("RNNPOSE_SPLIT_ENCODER", "1") != "0"
        self.parts = int(os.environ.get("RNNPOSE_ENCODER_PARTS", "1"))
        # one stream per image set (rendered | observed) -- the r02-r03 schedule, the default again since r05 for image sets that fill
        # the chip: one set's HBM-bound normalisation passes and latency-bound finalize launches run under the other set's convolutions
        # (+1-2 % on the headline step, +2 % at B = 16 240 x 240).  Small sets run as ONE concatenated batch on one stream: the fork / join
        # and half-empty launches cost more than the overlap gives (B = 1 240 x 240: 4.19 vs 4.53 ms per refinement, B = 4: 5.33 vs 5.55;
        # B = 2 480 x 640: equal; profiles/r05_ab_encoder_streams_small.txt).  r04 had merged ALL sets because runs with two active
        # streams were not bit-reproducible; r05 found and removed the cause (UpdateEngine.__init__).
        # RNNPOSE_ENCODER_MERGE=1 / 0: always one batch / always one stream per set.
        m = os.environ.get("RNNPOSE_ENCODER_MERGE")
        self.merge_sets = None if m is None else (m != "0")

    MIN_SET_PIXELS = 800_000      # full-resolution pixels per image set from which the sets get their own streams

    def _mine(self):
        f = self.fnet
        convs = {}
        for li, layer in enumerate((f.layer1, f.layer2, f.layer3), start=1):
            for bi, blk in enumerate(layer):
                convs[f"l{li}.{bi}.c1"] = blk.conv1
                convs[f"l{li}.{bi}.c2"] = blk.conv2
                if blk.downsample is not None:
                    convs[f"l{li}.{bi}.down"] = blk.downsample[0]
        convs["out"] = f.conv2
        return convs

    def refresh(self):
        self._weights()
        return self._key

    def _weights(self):
        convs = self._mine()
        convs["stem"] = self.fnet.conv1
        key = tuple((m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()) for m in convs.values())
        if key != self._key:
            stem = convs.pop("stem")
            self._w = {k: ops.PackedConv(m.weight, m.bias, [m.weight.shape[1]]) for k, m in convs.items()}
            self._w["stem"] = ops.PackedStem(stem.weight, stem.bias)
            oc = convs["out"]           # the 1x1 output convolution (extractor.py:155: 128 -> 256) in the LDS-resident 1x1 kernel
            ok = tuple(oc.weight.shape[2:]) == (1, 1) and oc.weight.shape[0] == 256 and oc.weight.shape[1] <= 352 and oc.weight.shape[1] % 4 == 0
            self._w["outr"] = ops.PackedConv1x1(oc.weight, oc.bias) if ok else None
            self._key = key
        return self._w

    @staticmethod
    def _conv(pc, x, stride=1, stats=True, in_norm=None, ks=None, sp=False):
        """-> (out, tile_stats or None): the convolution's epilogue also emits the per-tile column statistics the following
        instance norm needs (saves re-reading the tensor once; output rows are tiled per image for that).
        in_norm: mean / rstd of `x`, which is then a RAW convolution output normalised (+ ReLU) in this convolution's load."""
        B, H, W, _ = x.shape
        Ho, Wo = -(-H // stride), -(-W // stride)
        out = torch.empty(B, Ho, Wo, pc.c_out, device=x.device, dtype=torch.float32)
        ts = torch.empty(B * ops.conv_tiles_per_image(H, W, pc.kh, pc.kw, stride, pc.c_out, 0, B, src_counts=pc.seg_counts, fused_norm=in_norm is not None),
                         pc.c_out, 2, device=x.device, dtype=torch.float64) if stats else None
        # src_bounded: every convolution input of the encoder is an instance-normalised map (|.| <= sqrt(H*W)) or a ReLU sum of a
        # few of them (extractor.py:48-58): two orders of magnitude inside the fp16x3 range, so the in-loop range check is skipped
        # (the stem, which sees the raw image, and the update block keep theirs)
        ops.conv2d_nhwc(pc, [(x, 0)], (out, 0), ops.EPI_LINEAR, stride=stride, tile_stats=ts, in_norm=in_norm, src_bounded=True,
                        ksplit_ws=ks, single_product=sp)
        return out, ts

    @staticmethod
    def _block_gen(W, name, blk, x, x_norm=None, ks=None, sp=False):
        """ResidualBlock (extractor.py:48-58) on an NHWC tensor, as a generator (yields after every launch, returns the block
        output).  Instance norms are never materialised on their own: norm1 + ReLU happens in conv2's load, the residual's
        norm (x_norm = (mean_rstd, relu) when `x` is the RAW stem output; norm3 of the down-sampling branch) inside the one
        pass that forms relu(residual + relu(norm2(conv2 .))) -- the only tensor written per block besides the raw
        convolution outputs."""
        E = EncoderEngine
        st = blk.conv1.stride[0]
        c1, ts1 = E._conv(W[name + ".c1"], x, st, in_norm=None if x_norm is None else x_norm[0], ks=ks, sp=sp)
        yield
        mr1 = ops.instnorm_tiles_nhwc(c1, ts1, stats_only=True)          # relu(norm1(conv1 x)): applied in conv2's load
        yield
        res, res_norm = x, x_norm
        if blk.downsample is not None:
            assert x_norm is None
            res, tsd = E._conv(W[name + ".down"], x, st, ks=ks, sp=sp)
            yield
            res_norm = (ops.instnorm_tiles_nhwc(res, tsd, stats_only=True), False)            # norm3, no ReLU
            yield
        c2, ts2 = E._conv(W[name + ".c2"], c1, in_norm=mr1, ks=ks, sp=sp)
        yield
        out = ops.instnorm_tiles_nhwc(c2, ts2, relu=True, residual=res, residual_norm=None if res_norm is None else res_norm[0],
                                      residual_relu=bool(res_norm and res_norm[1]))               # relu(x + relu(IN(.)))
        yield
        return out

    @staticmethod
    def _block(W, name, blk, x):
        g = EncoderEngine._block_gen(W, name, blk, x)
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    @torch.no_grad()
    def __call__(self, images, normalize=True, split_out=False):
        """images: one (N,3,H,W) tensor or a list of them (rendered, observed: one stream each; RNNPOSE_ENCODER_MERGE=1: concatenated into one batch)
        -> (sum N, 256, H/8, W/8) NCHW.  normalize: apply 2*(x/255)-1 in the stem's load (model/CFNet.py:42-43).
        split_out: return an ops.SplitTensor instead (pixel-major fp16 hi|lo, written by the output convolution itself) --
        the operand format of the volume build, which then needs neither the NCHW transposition nor its split pre-pass."""
        W = self._weights()
        imgs = [images] if torch.is_tensor(images) else list(images)
        imgs = [ops._chk(t, "image") for t in imgs]
        merge = self.merge_sets
        if merge is None and len(imgs) > 1:          # automatic: small image sets as ONE batch on ONE stream
            merge = min(t.shape[0] * t.shape[2] * t.shape[3] for t in imgs) < self.MIN_SET_PIXELS
        if merge and len(imgs) > 1:
            imgs = [torch.cat(imgs, 0)]
        _, _, H, Wd = imgs[0].shape
        N = sum(t.shape[0] for t in imgs)
        dev = imgs[0].device
        H8, W8 = ((H + 1) // 2 + 1) // 2, ((Wd + 1) // 2 + 1) // 2
        H8, W8 = (H8 + 1) // 2, (W8 + 1) // 2
        if split_out:
            out = torch.empty(N, H8, W8, self.fnet.conv2.out_channels, device=dev, dtype=torch.float32)
        else:
            out = torch.empty(N, self.fnet.conv2.out_channels, H8, W8, device=dev, dtype=torch.float32)
        # Two halves of the image batch (rendered | observed) as two independent streams = two hipGraph branches with a
        # single join at the end: the chains drift apart, so one half's HBM-bound instance-norm passes and latency-bound
        # finalize launches run under the other half's convolutions.  Bit-identical (instance norm is per image).
        # (r03: small image sets -- 2 x 1 x 240 x 240 -- as ONE concatenated batch instead of two streams measured 1.5 % slower:
        #  5.16 vs 5.08 ms per refinement; the two launch chains do overlap in graph replay)
        if len(imgs) > 1:                      # one stream per input tensor (rendered | observed)
            jobs, o = [], 0
            for t in imgs:
                jobs.append((t, o, o + t.shape[0]))
                o += t.shape[0]
        else:
            parts = max(1, min(self.parts if self.split_batch else 1, N))
            cuts = [N * i // parts for i in range(parts + 1)]
            jobs = [(imgs[0][a:b], a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []

        def job(x_part, b0, b1, st, ji):
            if st is not main:
                st.wait_event(fork)
            yield from self._forward_gen(W, x_part, out[b0:b1], normalize, split_out, ks=self._ksplit_ws(ji, x_part))
            if st is not main:
                j = torch.cuda.Event()
                j.record(st)
                joins.append(j)

        # launches of the jobs are issued alternately (see UpdateEngine.step: hipGraphLaunch submits in capture order)
        active = []
        for hi_, (x_part, b0, b1) in enumerate(jobs):
            st = main if hi_ == 0 else self._second_stream(dev, hi_)
            active.append((job(x_part, b0, b1, st, hi_), st))
        while active:
            for item in list(active):
                g, st = item
                with torch.cuda.stream(st):
                    try:
                        next(g)
                    except StopIteration:
                        active.remove(item)
        for j in joins:
            main.wait_event(j)
        return ops.SplitTensor(out, ops.A_SCALE) if split_out else out

    def _ksplit_ws(self, ji, x_part):
        """K-split workspace of job (stream) ji, or None when even the 1/8-resolution layers have too many tiles to split."""
        n, _, H, Wd = x_part.shape
        if not self.ksplit or n * ((H + 7) // 8) * ((Wd + 7) // 8) > 96 * 128:
            return None
        reg = self.__dict__.setdefault("_ksws", {})
        key = (ji, str(x_part.device))
        if key not in reg:
            reg[key] = ops.conv_ksplit_workspace(x_part.device)
        return reg[key]

    def _second_stream(self, device, i):
        """Stream of image set / batch part i >= 1 (rnnpose_amd/streams.py: distinct hardware queues)."""
        from .streams import reserve
        if ops.profiling():          # per-launch timing: one stream, every launch alone on the chip (UpdateEngine._stream)
            return torch.cuda.current_stream()
        ss = reserve(device)
        return (ss.chain + [ss.aux])[(i - 1) % 3]

    def _forward_gen(self, W, x_nchw, out, normalize, split_out=False, ks=None):
        """One image set through the encoder (extractor.py:187-232); yields after every launch group."""
        E = EncoderEngine
        f = self.fnet
        x, ts = ops.stem_conv(W["stem"], x_nchw, normalize)                          # extractor.py:197 (+ CFNet.py:42-43)
        yield
        if ts is not None:             # relu(norm1(conv1 .)) (:198-199) stays lazy: applied where the stem output is consumed
            x_norm = (ops.instnorm_tiles_nhwc(x, ts, stats_only=True), True)
        else:                          # (ragged stem tiling: no tile statistics -> one full instance-norm pass)
            x, x_norm = ops.instnorm_nhwc(x, relu=True), None
        yield
        for li, layer in enumerate((f.layer1, f.layer2, f.layer3), start=1):
            for bi, blk in enumerate(layer):
                x = yield from E._block_gen(W, f"l{li}.{bi}", blk, x, x_norm, ks=ks, sp=self.single_product)
                x_norm = None
        if W["outr"] is not None and self.resident_1x1:
            if split_out:                # the volume build's operand, written in place (a batch slice of NHWC is contiguous)
                ops.conv1x1_resident(W["outr"], (x, 0), (out, 0), relu=False, dst_split=True)
                yield
                return
            o = torch.empty(*x.shape[:3], 256, device=x.device, dtype=torch.float32)
            ops.conv1x1_resident(W["outr"], (x, 0), (o, 0), relu=False)
        else:
            o, _ = E._conv(W["out"], x, stats=False, ks=ks, sp=self.single_product)
        yield
        if split_out:
            ops.split_hl(o, dst=out, a_scale=ops.A_SCALE)
        else:
            ops.nhwc_to_nchw(o, out=out)
        yield
