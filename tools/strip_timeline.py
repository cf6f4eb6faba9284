#!/usr/bin/env python3
"""Where a strip-kernel workgroup spends its life: a -DRS_ABL=512 build (tools/strip_ablate.sh 512 -> gpurun_extra/sabl_512.so)
stamps the 100-MHz clock in wave 0 of every workgroup at kernel entry, after the prologue's barrier, after the main loop and at
the end of the epilogue.  Prints, per layer, the mean / median microseconds of the three phases, the span of the launch
(first entry -> last exit) and how many workgroups were resident on average (sum of lifetimes / span).
    RNNPOSE_LIB=$PWD/gpurun_extra/sabl_512.so python tools/strip_timeline.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnpose_amd import ops, _lib  # noqa: E402

lib = ctypes.CDLL(_lib.LIB_PATH)
lib.rnnpose_debug_strip_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
cases = [("enc l1 3x3 64->64 @240x320", [64], 64, 3, 3, 240, 320, False), ("enc l1 (split src)", [64], 64, 3, 3, 240, 320, True),
         ("enc l2 3x3 96->96 @120x160", [96], 96, 3, 3, 120, 160, False), ("enc l3 3x3 128->128 @60x80", [128], 128, 3, 3, 60, 80, False),
         ("gru zr 1x5 256->256", [128, 128], 256, 1, 5, 60, 80, True), ("heads 3x3 128->512", [128], 512, 3, 3, 60, 80, True),
         ("convc2 3x3 256->192", [256], 192, 3, 3, 60, 80, True)]
B = 8
for name, segs, co, kh, kw, hh, ww, hl in cases:
    ci = sum(segs)
    wt = torch.randn(co, ci, kh, kw, device="cuda") * (2.0 / (ci * kh * kw)) ** 0.5
    pc = ops.PackedConv(wt, torch.randn(co, device="cuda"), segs)
    xf = [(torch.randn(B, hh, ww, c, device="cuda"), 0) for c in segs]
    xs = [(ops.split_hl(t), 0) for t, _ in xf]
    out = torch.empty(B, hh, ww, (co + 7) // 8 * 8, device="cuda")
    run = lambda: ops.conv2d_nhwc(pc, xs if hl else xf, (out, 0), ops.EPI_RELU, src_hl=hl, dst_hl=hl, tile=5)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    n = 65536
    buf = np.zeros(n * 8, dtype=np.uint64)
    rc = lib.rnnpose_debug_strip_clk(buf.ctypes.data_as(ctypes.c_void_p), n)
    assert rc == 0, rc
    t = buf.reshape(n, 8).astype(np.int64)
    tpi = ops.conv_tiles_per_image(hh, ww, kh, kw, 1, co, 5)
    # number of workgroups of this launch: entries stamped in the last launch = those whose entry time >= the launch's first entry
    live = t[:, 3] > 0
    last = t[live]
    # the last launch: stamps within 1 ms of the newest exit
    newest = last[:, 3].max()
    idx_all = np.nonzero(live)[0]
    keep = last[:, 0] > newest - 100000 // 100 * 30           # 300 us window (100 MHz ticks: 100 per us)
    sel = last[keep]
    wg_id = idx_all[keep]
    pro, loop, epi = (sel[:, 1] - sel[:, 0]) / 100.0, (sel[:, 2] - sel[:, 1]) / 100.0, (sel[:, 3] - sel[:, 2]) / 100.0
    span = (sel[:, 3].max() - sel[:, 0].min()) / 100.0
    life = (sel[:, 3] - sel[:, 0]).sum() / 100.0
    print(f"{name:30s} wgs {len(sel):5d} span {span:7.1f} us | prologue {pro.mean():5.2f} (med {np.median(pro):5.2f}) loop {loop.mean():6.2f} (med {np.median(loop):6.2f}) "
          f"epilogue {epi.mean():5.2f} (med {np.median(epi):5.2f}) us | resident wgs {life / span:6.1f} = {life / span / 256:4.2f} per CU", flush=True)
    st = np.sort(sel[:, 0] - sel[:, 0].min()) / 100.0
    en = np.sort(sel[:, 3] - sel[:, 0].min()) / 100.0
    pc = lambda a, q: a[min(len(a) - 1, int(q * len(a)))]
    print(f"{'':30s} workgroup entry times (us after the first): 10 % {pc(st, .1):5.1f} | 50 % {pc(st, .5):5.1f} | 90 % {pc(st, .9):5.1f} | last {st[-1]:5.1f};  "
          f"exits: first {en[0]:5.1f} | 50 % {pc(en, .5):5.1f} | last {en[-1]:5.1f}", flush=True)
    lifeus = (sel[:, 3] - sel[:, 0]) / 100.0
    hist = np.histogram(lifeus, bins=8)
    print(f"{'':30s} lifetime histogram (us): " + " ".join(f"{int(c)}@{e:.0f}" for c, e in zip(hist[0], hist[1][:-1])), flush=True)
    print(f"{'':30s} mean lifetime by blockIdx % 8 (XCD): " + " ".join(f"{lifeus[wg_id % 8 == x].mean():5.1f}" for x in range(8))
          + " | by decile of blockIdx: " + " ".join(f"{lifeus[(wg_id * 10 // (wg_id.max() + 1)) == q].mean():5.1f}" for q in range(10)), flush=True)
    if sel[:, 5].max() > 0:      # finer stamps: address set-up done, all prologue requests issued, epilogue set-up done, last store issued
        d = lambda a, b: (sel[:, b] - sel[:, a]).mean() / 100.0
        print(f"{'':30s} prologue: set-up {d(0, 5):5.2f} | requests (+ split of the fp32 forms) {d(5, 6):5.2f} | wait + barrier {d(6, 1):5.2f} us;  "
              f"epilogue: set-up {d(2, 4):5.2f} | five blocks {d(4, 7):5.2f} | tail {d(7, 3):5.2f} us", flush=True)
