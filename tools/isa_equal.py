#!/usr/bin/env python3
"""Are the kernels whose HBM counters profiles/traffic.json holds the SAME MACHINE CODE at HEAD as at the commit they were measured on?

    python tools/isa_equal.py <rev> [--write]

profiles/traffic.json records the digest of the kernel SOURCES it was measured on and bench.py refuses counters of other sources.  A
source edit that only ADDS template instantiations (or edits code that is compiled out of every existing one) changes the digest but
not one instruction of the kernels that were measured.  This tool proves or refutes that, on CPU (hipcc cross-compiles gfx950):

  1. `git archive <rev>` of rnnpose_amd/csrc + include into a scratch directory;
  2. every file of build.TRAFFIC_KERNEL_FILES compiled to ISA (-S, the library's own flags) in both trees;
  3. per kernel symbol: the instruction stream (label .. s_endpgm) AND its .amdhsa kernel descriptor block compared as text;
  4. the host-side kernel choice compared too: both trees' libraries answer rnnpose_conv_tiles_per_image_desc / rnnpose_conv_products_desc
     for every convolution launch of the headline step (the encoder and update-block layer list at B = 8 and B = 4, 480x640), which
     encodes the kernel family and strip height a launch takes.

Every symbol of <rev> identical at HEAD and every headline launch tiled identically  =>  the counters are counters of HEAD's kernels.
--write then re-stamps profiles/traffic.json with HEAD's digest and records where the numbers were measured and what was compared."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnpose_amd import build  # noqa: E402


def isa_of(tree, fname, tmp):
    src = os.path.join(tree, "rnnpose_amd", "csrc", fname)
    out = os.path.join(tmp, hashlib.md5(src.encode()).hexdigest() + ".s")
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + build.PER_FILE_FLAGS.get(fname, [])
    cmd = [build.hipcc(), "-S", "--cuda-device-only", "-x", "hip", src, "-o", out, "-I", os.path.join(tree, "include"),
           "-I", os.path.join(tree, "rnnpose_amd", "csrc")] + flags
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
    if not os.path.exists(out):
        raise SystemExit(r.stderr[-3000:])
    txt = open(out).read()
    txt = re.sub(r"^\s*;.*\n", "", txt, flags=re.M)                    # comment-only lines (source paths, remarks)
    txt = re.sub(r"\s;[^\n]*", "", txt)                                # trailing comments
    txt = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", txt)                    # basic-block labels carry the function's INDEX in the file: new kernels renumber them
    txt = re.sub(r"[ \t]+$", "", txt, flags=re.M)                       # (the comment column after a label moves with the label's width)
    bodies = {m.group(1): m.group(0) for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M)}
    descs = {m.group(1): m.group(0) for m in re.finditer(r"\.amdhsa_kernel (\S+).*?\.end_amdhsa_kernel", txt, re.S)}
    return bodies, descs


def headline_launches():
    """(B, H, W, kh, kw, stride, c_out, [source channel counts], split sources, fused norm) of the convolutions of one headline step."""
    L = []
    for B in (8, 4):                      # merged encoder batch (2 sets x 4 = 8 when merged; one set per stream: 4 x 2 images ... both asked)
        for (h, w) in ((240, 320),):
            L += [(B, h, w, 3, 3, 1, 64, [64], 0, n) for n in (0, 1)]
        L += [(B, 240, 320, 3, 3, 2, 96, [64], 0, 0), (B, 240, 320, 1, 1, 2, 96, [64], 0, 0)]
        L += [(B, 120, 160, 3, 3, 1, 96, [96], 0, n) for n in (0, 1)]
        L += [(B, 120, 160, 3, 3, 2, 128, [96], 0, 0), (B, 120, 160, 1, 1, 2, 128, [96], 0, 0)]
        L += [(B, 60, 80, 3, 3, 1, 128, [128], 0, n) for n in (0, 1)]
        L += [(B, 60, 80, 1, 1, 1, 256, [128], 0, 0)]
    for B in (8, 4):                      # update block at 1/8 resolution: full batch and half-batch chains
        h, w = 60, 80
        L += [(B, h, w, 3, 3, 1, 192, [256], hl, 0) for hl in (0, 1)]                       # convc2
        L += [(B, h, w, 3, 3, 1, 64, [128], hl, 0) for hl in (0, 1)]                        # convf2
        L += [(B, h, w, 3, 3, 1, 126, [192, 64], hl, 0) for hl in (0, 1)]                   # conv
        for (kh, kw) in ((1, 5), (5, 1)):
            L += [(B, h, w, kh, kw, 1, 256, [128, 128, 128], hl, 0) for hl in (0, 1)]       # z | r
            L += [(B, h, w, kh, kw, 1, 128, [128, 128, 128], hl, 0) for hl in (0, 1)]       # q
        L += [(B, h, w, 3, 3, 1, 256, [128], hl, 0) for hl in (0, 1)]                       # flow head / mask head first layers
        L += [(B, h, w, 3, 3, 1, 512, [128], hl, 0) for hl in (0, 1)]
    return L


def tiling_answers(tree):
    """Build that tree's library (its own build.py, into its own lib/) and ask it how it would tile every headline launch."""
    r = subprocess.run([sys.executable, "-m", "rnnpose_amd.build"], cwd=tree, capture_output=True, text=True)
    lib_path = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else None
    if not lib_path or not os.path.exists(lib_path):
        raise SystemExit("build failed in " + tree + "\n" + r.stdout[-2000:] + r.stderr[-2000:])
    code = r"""
import sys, json
sys.path.insert(0, %r)
from rnnpose_amd import ops
import tools_headline as th
out = []
for (B, H, W, kh, kw, s, co, srcs, hl, norm) in th.L:
    out.append(ops.conv_tiles_per_image(H, W, kh, kw, s, co, 0, B, src_counts=srcs, fused_norm=bool(norm)))
print(json.dumps(out))
""" % tree
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "tools_headline.py"), "w").write("L = " + repr(headline_launches()) + "\n")
        r = subprocess.run([sys.executable, "-c", code], cwd=td, capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=td))
    if r.returncode != 0:
        raise SystemExit("tile query failed in " + tree + "\n" + r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    rev = sys.argv[1]
    write = "--write" in sys.argv
    sha = subprocess.run(["git", "rev-parse", "--short", rev], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    with tempfile.TemporaryDirectory(prefix="isa_equal_") as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        ar = subprocess.run(["git", "archive", rev, "rnnpose_amd", "include"], cwd=ROOT, capture_output=True)
        subprocess.run(["tar", "-x", "-C", old], input=ar.stdout, check=True)
        report = {"measured_at": sha, "files": {}, "identical": True}
        for f in build.TRAFFIC_KERNEL_FILES:
            if not os.path.exists(os.path.join(old, "rnnpose_amd", "csrc", f)):
                report["files"][f] = "absent at " + sha
                continue
            b0, d0 = isa_of(old, f, tmp)
            b1, d1 = isa_of(ROOT, f, tmp)
            diff = sorted(k for k in b0 if b0[k] != b1.get(k) or d0.get(k) != d1.get(k))
            report["files"][f] = {"kernels_at_rev": len(b0), "kernels_at_head": len(b1), "new_at_head": sorted(set(b1) - set(b0)),
                                  "differing": diff}
            report["identical"] &= not diff
            print(f"{f}: {len(b0)} kernels at {sha}, {len(b1)} at HEAD, {len(diff)} differ, {len(set(b1) - set(b0))} new")
        t0, t1 = tiling_answers(old), tiling_answers(ROOT)
        L = headline_launches()
        moved = [L[i] for i in range(len(L)) if t0[i] != t1[i]]
        report["headline_launches_compared"] = len(L)
        report["headline_launches_tiled_differently"] = moved
        report["identical"] &= not moved
        print(f"tiling of {len(L)} headline launches: {len(moved)} differ")
    print(json.dumps(report, indent=1)[:3000])
    if write:
        if not report["identical"]:
            raise SystemExit("not identical: profiles/traffic.json left alone")
        p = os.path.join(ROOT, "profiles", "traffic.json")
        t = json.load(open(p))
        t["carried_over"] = {"measured_on_digest": t.get("carried_over", {}).get("measured_on_digest", t["csrc_digest"]), "measured_at": sha,
                             "check": "tools/isa_equal.py " + sha + ": every kernel of " + ", ".join(build.TRAFFIC_KERNEL_FILES) +
                                      " present at the measured commit has byte-identical ISA and kernel descriptor at this digest, and every "
                                      "convolution launch of the headline step is tiled (= dispatched) identically; the sources differ by "
                                      "instantiations that the headline step does not launch",
                             "new_kernels": sum(len(v["new_at_head"]) for v in report["files"].values() if isinstance(v, dict))}
        t["csrc_digest"] = build.source_digest()
        json.dump(t, open(p, "w"), indent=1)
        print("re-stamped", p)


if __name__ == "__main__":
    main()
