"""Builds librnnpose_hip.so (the C-ABI library of include/rnnpose_hip.h) with hipcc for gfx950.

    python -m rnnpose_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored but
travels to the GPU box with the tree.  No torch headers are involved: the library only needs the HIP
runtime, and Python talks to it through ctypes (rnnpose_amd/_lib.py).
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "librnnpose_hip.so")
STAMP = os.path.join(LIBDIR, "librnnpose_hip.stamp")
ARCH = "gfx950"
# -fno-slp-vectorize for EVERY file (r05): no packed fp32 vector instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) may exist in
# this library.  On MI355X such an instruction occasionally returns wrong values for a group of 16 lanes while another wave on the
# same SIMD issues v_mfma_f32_16x16x32_{f16,bf16} (tools/probes/pk_f32_vs_mfma.hip: plain HIP, two streams -- 694 of 2000 launches
# differ next to the bf16 form, 2-9 next to the f16 form, 0 for scalar fp32, 0 next to 32x32x16 or 16x16x16 MFMAs, 0 for packed fp16).
# That was r04's "kernel-to-kernel visibility" finding: corr_weight, SLP-vectorised by plain -O3, running next to mask_upsample /
# conv1x1_resident of the other stream (profiles/r05_determinism.txt).  tests/test_isa_guard.py fails if one comes back.  (r02 had the
# flag on the MFMA files only, because packed fp32 next to a wave's OWN MFMAs is slow.)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-DNDEBUG", "-fno-slp-vectorize"]
PER_FILE_FLAGS = {}
FLAGS += os.environ.get("RNNPOSE_HIPCC_EXTRA", "").split()     # diagnostics builds (e.g. -DRP_ABL=..., tools/conv_ablate.sh), part of the stamp


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


TRAFFIC_KERNEL_FILES = ("conv_igemm.hip", "conv_strip.hip", "conv_strip_r32.hip", "conv_strip_r96.hip", "conv_strip_p1.hip", "corr_pyramid.hip")      # the kernels whose HBM counters profiles/traffic.json holds


def source_digest() -> str:
    """Digest of the sources the two kernels measured in profiles/traffic.json are compiled from (their .hip files, every shared
    csrc header, the compiler flags): bench.py refuses counters measured on other kernel sources.  (The stamp of the library
    itself, _digest(), covers every source and the public header.)"""
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in TRAFFIC_KERNEL_FILES] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
        sorted(glob.glob(os.path.join(CSRC, "*.cuh")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted((k, v) for k, v in PER_FILE_FLAGS.items() if k in TRAFFIC_KERNEL_FILES)).encode())
    return h.hexdigest()


def _digest() -> str:
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(CSRC, "*.cuh")))
    files.append(os.path.join(ROOT, "include", "rnnpose_hip.h"))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())      # repo-relative: the digest is the same wherever the tree is checked out
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into lib/librnnpose_hip.so unless the stamp matches.  Safe to call from several
    processes at once (one rank per GPU all call it): an exclusive file lock serialises them and the losers find the
    fresh stamp."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
                return LIB
            return _build_locked(dig, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(dig: str, verbose: bool) -> str:
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src) + ".o")
        cmd = [hipcc(), "-c", "-x", "hip", src, "-o", obj, "-I", os.path.join(ROOT, "include"), "-I", CSRC] + \
              [f for f in FLAGS if f != "-shared"] + PER_FILE_FLAGS.get(os.path.basename(src), [])
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
