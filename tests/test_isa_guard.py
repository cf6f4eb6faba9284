"""Compile-time guard for the memory-bound kernels (CPU only: hipcc cross-compiles gfx950).

r02 finding (rule 1): a load under a bounds test, or a divergent branch around a load loop, makes
hipcc wait `s_waitcnt vmcnt(0)` after every load -- one dependent memory round trip each (corr_lookup 47 vs 28 us,
corr_weight 259 vs 64 us).  The kernels below issue their loads in batches behind a scheduling fence; this test reads
the generated ISA and fails when a change brings the serialised form back (many vmcnt(0) waits, few counted ones) or
makes one of them spill."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel substring -> (min loads, max vmcnt(0) waits, min counted waits)
EXPECT = {
    "corr_lookup.hip": {"corr_lookup_kernel": (30, 10, 20)},
    "pointwise.hip": {"corr_weight_kernel": (15, 8, 6)},      # (r06: tap pairs -- 3 loads per channel instead of 5)
    "lm.hip": {"lm_normal_eq_kernel": (16, 12, 10)},       # (the fused-tail instantiation adds its hand-off waits, the finalize loads and the inlined solve: 9; the plain one has 1)
    "nhwc_ops.hip": {"convex_upsample_nhwc_kernel": (18, 4, 3), "instnorm_apply_kernel": (6, 2, 2), "conv7x7_cin2_kernel": (50, 5, 40)},
    "stem.hip": {"stem_conv7x7_s2_kernel": (20, 4, 1)},        # (r06 persistent form: 2 x 10 patch loads + weights + bias; the next tile's patch is waited for once,
                                                               #  behind the MFMAs; the serialised form has one vmcnt(0) per load)
}


_ISA_CACHE = {}


def _isa(src, tmp_path):
    """-> (ISA text, {kernel symbol: spilled VGPRs}) of one source file under the library's own flags; compiled once per session."""
    if src not in _ISA_CACHE:
        _ISA_CACHE[src] = _isa_compile(src, tmp_path)
    return _ISA_CACHE[src]


def _isa_compile(src, tmp_path):
    from rnnpose_amd import build
    out = tmp_path / (os.path.basename(src) + ".s")
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + build.PER_FILE_FLAGS.get(os.path.basename(src), [])
    cmd = [build.hipcc(), "-S", "--cuda-device-only", "-x", "hip", src, "-o", str(out), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "rnnpose_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage"] + flags
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert out.exists(), r.stderr[-2000:]
    spills, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name:
            spills[name] = int(m.group(1))
    return out.read_text(), spills


@pytest.mark.parametrize("fname", sorted(EXPECT))
def test_loads_are_batched_not_serialised(fname, tmp_path):
    txt, spills = _isa(os.path.join(ROOT, "rnnpose_amd", "csrc", fname), tmp_path)
    bodies = {m.group(1): m.group(0) for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M)}
    for key, (min_loads, max_w0, min_counted) in EXPECT[fname].items():
        hits = [(sym, b) for sym, b in bodies.items() if key in sym]
        assert hits, f"{key} not found in the ISA of {fname}"
        for sym, body in hits:
            loads = len(re.findall(r"global_load|buffer_load", body))
            w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
            counted = len(re.findall(r"s_waitcnt vmcnt\([1-9]", body))
            assert loads >= min_loads, (sym, loads)
            assert w0 <= max_w0, f"{key}: {w0} vmcnt(0) waits for {loads} loads -- the loads are serialised again"
            assert counted >= min_counted, f"{key}: only {counted} counted waits for {loads} loads"
            assert spills.get(sym, 0) == 0, f"{key} spills {spills.get(sym)} VGPRs"


def test_no_packed_fp32_instruction_anywhere(tmp_path):
    """r05 (profiles/r05_determinism.txt, tools/probes/pk_f32_vs_mfma.hip): v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 occasionally
    compute wrong values for 16 lanes while another wave of the SIMD issues v_mfma_f32_16x16x32_f16 -- whichever kernel or stream that
    wave belongs to.  The library is built with -fno-slp-vectorize and writes no two-wide fp32 arithmetic, so none may appear in the
    ISA of any source file."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(ROOT, "rnnpose_amd", "csrc", "*.hip")))
    assert len(srcs) >= 17

    def scan(src):
        d = tmp_path / os.path.basename(src)
        d.mkdir()
        txt, _ = _isa(src, d)
        return os.path.basename(src), sorted(set(re.findall(r"v_pk_(?:mul|add|fma)_f32", txt)))
    with ThreadPoolExecutor(max_workers=8) as ex:
        found = {name: ops for name, ops in ex.map(scan, srcs) if ops}
    assert not found, f"packed fp32 instructions in {found}: build with -fno-slp-vectorize and keep fp32 arithmetic scalar"
