"""The C-ABI library builds for gfx950, loads, and exports every symbol include/rnnpose_hip.h declares.
No GPU needed: only host-side entry points (layout, workspace size, argument validation) are called."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    from rnnpose_amd import build, _lib
    build.build()
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rnnpose_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rnnpose_[a-z0-9_]+|findNearestPointIdxLauncher)\s*\(", src)))


def test_header_symbols_exported_and_typed(lib):
    from rnnpose_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.rnnpose_abi_version() == 3


def test_gfx950_code_object_present():
    """The fat binary must carry a gfx950 code object (and nothing else: no portability layer)."""
    from rnnpose_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_"):
        assert b"amdgcn-amd-amdhsa--" + other not in blob


def test_pyramid_layout(lib):
    offs = (C.c_int64 * 5)()
    hl = (C.c_int * 4)()
    wl = (C.c_int * 4)()
    assert lib.rnnpose_corr_pyramid_layout(8, 60, 80, 4, offs, hl, wl) == 0
    assert list(hl) == [60, 30, 15, 7] and list(wl) == [80, 40, 20, 10]
    n = 8 * 4800
    l0 = n * (8 * 5) * 128                                                    # level 0 j-patch-major: 8 x 5 whole patches of 8 x 16 cells
    assert list(offs) == [0, l0, l0 + n * 1200, l0 + n * 1500, l0 + n * 1570]
    assert lib.rnnpose_corr_pyramid_layout(1, 30, 30, 4, offs, hl, wl) == 0
    assert list(hl) == [30, 15, 7, 3]
    assert lib.rnnpose_corr_pyramid_layout(1, 4, 4, 4, offs, hl, wl) == 1        # level 3 would be empty
    assert b"too small" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_pyramid_layout(1, 16, 16, 5, offs, hl, wl) == 1


def test_workspace_and_argument_validation(lib):
    assert lib.rnnpose_lm_workspace_bytes(8, 480, 640) == 8 * 75 * 32 * 8 + 8 * 8       # one arrival counter per image + partial records (4096 pixels each)
    assert lib.rnnpose_lm_workspace_bytes(1, 16, 16) == 32 * 8 + 8
    assert lib.rnnpose_lm_workspace_bytes(0, 16, 16) == 0
    null = C.c_void_p(0)
    assert lib.rnnpose_corr_pyramid_f32(null, null, 1, 256, 16, 16, 4, null, null) == 1
    assert b"null pointer" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_lookup_f32(null, null, 1, 16, 16, 4, 4, null, null) == 1
    one = C.c_void_p(16)    # non-null dummy, never dereferenced on the host
    assert lib.rnnpose_corr_lookup_f32(one, one, 1, 16, 16, 4, 3, one, null) == 1
    assert b"radius" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_lookup_f32(one, one, 1, 8, 8, 4, 4, one, null) == 1     # 1x1 top level
    assert lib.rnnpose_convex_upsample_f32(one, one, 1, 8, 8, 4, one, null) == 1
    assert lib.rnnpose_corr_weight_f32(one, one, one, 2, one, one, 1, 32, 8, 8, one, null) == 1
    assert lib.rnnpose_corr_pyramid_f32(one, one, 1, 250, 16, 16, 4, one, null) == 1
    assert b"multiple of 16" in lib.rnnpose_last_error()
    # r03 entry points: operands handed over as split tensors; K-split workspace / limits of the convolution
    assert lib.rnnpose_corr_pyramid_split(null, null, 1, 256, 16, 16, 4, 8.0, null, null) == 1
    assert b"null pointer" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_pyramid_split(C.c_void_p(8), C.c_void_p(16), 1, 256, 16, 16, 4, 8.0, one, null) == 1
    assert b"16-byte aligned" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_pyramid_split(one, one, 1, 48, 16, 16, 4, 8.0, one, null) == 1
    assert b"multiple of 32" in lib.rnnpose_last_error()
    assert lib.rnnpose_conv_ksplit_workspace_bytes() == 1024 + 192 * 256 * 32 * 4
    assert lib.rnnpose_conv_ksplit_limits(0, 4) == 1 and b"max_tiles" in lib.rnnpose_last_error()
    assert lib.rnnpose_conv_ksplit_limits(24, 9) == 1
    assert lib.rnnpose_conv_ksplit_limits(24, 4) == 0
    assert lib.rnnpose_conv_ksplit(1) == 0
    # volume-free lookup (AlternateCorrBlock)
    assert lib.rnnpose_fmap_pyramid_floats(2, 16, 24, 256, 4) == 2 * 256 * (8 * 12 + 4 * 6 + 2 * 3)
    assert lib.rnnpose_fmap_pyramid_floats(2, 16, 24, 256, 1) == 0
    assert lib.rnnpose_corr_alt_lookup_f32(null, null, null, null, 1, 16, 16, 256, 4, 4, null, 324, 0, null) == 1
    assert lib.rnnpose_corr_alt_lookup_f32(one, one, one, one, 1, 16, 16, 256, 4, 3, one, 324, 0, null) == 1
    assert b"radius" in lib.rnnpose_last_error()
    assert lib.rnnpose_corr_alt_lookup_f32(one, one, one, one, 1, 16, 16, 258, 4, 4, one, 324, 0, null) == 1
    assert lib.rnnpose_corr_alt_lookup_f32(one, one, one, one, 1, 16, 16, 256, 4, 4, one, 300, 0, null) == 1
    assert b"channel stride" in lib.rnnpose_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rnnpose_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_header_is_plain_c99_and_links_against_the_library(tmp_path):
    """The boundary is a C ABI: the header must compile as strict C99 (no C++-isms, no torch types), and a C program
    must link against the shared library and resolve a symbol without any Python or torch in the process."""
    import shutil
    import subprocess
    from rnnpose_amd import build
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    lib_path = build.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <stdint.h>\n#include "rnnpose_hip.h"\n'
                   "int main(void) {\n"
                   "  rnnpose_conv_desc_t d; rnnpose_conv_src_t s; (void)d; (void)s;\n"
                   "  int64_t off[5]; int hl[4], wl[4];\n"
                   "  if (rnnpose_corr_pyramid_layout(2, 16, 24, 4, off, hl, wl) != 0) return 2;\n"
                   '  printf("%d %lld\\n", rnnpose_abi_version(), (long long)off[4]);\n'
                   "  return 0;\n}\n")
    exe = tmp_path / "t"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                        str(src), "-o", str(exe), lib_path, "-Wl,-rpath," + os.path.dirname(lib_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    ver, total = out.stdout.split()
    n = 2 * 16 * 24
    assert int(ver) == 3 and int(total) == n * (2 * 2 * 128 + 8 * 12 + 4 * 6 + 2 * 3)     # level 0: 2 x 2 whole patches (16 x 24 -> 16 x 32 cells)


def test_graft_entry_build_runs():
    """__graft_entry__.build() (what the driver calls on the CPU box each round): compiles / finds the library, checks its ABI version
    against the front end's and imports the package -- r04's record pass caught a stale version assert in it."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
    from rnnpose_amd import _lib
    assert _lib.load().rnnpose_abi_version() == _lib.ABI_VERSION == 3


def test_conv_tiling_rule_is_a_host_function(lib):
    """rnnpose_conv_tiles_per_image_ex (records per image of a convolution's tile statistics = how the launch will tile) runs on
    the host: 160-row strips (10 x 16 patches / runs of 160 pixels) when they give the launch >= 240 workgroups, 32-row strips
    (2 x 16 / runs of 32) for launches of at most 8192 pixels, else the 128-row kernels (8 x 16 patches / runs of 128)."""
    f = lib.rnnpose_conv_tiles_per_image_ex
    assert f(60, 80, 3, 3, 1, 192, 0, 8) == 30 and f(60, 80, 3, 3, 1, 192, 1, 8) == 40       # headline, B = 8: strips / forced 128-row
    assert f(60, 80, 3, 3, 1, 192, 0, 1) == 150 and f(60, 80, 3, 3, 1, 192, 6, 8) == 150     # one image: 32-row strips / forced
    assert f(60, 80, 1, 5, 1, 256, 0, 8) == 30 and f(60, 80, 1, 5, 1, 256, 0, 1) == 150      # linear strips: runs of 160 / 32 pixels
    assert f(30, 30, 1, 5, 1, 256, 0, 1) == 29 and f(30, 30, 3, 3, 1, 128, 0, 1) == 30       # a single LINEMOD crop: 32-row strips
    assert f(30, 30, 1, 5, 1, 256, 0, 16) == 8 and f(30, 30, 3, 3, 1, 128, 0, 16) == 8       # 16 crops (14 400 pixels): 128-row kernels
    assert f(240, 320, 3, 3, 1, 64, 0, 8) == 24 * 20                                         # encoder, 64 channels: two-wave strips
    assert f(60, 80, 3, 3, 2, 128, 0, 8) == lib.rnnpose_conv_tiles_per_image(60, 80, 3, 3, 2)    # stride 2: never strips
    assert f(60, 80, 3, 3, 1, 32, 5, 8) == -1 and f(60, 80, 3, 3, 1, 192, 8, 8) == -1 and f(60, 80, 3, 3, 1, 192, 0, 0) == -1
    # r06: tile code 7 = 96-row strips (6 x 16 patches / runs of 96 pixels); the automatic choice takes them when the launch's 160-row strips
    # would be at most 512 waves (the 128-column layers of a half-batch chain), 32-row strips when at most 256 waves (convf2 there)
    assert f(60, 80, 3, 3, 1, 192, 7, 8) == 50 and f(60, 80, 1, 5, 1, 128, 7, 8) == 50
    assert f(60, 80, 1, 5, 1, 128, 0, 4) == 50 and f(60, 80, 1, 5, 1, 128, 0, 8) == 30 and f(60, 80, 1, 5, 1, 256, 0, 4) == 30
    assert f(60, 80, 3, 3, 1, 126, 0, 4) == 50 and f(60, 80, 3, 3, 1, 64, 0, 4) == 150 and f(60, 80, 3, 3, 1, 64, 0, 8) == 50
