"""TEST INFRASTRUCTURE: builds `librnnpose_hostexec.so` -- the library's kernel sources compiled as plain C++ over
tests/host_exec/harness.hpp, so that their C-ABI entry points run on host memory (tests/test_kernels_on_host.py).

Sources are compiled WHERE THEY LIE (rnnpose_amd/csrc); the few statements a host compiler cannot take are rewritten in a scratch copy
by the PATCHES table below -- every rewrite is listed with the number of places it must hit, so a source edit that changes one of them
fails the build here instead of silently testing something else.  Nothing of this is reachable from the product path."""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "rnnpose_amd", "csrc")

# file -> [(regex, replacement, expected hits)]
PATCHES = {
    # `extern __shared__` (dynamic LDS): the harness hands out the launch's dynamic allocation
    "nhwc_ops.hip": [(r"extern __shared__ float wl\[\];", "float* wl = reinterpret_cast<float*>(hostexec::dyn_lds());", 1)],
    # hardware waits around the arrival counter of the fused LM tail: no meaning on the host (stores are visible in program order)
    "lm.hip": [(r'asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\);', ";", 2)],
    # WAVE-SYNCHRONOUS LDS: the epilogue passes the accumulators through a wave-private LDS tile -- every lane writes its 16 values,
    # then reads rows other lanes wrote, with no barrier in between: a wave's LDS operations execute in program order on the GPU.
    # Fibers of a wave do not run in lock step, so the scratch copy gets a wave synchronisation after the writes (RAW) and at the
    # end of the row-block iteration (WAR: the next iteration overwrites the tile).
    "conv_igemm.hip": [
        (r"(S\[\(\(r & 3\) \+ 8 \* \(r >> 2\) \+ 4 \* lh\) \* ES \+ ni \* 32 \+ l31\] = acc\[mi\]\[ni\]\[r\] \* p\.out_scale;)",
         r"\1\n    __builtin_amdgcn_wave_barrier();", 1),
        (r"(if \(p\.dsth\) store_quad_hl\(p\.dsth \+ pix \* p\.dsth_cs, p\.dsth_co \+ col, y\[0\], y\[1\], y\[2\], y\[3\], nv, p\.a_scale, sat_n\);\n    \})",
         r"\1\n    __builtin_amdgcn_wave_barrier();", 1),
    ],
    # r06: the LDS-DMA form of the volume kernel -- request / wait helpers and the LDS base as for the strip kernels (HEADER_PATCHES below)
    "corr_pyramid.hip": [
        (r"(?s)__device__ __forceinline__ void cp_glds16\(const void\* g, unsigned dst\) \{.*?\n\}\n",
         "__device__ __forceinline__ void cp_glds16(const void* g, unsigned dst) { hostexec::lds_dma(g, dst, 1); }\n", 1),
        (r'__device__ __forceinline__ void cp_wait_vm0\(\) \{ asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\); \}',
         "__device__ __forceinline__ void cp_wait_vm0() { hostexec::vm_wait(0); }", 1),
        (r"const unsigned lds0 = static_cast<unsigned>\(reinterpret_cast<size_t>\(\(__attribute__\(\(address_space\(3\)\)\) unsigned char\*\)sD\)\);",
         "const unsigned lds0 = hostexec::lds_register(sD);", 2),          # (variants 1 and 2)
    ],
    # an empty asm that only makes a value opaque to the optimiser: AMDGPU register class "v" -> a host register
    "mask_upsample.hip": [(r'asm volatile\("" : "\+v"\(aoff\)\);', 'asm volatile("" : "+r"(aoff));', 2)],          # (both tile geometries)
}

# The strip convolution kernels request their operands by LDS-DMA (inline assembly).  Their scratch copy gets host versions of the
# three request / wait helpers -- a request copies each lane's 16 bytes to LDS at once (the hardware completes it later and the kernel
# waits with counted s_waitcnt: instant completion is one of the timings a correct kernel must tolerate) and synchronises the wave
# around the copy (the hardware issues it for all lanes at one program point) -- and an LDS base that maps the kernel's numeric LDS
# addresses back to the host array.
HEADER_PATCHES = {
    "conv_strip_kernel.cuh": [
        (r"__device__ __forceinline__ void glds16\(const void\* g, unsigned dst\) \{.*?\n\}\n",
         "__device__ __forceinline__ void glds16(const void* g, unsigned dst) { hostexec::lds_dma(g, dst, 1); }\n", 1),
        (r"template <int NI>\n__device__ __forceinline__ void glds_rec\(const void\* base, unsigned voff, unsigned dst\) \{.*?\n  \}\n\}\n",
         "template <int NI>\n__device__ __forceinline__ void glds_rec(const void* base, unsigned voff, unsigned dst) {\n"
         "  hostexec::lds_dma(static_cast<const unsigned char*>(base) + voff, dst, 2 * NI);\n}\n", 1),
        (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(N\) : "memory"\);', "hostexec::vm_wait(N);", 1),      # r06: the counted wait retires the queue (harness.hpp)
        # the register loads of the fp32-source forms are vector-memory requests too: PAW places in the in-order queue per half block
        (r"(_Pragma\(\"unroll\"\) for \(int i = 0; i < NQ; \+\+i\) areg\[i\] = \*reinterpret_cast<const float4\*>\(asp\[i\]\);       \\\n)", "BACKREF1    hostexec::vm_note(PAW);  \\\n", 1),
        (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', ";", 1),
        (r'asm volatile\("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt\(0\)" : "=s"\(t_\)::"memory"\);', "t_ = 0;", 1),
        # r06, persistent form: an opaque move of the thread index (AMDGPU register class "v" -> a host register), the v_mov that puts the
        # compiler's wait for the bias loads in front of the sink requests (no meaning on the host), the start-delay loop of a measurement
        (r'asm volatile\("" : "\+v"\(tid_e_\)\);', 'asm volatile("" : "+r"(tid_e_));', 1),
        (r'asm volatile\("v_mov_b32 %0, %1" : "=v"\(bq\[e\]\) : "v"\(bq\[e\]\)\);', ";", 1),
        (r"while \(wall_clock64\(\) < t_end\) __builtin_amdgcn_s_sleep\(32\);", ";", 1),
        (r"const unsigned long long t_end = wall_clock64\(\) \+", "const unsigned long long t_end = 0ull +", 1),
        # wave-synchronous LDS staging of the two epilogues (as in conv_igemm.hip): a wave synchronisation before and after every
        # staging write
        (r"\n  stage\(0, acc\[0\]\[0\]\);", "\n  __builtin_amdgcn_wave_barrier(); stage(0, acc[0][0]); __builtin_amdgcn_wave_barrier();", 1),
        (r"\n    if \(mi \+ 1 < NBLK\) stage\(mi \+ 1, acc\[mi \+ 1 < NBLK \? mi \+ 1 : mi\]\[0\]\);",
         "\n    __builtin_amdgcn_wave_barrier();\n    if (mi + 1 < NBLK) stage(mi + 1, acc[mi + 1 < NBLK ? mi + 1 : mi][0]);\n    __builtin_amdgcn_wave_barrier();", 1),
        (r"(S_\[\(\(r & 3\) \+ 8 \* \(r >> 2\) \+ 4 \* lh\) \* ES \+ ni \* 32 \+ l31\] = at\[ni\]\[r\] \* p\.out_scale;)",
         "BACKREF1\n    __builtin_amdgcn_wave_barrier();", 1),
        (r"(if \(p\.dsth\) store_quad_hl\(p\.dsth \+ pix \* p\.dsth_cs, p\.dsth_co \+ colq, y\[0\], y\[1\], y\[2\], y\[3\], nv, p\.a_scale, sat_n\);\n    \})",
         "BACKREF1\n    __builtin_amdgcn_wave_barrier();", 1),
        (r"const unsigned lds0 = static_cast<unsigned>\(reinterpret_cast<size_t>\(\(__attribute__\(\(address_space\(3\)\)\) unsigned char\*\)lds\)\);",
         "const unsigned lds0 = hostexec::lds_register(lds);", 1),
    ],
}
STRIP_SOURCES = ["conv_strip.hip", "conv_strip_r32.hip", "conv_strip_r96.hip", "conv_strip_p1.hip"]      # copied next to the patched header (quote includes look there first)

SOURCES = ["pointwise.hip", "corr_pyramid.hip", "corr_lookup.hip", "corr_convc1.hip", "corr_alt.hip", "conv_igemm.hip", "conv1x1_resident.hip", "stem.hip",
           "nhwc_ops.hip", "eval_metrics.hip", "zoom_crop.hip", "raster.hip", "lm.hip", "mask_upsample.hip"]
EXTRA = ["runtime_host.cpp"]


def clang() -> str:
    for c in (os.environ.get("HOSTEXEC_CXX"), "/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("no clang++ of the ROCm toolchain found (needed for _Float16 vectors and the HIP headers)")


def _digest() -> str:
    h = hashlib.sha256()
    h.update((os.environ.get("HOSTEXEC_CXXFLAGS", "") + os.environ.get("HOSTEXEC_MUTATE", "")).encode())
    for d, names in ((CSRC, sorted(os.listdir(CSRC))), (HERE, sorted(os.listdir(HERE)))):
        for n in names:
            p = os.path.join(d, n)
            if os.path.isfile(p) and not n.endswith((".pyc", ".so")):
                h.update(n.encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def build(outdir: str) -> str:
    """-> path of the host library (built once per source digest inside outdir)."""
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, f"librnnpose_hostexec_{_digest()}.so")
    if os.path.exists(lib):
        return lib
    extra = os.environ.get("HOSTEXEC_CXXFLAGS", "").split()       # e.g. -fsanitize=address -fno-omit-frame-pointer (see tests/host_exec/README.md)
    flags = extra + ["-x", "c++", "-std=c++20", "-O1", "-fPIC", "-pthread", "-D__HIP_PLATFORM_AMD__", "-DNDEBUG", "-U_FORTIFY_SOURCE", "-w", "-I", outdir, "-I", "/opt/rocm/include",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", HERE]
    units = []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        if name in PATCHES:
            txt = open(src).read()
            for rx, rep, hits in PATCHES[name]:
                txt, n = re.subn(rx, rep if '\\1' in rep else (lambda m, rep=rep: rep), txt)
                if n != hits:
                    raise RuntimeError(f"host-exec patch of {name}: {rx!r} matched {n} places, expected {hits}")
            src = os.path.join(outdir, "patched_" + name)
            open(src, "w").write(txt)
        tu = os.path.join(outdir, name + ".host.cpp")
        open(tu, "w").write(f'#include "harness.hpp"\n#include "{src}"\n')
        units.append(tu)
    # HOSTEXEC_MUTATE="file::regex::replacement": one extra rewrite of a header's scratch copy -- mutation checks ("does the tier see this
    # bug?"), e.g. the round-5 fourth-fragment bug of the 32-row stride-2 strips (profiles/r05_hostexec_mutation.txt)
    mutate = os.environ.get("HOSTEXEC_MUTATE", "")
    for name, patches in HEADER_PATCHES.items():
        txt = open(os.path.join(CSRC, name)).read()
        if mutate and mutate.split("::")[0] == name:
            _, rx, rep = mutate.split("::")
            txt, n = re.subn(rx, lambda m: rep, txt)
            if n != 1:
                raise RuntimeError(f"HOSTEXEC_MUTATE matched {n} places")
        for rx, rep, hits in patches:
            txt, n = re.subn(rx, lambda m, rep=rep: rep.replace("BACKREF1", m.group(1) if m.groups() else ""), txt, flags=re.S)
            if n != hits:
                raise RuntimeError(f"host-exec patch of {name}: {rx!r} matched {n} places, expected {hits}")
        open(os.path.join(outdir, name), "w").write(txt)
    for name in STRIP_SOURCES:
        cp = os.path.join(outdir, name)
        open(cp, "w").write(open(os.path.join(CSRC, name)).read())
        tu = os.path.join(outdir, name + ".host.cpp")
        open(tu, "w").write(f'#include "harness.hpp"\n#include "{cp}"\n')
        units.append(tu)
    units += [os.path.join(HERE, n) for n in EXTRA]

    def cc(tu):
        obj = os.path.join(outdir, os.path.basename(tu) + ".o")
        fl = [("-O0" if (f == "-O1" and "conv_strip" in os.path.basename(tu)) else f) for f in flags]     # (the strip units: 68 kernel instantiations, compile time over run time)
        r = subprocess.run([clang(), "-c", tu, "-o", obj] + fl, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"host build of {tu} failed:\n{r.stderr[-4000:]}")
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, units))
    r = subprocess.run([clang(), "-shared", "-pthread", "-o", lib] + [f for f in extra if f.startswith("-fsanitize")] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host link failed:\n" + r.stderr[-4000:])
    return lib


if __name__ == "__main__":
    import sys
    print(build(sys.argv[1] if len(sys.argv) > 1 else "/tmp/rnnpose_hostexec"))
