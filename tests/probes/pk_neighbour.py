"""TEST INFRASTRUCTURE: a co-tenant kernel made of PACKED fp32 arithmetic (pk_neighbour.hip, built with plain -O3 the way an integrator's
own code would be) on its own stream, every launch compared ON THE DEVICE with the kernel's solo output.  Used by
tests/test_gpu_reproducibility.py::test_packed_fp32_neighbour_is_exact_next_to_the_refinement_loop and tools/pk_neighbour_probe.py."""
import ctypes as C
import os
import subprocess
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_neighbour(outdir=None):
    outdir = outdir or tempfile.mkdtemp(prefix="pk_neighbour_")
    so = os.path.join(outdir, "libpk_neighbour.so")
    hipcc = "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tests", "probes", "pk_neighbour.hip"), "-o", so],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-2000:])
    asm = os.path.join(outdir, "pk_neighbour.s")          # the device ISA as text (what the test greps for v_pk_*_f32)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", os.path.join(ROOT, "tests", "probes", "pk_neighbour.hip"), "-o", asm],
                       capture_output=True, text=True)
    isa = open(asm).read() if r.returncode == 0 else ""
    lib = C.CDLL(so)
    lib._isa = isa
    lib.pk_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pk_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib, so


class Neighbour:
    """The packed-fp32 kernel on its own stream: launch(k) writes slot k's counter of differing outputs."""

    def __init__(self, nwg=2048, chain=192, slots=4096, device="cuda"):
        self.lib, self.so = build_neighbour()
        self.isa = self.lib._isa
        self.nwg, self.chain, self.n = nwg, chain, nwg * 256
        g = torch.Generator(device="cpu").manual_seed(3)
        self.inp = (torch.rand(1024, generator=g) * 0.5 + 0.25).to(device)
        self.out = torch.empty(self.n, device=device)
        self.bad = torch.zeros(slots, device=device, dtype=torch.int32)
        self.stream = torch.cuda.Stream()
        self.k = 0
        self.run(self.out, torch.cuda.current_stream())
        torch.cuda.synchronize()
        self.solo = self.out.clone()
        assert float(self.solo.abs().max()) > 0 and bool(torch.isfinite(self.solo).all())

    def run(self, out, stream):
        rc = self.lib.pk_run(self.inp.data_ptr(), out.data_ptr(), self.nwg, self.chain, C.c_void_p(stream.cuda_stream))
        assert rc == 0, rc

    def launch(self, count=1):
        for _ in range(count):
            self.run(self.out, self.stream)
            rc = self.lib.pk_check(self.out.data_ptr(), self.solo.data_ptr(), self.n, self.bad[self.k:].data_ptr(), C.c_void_p(self.stream.cuda_stream))
            assert rc == 0, rc
            self.k += 1

    def result(self):
        torch.cuda.synchronize()
        b = self.bad[: self.k].cpu()
        r = (int((b != 0).sum()), self.k, int(b.sum()))
        self.bad.zero_()
        self.k = 0
        return r          # (launches that differ, launches, differing values)


def next_to(nb, work, launches, per=8):
    """`work()` enqueues library launches on the library's streams; `per` neighbour launches are enqueued after each call."""
    done = 0
    while done < launches:
        work()
        nb.launch(per)
        done += per
    return nb.result()
