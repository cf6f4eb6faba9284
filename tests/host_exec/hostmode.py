"""TEST INFRASTRUCTURE: lets the `-m gpu` parity tests (and the Python front end they drive) run against the host-executed kernels.

`host_mode(lib_path)` is a context manager that, for its duration only and in this process only,
  * makes rnnpose_amd._lib.load() return the host library built by tests/host_exec/build_host.py (same 80 C-ABI entry points, the
    product's own prototypes),
  * replaces the front end's "must be a GPU tensor" guards (ops._chk, ops._nhwc) by their dtype / contiguity half,
  * maps the device "cuda" to the CPU for tensor factories and .to() / .cuda() (a TorchFunctionMode), and gives torch.cuda's stream /
    event / synchronize calls inert stand-ins (work is synchronous on the host).
Nothing here is imported by the package; outside the context the front end refuses CPU tensors as before (tests/test_host_logic.py
::test_ops_refuse_cpu_tensors keeps checking that)."""
from __future__ import annotations

import contextlib
import ctypes as C

import torch
from torch.overrides import TorchFunctionMode


def _is_cuda_dev(d):
    if isinstance(d, torch.device):
        return d.type == "cuda"
    if isinstance(d, str):
        return d.startswith("cuda")
    return False


def _map(a):
    if _is_cuda_dev(a):
        return torch.device("cpu")
    if isinstance(a, (list, tuple)) and any(_is_cuda_dev(x) for x in a):
        return type(a)(_map(x) for x in a)
    return a


class CudaIsCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs:
            kwargs["device"] = _map(kwargs["device"])
        name = getattr(func, "__name__", "")
        if name == "cuda":                                   # Tensor.cuda()
            return args[0]
        if name == "to" or name == "empty_like" or name == "zeros_like":
            args = tuple(_map(a) for a in args)
        if name == "pin_memory":
            return args[0]
        return func(*args, **kwargs)


class _Inert:
    """stream / event stand-in: every ordering call is a no-op (host execution is synchronous)"""
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def elapsed_time(self, other):
        return 0.0

    def query(self):
        return True


@contextlib.contextmanager
def host_mode(lib_path: str):
    from rnnpose_amd import _lib, ops
    lib = C.CDLL(lib_path)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.rnnpose_abi_version() == _lib.ABI_VERSION
    saved = {"lib": _lib._lib, "chk": ops._chk, "nhwc": ops._nhwc, "stream": ops._stream, "arm": ops.range_guard_arm,
             "cuda": {k: getattr(torch.cuda, k) for k in ("synchronize", "current_stream", "Stream", "Event", "stream", "is_available",
                                                          "current_device", "device", "set_device", "device_count")}}

    def chk(t, name, dtype=ops.F32):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        return (t if t.dtype == dtype else t.to(dtype)).contiguous()

    def nhwc(t, name):
        if not (t.dtype == ops.F32 and t.is_contiguous() and t.dim() == 4):
            raise ValueError(f"{name} must be a contiguous fp32 tensor shaped (B,H,W,C)")
        return t

    def arm(device=None):
        _lib.call("rnnpose_f16x3_saturation_check", 1)
        ops._guard_on = True

    _lib._lib = lib
    ops._chk, ops._nhwc, ops._stream, ops.range_guard_arm = chk, nhwc, (lambda: C.c_void_p(0)), arm
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Inert()
    torch.cuda.Stream = _Inert
    torch.cuda.Event = _Inert
    torch.cuda.stream = lambda s=None: _Inert()
    torch.cuda.is_available = lambda: True
    torch.cuda.current_device = lambda: 0
    torch.cuda.device = lambda *a, **k: _Inert()
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    try:
        with CudaIsCpu():
            yield ops
    finally:
        _lib._lib = saved["lib"]
        ops._chk, ops._nhwc, ops._stream, ops.range_guard_arm = saved["chk"], saved["nhwc"], saved["stream"], saved["arm"]
        for k, v in saved["cuda"].items():
            setattr(torch.cuda, k, v)
