"""TEST INFRASTRUCTURE: lets the `-m gpu` parity tests (and the Python front end they drive) run against the host-executed kernels.

`host_mode(lib_path)` is a context manager that, for its duration only and in this process only,
  * makes rnnpose_amd._lib.load() return the host library built by tests/host_exec/build_host.py (same 80 C-ABI entry points, the
    product's own prototypes),
  * maps the device "cuda" to the CPU for tensor factories and .to() / .cuda() and answers `tensor.is_cuda` with True (a
    TorchFunctionMode): the front end itself -- ops.py, the engines, the module classes with their "GPU tensors only" guards -- runs
    UNMODIFIED,
  * gives torch.cuda's stream / event / synchronize calls inert stand-ins (work is synchronous on the host) and refuses hipGraph
    capture (PoseRefiner then takes its eager path).
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
        if name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", None) == "is_cuda":
            return True                                      # the front end's "GPU tensors only" guards (ops._chk, module forwards)
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
    names = ("synchronize", "current_stream", "Stream", "Event", "stream", "is_available", "current_device", "device", "set_device",
             "device_count", "CUDAGraph", "graph")
    saved_lib, saved_cuda = _lib._lib, {k: getattr(torch.cuda, k) for k in names}

    def no_graphs(*a, **k):
        raise RuntimeError("host execution: no hipGraph capture (callers fall back to eager launches)")

    _lib._lib = lib
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
    torch.cuda.CUDAGraph = no_graphs
    torch.cuda.graph = no_graphs
    try:
        with CudaIsCpu():
            yield ops
    finally:
        _lib._lib = saved_lib
        for k, v in saved_cuda.items():
            setattr(torch.cuda, k, v)
