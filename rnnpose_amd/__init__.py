"""rnnpose_amd -- MI355X-native (gfx950) implementation of RNNPose's recurrent pose-refinement hot path.

Layers:
  csrc/*.hip + include/rnnpose_hip.h   hand-written HIP kernels behind a C ABI (librnnpose_hip.so)
  _lib / ops                           ctypes binding + torch-tensor front end (device memory, streams)
  corr / update / cfnet / extractor    CorrBlock, BasicUpdateBlock, GRU_CFUpdator, ImageFeaEncoder
  transformation                       SE3 / SE3Sequence (+ cholesky.solve, se3_matrix_* helpers)
  pose_refiner                         PoseRefiner (the loop), SyntheticRenderer
  distributed                          one-process-per-GPU sharding + the end-of-epoch RCCL all-reduce
  synthetic                            closed-form synthetic inputs shared by tests, bench and golden generator
"""
from . import synthetic  # noqa: F401

__all__ = ["synthetic", "ops", "CorrBlock", "BasicUpdateBlock", "GRU_CFUpdator", "ImageFeaEncoder", "SE3",
           "SE3Sequence", "PoseRefiner", "SyntheticRenderer"]


def __getattr__(name):   # lazy: importing the package must not require torch or the built library
    import importlib
    table = {
        "ops": ("rnnpose_amd.ops", None),
        "CorrBlock": ("rnnpose_amd.corr", "CorrBlock"),
        "BasicUpdateBlock": ("rnnpose_amd.update", "BasicUpdateBlock"),
        "GRU_CFUpdator": ("rnnpose_amd.cfnet", "GRU_CFUpdator"),
        "ImageFeaEncoder": ("rnnpose_amd.cfnet", "ImageFeaEncoder"),
        "SE3": ("rnnpose_amd.transformation", "SE3"),
        "SE3Sequence": ("rnnpose_amd.transformation", "SE3Sequence"),
        "PoseRefiner": ("rnnpose_amd.pose_refiner", "PoseRefiner"),
        "SyntheticRenderer": ("rnnpose_amd.pose_refiner", "SyntheticRenderer"),
    }
    if name in table:
        mod, attr = table[name]
        m = importlib.import_module(mod)
        return m if attr is None else getattr(m, attr)
    raise AttributeError(name)
