"""fmx: host side of the MI355X-native FM / DeepFM / NFM online hot path.

PyTorch-ROCm is used for device memory, streams and torch.distributed only; the arithmetic of the hot path runs in
libfmx.so (hand-written gfx950 kernels, fm-for-online-recommendation_amd/csrc/) through the C ABI of include/fmx.h.
There is no CPU path and no PyTorch fallback: without a ROCm GPU or without the built library the constructors raise.
"""
from . import _lib  # noqa: F401
from .table import FlatTable, padded_k  # noqa: F401
from .engine import FMEngine, Hyper, normalize_inputs  # noqa: F401
from .distributed import DataParallelFM, HipBackend  # noqa: F401
from .deep import DeepFMTrainer, HipDeepBackend, OwnerDeepFMTrainer  # noqa: F401
