"""mft_amd -- MI355X-native hot path of the MFT dense point tracker.

Drop-in names: ``MFT`` (tracker), ``FlowOUTrackingResult`` / ``FlowOUResult``,
``RAFTWrapper`` (flow plugin), ``chain_results``, ``get_flowou_with_cache``.
Compute runs in ``libmftx.so`` (hand-written gfx950 HIP kernels, C ABI in
``include/mftx.h``); there is no CPU fallback.
"""
from .config import Config, load_config  # noqa: F401
from .results import FlowOUTrackingResult, FlowOUResult  # noqa: F401


def __getattr__(name):  # lazy: these import torch-heavy modules
    if name in ("MFT", "chain_results", "get_flowou_with_cache"):
        from . import MFT as _m
        return getattr(_m, name)
    if name == "RAFTWrapper":
        from .raft import RAFTWrapper
        return RAFTWrapper
    raise AttributeError(name)
