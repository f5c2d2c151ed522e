"""mft_amd -- MI355X-native hot path of the MFT dense point tracker.

Drop-in names: ``MFT`` (tracker), ``FlowOUTrackingResult`` / ``FlowOUResult``,
``RAFTWrapper`` (flow plugin), ``chain_results``, ``get_flowou_with_cache``.
Compute runs in ``libmftx.so`` (hand-written gfx950 HIP kernels, C ABI in
``include/mftx.h``); there is no CPU fallback.
"""
import os as _os

# The tracker runs on five or more HIP streams (the caller's, the encoders' / feature exchange's, one per frame in flight, the graph
# proxies); HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a stream waiting for an event blocks the
# queue it shares.  Eight queues cost nothing on one GPU and are worth + 9 % (windows) / + 70 % (per-frame mode) on the sharded path
# (DESIGN.md section 5).  The HIP runtime reads the variable when it initialises, i.e. at the process's first GPU call: importing this
# package before that is enough; an explicit setting in the environment wins.
import sys as _sys


def _hip_already_initialised():
    t = _sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


#: what became of the setting: {"value": what the environment says now, "explicit": it was set by the user, "too_late": the HIP
#: runtime was already initialised in this process when the package was imported, so a value set HERE did not take effect}
HW_QUEUES = {"explicit": "GPU_MAX_HW_QUEUES" in _os.environ, "too_late": False}
if not HW_QUEUES["explicit"] and _hip_already_initialised():
    HW_QUEUES["too_late"] = True
    import warnings as _warnings
    _warnings.warn("mft_amd was imported after the HIP runtime had been initialised (a torch.cuda call came first): "
                   "GPU_MAX_HW_QUEUES=8 cannot take effect any more and the tracker's streams share HIP's default 4 hardware queues "
                   "(-9 % on window-sharded, -40 % on per-frame-sharded tracking; one GPU alone does not care).  Import mft_amd "
                   "before the first GPU call, or export GPU_MAX_HW_QUEUES=8.", RuntimeWarning, stacklevel=2)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
HW_QUEUES["value"] = _os.environ["GPU_MAX_HW_QUEUES"]

from .config import Config, load_config  # noqa: F401,E402
from .results import FlowOUTrackingResult, FlowOUResult  # noqa: F401,E402


def __getattr__(name):  # lazy: these import torch-heavy modules
    if name in ("MFT", "chain_results", "get_flowou_with_cache"):
        import importlib
        _m = importlib.import_module(__name__ + ".MFT")       # (`from . import MFT` would ask this very function for "MFT": recursion)
        globals()["MFT"] = _m.MFT                             # the CLASS, as the docstring promises (importing the submodule bound its own name here)
        return getattr(_m, name)
    if name == "RAFTWrapper":
        from .raft import RAFTWrapper
        return RAFTWrapper
    raise AttributeError(name)
