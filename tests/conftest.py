import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
for p in (REPO, REPO / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return REPO / "tests" / "golden"


@pytest.fixture(scope="session")
def weights_np():
    from mft_amd.weights import make_weights
    import golden_inputs as gi
    return make_weights(gi.WEIGHT_SEED)


@pytest.fixture(scope="session")
def weights_cpu(weights_np):
    import torch
    return {k: torch.from_numpy(v) for k, v in weights_np.items()}
