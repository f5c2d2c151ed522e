"""Default tracker configuration -- same fields and values as the reference's
``configs/MFT_cfg.py`` (deltas, occlusion threshold, flow plugin)."""
from pathlib import Path

import numpy as np

from mft_amd.MFT import MFT
from mft_amd.config import Config, load_config


def get_config():
    conf = Config()
    conf.tracker_class = MFT
    conf.flow_config = load_config(Path(__file__).parent / "flow" / "RAFTou_kubric_huber_split_nonoccl.py")
    conf.deltas = [np.inf, 1, 2, 4, 8, 16, 32]
    conf.occlusion_threshold = 0.02
    conf.name = Path(__file__).stem
    return conf
