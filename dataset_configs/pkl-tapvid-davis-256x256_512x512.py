"""TAP-Vid-DAVIS, tracked at 512 x 512 on frames resampled 256 x 256 -> 512 x 512, scored at 256 x 256
(BASELINE.json configs[2]; same fields as the reference's dataset config of this name).  The pickle's
location comes from the environment: MFT_TAPVID_DAVIS=/path/to/tapvid_davis.pkl."""
import os
from pathlib import Path

from mft_amd.config import Config


def get_config():
    conf = Config()
    conf.pickles = [Path(os.environ.get("MFT_TAPVID_DAVIS", "tapvid_davis/tapvid_davis.pkl"))]
    conf.scaling = '256x256_512x512'
    conf.name = Path(__file__).stem
    return conf
