"""Flow plugin configuration -- same fields as the reference's
``configs/flow/RAFTou_kubric_huber_split_nonoccl.py``.  ``model`` points to a reference
checkpoint (``module.``-prefixed state_dict); like in the reference a missing file is an error.
The checkpoint is not distributed with this build: benchmarks and tests opt in to seeded
stand-in weights explicitly with ``model = None`` + ``synthetic_weights_seed = <int>``."""
from pathlib import Path

from mft_amd.config import AttrDict, Config
from mft_amd.raft import RAFTWrapper


def get_config():
    conf = Config()
    conf.of_class = RAFTWrapper
    conf.raft_params = AttrDict(occlusion_module="separate_with_uncertainty", small=False, mixed_precision=False)
    conf.model = "checkpoints/raft-things-sintel-kubric-splitted-occlusion-uncertainty-non-occluded-base-sintel.pth"
    conf.flow_iters = 12
    conf.name = Path(__file__).stem
    return conf
