"""Point-track read-out of a dense result -- ``MFT/point_tracking.py:6-27``."""
from __future__ import annotations

import numpy as np
import torch


def convert_to_point_tracking(MFT_result, queries):
    """MFT_result: FlowOUTrackingResult; queries: (N xy) coordinates on the init frame ->
    (current coordinates (N, 2), current occlusions (N,)) as numpy arrays."""
    if not isinstance(queries, torch.Tensor):
        queries = torch.from_numpy(np.asarray(queries))
    queries = queries.to(torch.float32)
    coords = MFT_result.warp_forward_points(queries)
    _, occl, _ = MFT_result.sample(queries)
    return coords.detach().cpu().numpy(), np.float32(occl.reshape(-1).detach().cpu().numpy())
