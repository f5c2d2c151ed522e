#!/usr/bin/env python3
"""cProfile of the host side of the per-frame sharded mode (bench.py --force-sharded --emulate-world 8 --window 1): where the ~1.7 ms of Python
per frame go.   python tools/profile_host_sharded.py [world] [window]"""
import cProfile
import pstats
import runpy
import sys
from pathlib import Path

world = sys.argv[1] if len(sys.argv) > 1 else "8"
window = sys.argv[2] if len(sys.argv) > 2 else "1"
repo = Path(__file__).resolve().parents[1]
sys.argv = [str(repo / "bench.py"), "--steps", "60", "--warmup", "8", "--force-sharded", "--emulate-world", world, "--window", window,
            "--no-cpu-baseline", "--no-parity", "--no-alt-arith", "--no-profile"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(str(repo / "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("cumulative").print_stats("mft_amd|bench.py:4", 40)
st.sort_stats("tottime").print_stats("mft_amd|torch/cuda|built-in|method", 40)
for fn in ("_refine_on_lane", "_start_window", "_finish_frame", "_start_feature_exchange", "compute_pairs"):
    st.sort_stats("cumulative").print_callees(fn)
