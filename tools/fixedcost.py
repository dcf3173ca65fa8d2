"""Small launches under rocprofv3 --kernel-trace: GPU-side duration of a launch with little work."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
for B in (1, 4, 16, 64):
    bench("conv3x3", B, 16, 256, 256, iters=10)
