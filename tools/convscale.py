"""Duration of the three 3x3 levels against the batch size (fixed cost per launch vs per-tile cost)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
for B in (4, 8, 16, 32, 64, 128):
    bench("conv3x3", B, 64, 64, 64)
    bench("conv3x3", B, 32, 128, 128)
    bench("conv3x3", B, 16, 256, 256)
