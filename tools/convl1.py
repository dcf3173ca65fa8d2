"""The three epilogue shapes of the 64-channel level, for A/B runs over LDM_HIP_LIB builds."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
B = int(os.environ.get("CB", "64"))
for _ in range(2):
    bench("conv3x3", B, 64, 64, 64)
    bench("conv3x3", B, 64, 64, 64, res=True, act="silu")
