import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
B = int(os.environ.get("CB", "64"))
bench("conv3x3", B, 64, 64, 64)
bench("conv3x3", B, 32, 128, 128)
bench("conv3x3", B, 16, 256, 256)
