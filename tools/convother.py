"""Strided, transposed and 1x1 convolutions of the UNet, for A/B runs over LDM_HIP_LIB builds."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
B = int(os.environ.get("CB", "64"))
bench("conv3x3_s2", B, 64, 64, 128)
bench("conv3x3_s2", B, 32, 128, 256)
bench("convT4x4_s2", B, 16, 256, 128)
bench("convT4x4_s2", B, 32, 128, 64)
bench("conv1x1", B, 32, 256, 128)
bench("conv1x1", B, 64, 128, 64)
bench("conv1x1", B, 16, 256, 768)
bench("conv1x1", B, 16, 256, 256, res=True)
