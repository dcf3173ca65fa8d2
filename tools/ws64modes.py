"""The 64-channel 3x3 convolution at [B,64,64,64] in its epilogue modes (plain, + residual, SiLU + residual), isolated
launches and a dependent chain of 20 launches: python3 tools/ws64modes.py [B]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.convbench import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bench("conv3x3", B, 64, 64, 64, iters=100)
bench("conv3x3", B, 64, 64, 64, res=True, act="silu", iters=100)
