"""One convolution shape, a few launches: target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.convbench import bench
kind, B, S, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
res = len(sys.argv) > 6 and sys.argv[6] == "res"
bench(kind, B, S, Cin, Cout, res=res, iters=int(os.environ.get("ITERS", "10")))
