"""bench.py against another build of the library (A/B of two kernels on ONE box): bench_with_lib.py <libtgp_hip.so> [bench args]"""
import os, runpy, sys
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(R, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
