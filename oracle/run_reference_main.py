"""Run the reference's OWN main.py, unmodified, on the CPU of the build container (never on the GPU box): the README
command `python main.py --model ID_TGP --dataset power --train_test_seed_split 1 --num_inducing 100` with the
builder-written third-party stand-ins of oracle/shims on sys.path (gpytorch, pytorchlib, ... are absent here).
Purpose: tell apart what the drop-in does differently from what the UNPINNED stand-ins assume -- if the reference's own
training loop on these stand-ins also lands above its README numbers, the offset is in the stand-ins' assumptions
(apply_linear layer order / initialisation), not in the HIP path.
    python oracle/run_reference_main.py ID_TGP power 1 100 [apply_linear order: post|pre] > log
"""
import os
import runpy
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference/code")
warnings.simplefilter("ignore")
import scipy.integrate as _si      # noqa: E402
import torch                       # noqa: E402

_si.cumtrapz = _si.cumulative_trapezoid
model, dataset, split, M = sys.argv[1:5]
if len(sys.argv) > 5:
    os.environ["TGP_SHIM_APPLY_LINEAR_ORDER"] = sys.argv[5]
torch.set_num_threads(int(os.environ.get("REF_THREADS", "6")))
_real = torch.__version__
torch.__version__ = "1.7.0"
import dsp.config   # noqa: E402,F401
torch.__version__ = _real
os.chdir("/root/reference/code")
sys.argv = ["main.py", "--model", model, "--dataset", dataset, "--train_test_seed_split", split, "--num_inducing", M]
try:
    runpy.run_path("/root/reference/code/main.py", run_name="__main__")
except SystemExit:
    pass
