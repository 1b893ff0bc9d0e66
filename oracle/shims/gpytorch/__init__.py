"""Builder-written stand-in for the gpytorch 1.1.1 surface that /root/reference/code/dsp touches.

TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py in the build container to execute the
reference's own Python files). It is NOT gpytorch and holds no reference code.  The arithmetic that
gpytorch owns on the hot path (ARD-RBF x outputscale, softplus constraints, Gauss-Hermite rule,
inv_softplus) is restated here from the published formulas => "parity unpinned" for those pieces
(SURVEY.md section 8c); everything the reference itself computes is executed from its own files.
"""
__version__ = "1.1.1"
from . import utils, kernels, means, lazy, variational  # noqa: E402,F401
