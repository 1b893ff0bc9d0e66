"""tgp.pytorch_amd -- MI355X-native implementation of the TGP sparse-variational ELBO hot path.

Host-side mirror of the reference's model-class API (jmaronas/TGP.pytorch, code/dsp) on top of
hand-written gfx950 HIP kernels reached through a C ABI (include/tgp_hip.h, libtgp_hip.so).
"""
from . import lib, ops  # noqa: F401

__all__ = ["lib", "ops"]
