"""ctypes binding of libtgp_hip.so (C ABI declared in include/tgp_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  PyTorch is used only as the owner of device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtgp_hip.so")

FLOW_AFFINE, FLOW_SAL, FLOW_STEPTANH = 0, 1, 2
FLAG_RESTRICT, FLAG_ADD_F0, FLAG_PER_ROW = 1, 2, 4
LIK_GAUSS, LIK_FLOW = 0, 1

_dp = C.c_void_p


KERNEL_SCALE_RBF, KERNEL_SCALE_MATERN32 = 0, 1
KERNELS = {"scale_rbf": KERNEL_SCALE_RBF, "scale_matern32": KERNEL_SCALE_MATERN32}
# tgp_model.plan (include/tgp_hip.h TGP_PLAN_*): kernel-selection overrides of ONE call; 0 = the library's choice
PLAN_ROWS_AUTO, PLAN_ROWS_K16, PLAN_ROWS_K, PLAN_ROWS4_NW4, PLAN_ROWS4_NW8 = 0, 1, 2, 3, 4
PLAN_NO_CHUNK_OVERLAP = 16


def plan_chunk_rows(n):
    """TGP_PLAN_CHUNK_ROWS(n): row chunks of at most n rows on the general-M path."""
    return (((int(n) + 127) // 128) & 0xffff) << 8


class TgpModel(C.Structure):
    _fields_ = [("N", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("S", C.c_int32), ("nblk", C.c_int32),
                ("P", C.c_int32), ("RP", C.c_int32), ("lik", C.c_int32), ("kernel", C.c_int32), ("plan", C.c_int32),
                ("scale", C.c_double),
                ("jitter", C.c_double), ("kl_scale", C.c_double), ("jitter_ladder", C.c_double), ("Z", _dp), ("raw_ls", _dp), ("raw_os", _dp),
                ("m", _dp), ("Lam", _dp), ("log_var_noise", _dp), ("theta", _dp), ("program", _dp), ("xs", _dp),
                ("wn", _dp)]


class TgpGrads(C.Structure):
    _fields_ = [("Z", _dp), ("raw_ls", _dp), ("raw_os", _dp), ("m", _dp), ("Lam", _dp), ("log_var_noise", _dp),
                ("theta", _dp), ("rowp", _dp)]


class TgpAdamArgs(C.Structure):
    _fields_ = [("params", _dp), ("grads", _dp), ("exp_avg", _dp), ("exp_avg_sq", _dp), ("n", C.c_int64), ("lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step_dev", _dp), ("maximize", C.c_int32),
                ("phases", C.c_uint32)]


class TgpMlp(C.Structure):
    _fields_ = [("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("nnets", C.c_int32),
                ("act", C.c_int32), ("training", C.c_int32), ("reserved0", C.c_int32), ("drop_p", C.c_double),
                ("seed", C.c_uint64)]


class TgpError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "tgp_version": (C.c_int, []),
    "tgp_last_error": (C.c_char_p, []),
    "tgp_source_hash": (C.c_char_p, []),
    "tgp_workspace_bytes": (C.c_size_t, [C.c_int32] * 7),
    "tgp_workspace_bytes_kernel": (C.c_size_t, [C.c_int32] * 8),
    "tgp_workspace_bytes_plan": (C.c_size_t, [C.c_int32] * 9),
    "tgp_kernel_matrix_f64": (C.c_int, [C.c_int32, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, _dp, _dp, C.c_double, _dp, _dp]),
    "tgp_elbo_step_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, C.POINTER(TgpGrads), _dp, _dp, _dp, _dp,
                                    C.c_size_t, _dp]),
    "tgp_elbo_step_phases_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, C.POINTER(TgpGrads), _dp, _dp, _dp,
                                           _dp, C.c_size_t, C.c_uint32, _dp]),
    "tgp_elbo_step_adam_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, C.POINTER(TgpGrads), _dp, _dp, _dp, _dp,
                                         C.c_size_t, C.POINTER(TgpAdamArgs), _dp]),
    "tgp_qf_moments_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, _dp, C.c_size_t, _dp]),
    "tgp_mlp_backward_adam_f64": (C.c_int, [C.POINTER(TgpMlp), _dp, _dp, _dp, _dp, _dp, _dp, C.c_size_t, C.POINTER(TgpAdamArgs),
                                            C.c_double, _dp]),
    "tgp_comm_load": (C.c_int, [C.c_char_p]),
    "tgp_comm_unique_id": (C.c_int, [_dp]),
    "tgp_comm_init": (C.c_int, [_dp, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "tgp_allreduce_f64": (C.c_int, [_dp, _dp, C.c_int64, _dp]),
    "tgp_comm_destroy": (C.c_int, [_dp]),
    "tgp_qf_moments_bwd_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, C.POINTER(TgpGrads), _dp, _dp, C.c_size_t, _dp]),
    "tgp_kmm_f64": (C.c_int, [_dp, _dp, _dp, C.c_int32, C.c_int32, C.c_double, _dp, _dp]),
    "tgp_knm_f64": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "tgp_cholesky_f64": (C.c_int, [_dp, C.c_int32, _dp, _dp, _dp, _dp, C.c_size_t, _dp]),
    "tgp_gemm_f64": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, _dp, C.c_int32,
                               _dp, C.c_int32, C.c_double, _dp, C.c_int32, _dp]),
    "tgp_cholesky_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "tgp_cholesky_bwd_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "tgp_cholesky_bwd_f64": (C.c_int, [_dp, _dp, _dp, C.c_int32, _dp, _dp, C.c_size_t, _dp]),
    "tgp_ell_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "tgp_kl_whitened_f64": (C.c_int, [_dp, _dp, C.c_int32, _dp, _dp, _dp, _dp]),
    "tgp_ell_gauss_f64": (C.c_int, [_dp, _dp, _dp, C.c_int32, _dp, C.c_double, _dp, _dp, _dp, _dp, C.c_size_t, _dp]),
    "tgp_ell_flow_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_size_t,
                                   _dp]),
    "tgp_flow_eval_f64": (C.c_int, [C.POINTER(TgpModel), _dp, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp]),
    "tgp_flow_logdet_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "tgp_flow_logdet_f64": (C.c_int, [C.POINTER(TgpModel), _dp, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, C.c_size_t, _dp]),
    "tgp_predict_f64": (C.c_int, [C.POINTER(TgpModel), _dp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp]),
    "tgp_kmeans_assign_f64": (C.c_int, [_dp, C.c_int32, C.c_int32, _dp, C.c_int32, _dp, _dp, _dp]),
    "tgp_kmeans_segsum_f64": (C.c_int, [_dp, C.c_int32, _dp, _dp, C.c_int32, _dp, _dp]),
    "tgp_kmeans_pp_f64": (C.c_int, [_dp, C.c_int32, C.c_int32, _dp, C.c_int32, _dp, _dp, _dp]),
    "tgp_mlp_workspace_bytes": (C.c_size_t, [C.POINTER(TgpMlp)]),
    "tgp_mlp_forward_f64": (C.c_int, [C.POINTER(TgpMlp), _dp, _dp, _dp, _dp, _dp]),
    "tgp_mlp_backward_f64": (C.c_int, [C.POINTER(TgpMlp), _dp, _dp, _dp, _dp, _dp, _dp, C.c_size_t, _dp]),
    "tgp_gather_rows_f64": (C.c_int, [_dp, _dp, C.c_int32, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp,
                                      _dp, _dp]),
    "tgp_adam_dev_groups_f64": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_int64, C.c_double, _dp, C.c_int32, _dp]),
    "tgp_adam_f64": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_double, C.c_int32, C.c_int32, _dp]),
    "tgp_adam_dev_f64": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, _dp, C.c_int32, _dp]),
}

EXPORTS = tuple(_SIGS.keys())


def source_hash():
    """sha256[:16] of the kernel sources in the tree, computed like the csrc Makefile does (sorted *.hip, *.hpp, then
    include/tgp_hip.h); None when the sources are not next to the library (a binary-only install)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    hdr = os.path.join(_HERE, "..", "..", "include", "tgp_hip.h")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")),
                   key=lambda f: os.path.basename(f))
    if not files or not os.path.exists(hdr):
        return None
    h = hashlib.sha256()
    for f in files + [hdr]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Load libtgp_hip.so once; raise (never fall back) when it is absent or was built from other sources than the
    ones in the tree (the binary is git-ignored and travels beside them: a stale one must not pass for the product)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgpError("libtgp_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C tgp/pytorch_amd/csrc` (%s)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        want, have = source_hash(), lib.tgp_source_hash().decode()
        if want is not None and have != want and not os.environ.get("TGP_ALLOW_STALE_LIB"):
            raise TgpError("libtgp_hip.so is stale: built from sources %s, the tree has %s -- rebuild with "
                           "`make -C tgp/pytorch_amd/csrc`" % (have, want))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().tgp_last_error().decode() if rc == -103 else ""
        raise TgpError("%s failed with code %d %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a contiguous float64/int32 CUDA(HIP) tensor, or NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise TgpError("tensor must live on the GPU (got %s)" % t.device)
    if not t.is_contiguous():
        raise TgpError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
