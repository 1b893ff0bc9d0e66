"""CPU: the C-ABI library loads and exports every symbol include/tgp_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "tgp", "pytorch_amd", "libtgp_hip.so")


def declared_symbols():
    text = open(os.path.join(REPO, "include", "tgp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ("tgp_elbo_step_f64", "tgp_qf_moments_f64", "tgp_kmm_f64", "tgp_cholesky_f64", "tgp_kl_whitened_f64",
                 "tgp_ell_gauss_f64", "tgp_ell_flow_f64", "tgp_flow_eval_f64", "tgp_predict_f64", "tgp_adam_f64",
                 "tgp_workspace_bytes", "tgp_version",
                 # SURVEY 8(b)'s minimum export set: the two stand-alone adjoints (round 3)
                 "tgp_qf_moments_bwd_f64", "tgp_cholesky_bwd_f64", "tgp_elbo_step_adam_f64", "tgp_mlp_backward_adam_f64",
                 "tgp_comm_load", "tgp_comm_unique_id", "tgp_comm_init", "tgp_allreduce_f64", "tgp_comm_destroy"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export %s" % s
    lib.tgp_version.restype = ctypes.c_int
    assert lib.tgp_version() >= 100


def test_python_binding_matches_header(lib):
    from tgp.pytorch_amd import lib as L
    assert sorted(L.EXPORTS) == declared_symbols()


def test_argument_errors_are_codes_not_crashes(lib):
    """Null model / null pointers return negative codes before anything touches the device."""
    lib.tgp_kmm_f64.restype = ctypes.c_int
    assert lib.tgp_kmm_f64(None, None, None, 4, 2, ctypes.c_double(0.0), None, None) == -1
    lib.tgp_workspace_bytes.restype = ctypes.c_size_t
    assert lib.tgp_workspace_bytes(8611, 4, 100, 32, 6, 30, 0) > 0
    assert lib.tgp_workspace_bytes(8611, 4, 129, 32, 6, 30, 0) > 0       # M > 128: tiled-GEMM path
    assert lib.tgp_workspace_bytes(8611, 4, 4097, 32, 6, 30, 0) == 0     # M > TGP_BIG_MAX_M unsupported in this build
    assert lib.tgp_workspace_bytes(8611, 17, 100, 32, 6, 30, 0) == 0     # D > 16 unsupported


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under tgp/ may import it."""
    pkg = os.path.join(REPO, "tgp", "pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle in oracle/", ""), fn


def test_workspace_query_covers_both_paths(lib):
    """tgp_workspace_bytes is host-only: sizes for the fused (M <= 128) and the general-M plan, monotone in N."""
    lib.tgp_workspace_bytes.restype = ctypes.c_size_t
    small = lib.tgp_workspace_bytes(8611, 4, 128, 32, 6, 30, 0)
    big = lib.tgp_workspace_bytes(8611, 4, 129, 32, 6, 30, 0)
    assert 0 < small < big
    assert lib.tgp_workspace_bytes(250000, 8, 1000, 32, 5, 130, 0) < 2 * 1024 ** 3      # C5 shard: < 2 GiB of 288 GB
    assert lib.tgp_workspace_bytes(20000, 8, 1000, 32, 5, 130, 0) >= lib.tgp_workspace_bytes(10000, 8, 1000, 32, 5, 130, 0)
    lib.tgp_gemm_f64.restype = ctypes.c_int
    assert lib.tgp_gemm_f64(0, 0, 0, 100, 128, 16, ctypes.c_double(1.0), None, 128, None, 128, ctypes.c_double(0.0), None,
                            128, None) == -4      # m not a multiple of 128: refused before touching the device


def test_host_only_queries_of_the_newer_entry_points(lib):
    """Workspace queries never touch the device: Matern (always the general path), large-M Cholesky, MLP partials."""
    lib.tgp_workspace_bytes_kernel.restype = ctypes.c_size_t
    rbf = lib.tgp_workspace_bytes_kernel(8611, 4, 100, 32, 6, 30, 0, 0)
    mat = lib.tgp_workspace_bytes_kernel(8611, 4, 100, 32, 6, 30, 0, 1)
    assert rbf == lib.tgp_workspace_bytes(8611, 4, 100, 32, 6, 30, 0) and mat > rbf
    lib.tgp_cholesky_workspace_bytes.restype = ctypes.c_size_t
    assert lib.tgp_cholesky_workspace_bytes(100) == 0 and lib.tgp_cholesky_workspace_bytes(1000) > 8 * 1024 * 1024
    from tgp.pytorch_amd import lib as L
    from tgp.pytorch_amd import ops
    spec = ops.MlpSpec(4, 50, 2, 6, act="relu", drop_p=0.25, seed=0)
    assert spec.weights_per_net == 4 * 50 + 50 + 50 * 50 + 50 + 50 + 1 == 2851       # 6 nets of 4 -> 50 -> 50 -> 1
    lib.tgp_mlp_workspace_bytes.restype = ctypes.c_size_t
    lib.tgp_mlp_workspace_bytes.argtypes = [ctypes.POINTER(L.TgpMlp)]
    nb = lib.tgp_mlp_workspace_bytes(spec.struct(8611, True))
    assert nb >= 68 * 6 * 2851 * 8                                                   # one partial per (row block, net)
    assert spec.lds_bytes() < 160 * 1024
    # the dropout mask restated on the host: deterministic, keeps ~1-p, changes with the step
    m1, m2 = ops.mlp_keep_mask(7, 3, 0, 1, 2000, 50, 0.25), ops.mlp_keep_mask(7, 3, 0, 1, 2000, 50, 0.25)
    assert (m1 == m2).all() and abs(m1.mean() - 0.75) < 0.01
    assert (ops.mlp_keep_mask(7, 4, 0, 1, 2000, 50, 0.25) != m1).mean() > 0.2


def test_collective_entry_points_fail_cleanly_without_rccl(lib):
    """RCCL is bound at run time: before tgp_comm_load (or when the path does not exist) the collective entries return
    TGP_E_COMM with a message -- no crash, no GPU call, and nothing else of the library depends on it."""
    import ctypes as C
    lib.tgp_comm_unique_id.restype = C.c_int
    lib.tgp_comm_unique_id.argtypes = [C.c_void_p]
    lib.tgp_comm_load.restype = C.c_int
    lib.tgp_comm_load.argtypes = [C.c_char_p]
    lib.tgp_last_error.restype = C.c_char_p
    buf = (C.c_char * 128)()
    assert lib.tgp_comm_unique_id(C.cast(buf, C.c_void_p)) == -104          # TGP_E_COMM: not loaded
    assert lib.tgp_comm_load(b"/nonexistent/librccl.so") == -104
    assert b"librccl.so" in lib.tgp_last_error()
    assert lib.tgp_comm_unique_id(None) == -1
