"""Step engine: the reference's inner training loop (code/dsp/trainers/trainer_base.py:329-349 --
ELBO -> (-ELBO).backward() -> Adam.step()) with everything resident on the GPU.

* every trainable tensor is a view into ONE flat float64 buffer (parameters, gradients, Adam moments),
  so the optimiser is a single launch and the multi-GPU exchange is a single all-reduce;
* the C-ABI argument structs are built once (pointers never change), so one step is
  `tgp_elbo_step_adam_f64` = 4 kernel launches (M <= 128) with no host synchronisation, plus
  the MLP forward/backward launches for input-dependent flows;
* `capture()` records that sequence into a HIP graph (torch.cuda.CUDAGraph is only the stream/graph
  plumbing) and `replay()` re-launches it;
* multi-GPU (one process per GPU): rows are sharded, every rank runs the same M x M work, and ONE
  all-reduce(sum) over RCCL/xGMI of [gradients | ELL | KL/W] makes every rank's buffer complete
  (SURVEY.md 8e); the KL gradient enters each rank with weight 1/world so the sum counts it once.
"""
import ctypes as C

import os

import torch

# Stream-capture mode of every graph this package records.  "thread_local", not torch's default "global": with an NCCL
# (= RCCL) process group alive, its watchdog thread polls the events of earlier collectives with hipEventQuery, which the
# global mode forbids while ANY stream of the process captures -- measured on this stack (ROCm 7.0 / RCCL 2.26,
# tools/probes/rccl_smoke.py): the process dies with hipErrorStreamCaptureUnsupported; thread-local capture records the
# same graph (collective included) and replays it.
CAPTURE_MODE = os.environ.get("TGP_CAPTURE_MODE", "thread_local")   # the override exists to reproduce the failure

from . import lib as L
from . import ops

ORDER = ("Z", "raw_ls", "raw_os", "m", "Lam", "lvn", "theta", "nn")   # "nn" (packed MLP weights) last: the weight-decay group


def allreduce_flat(grad, n, world_size, group=None):
    """The ONE exchange of the data-parallel step: all-reduce(sum) of [gradients(n) | ELBO, ELL, KL, 0].
    Every rank enters with gradients of (ELL_shard - KL/world) and out = [ELL_shard - KL, ELL_shard, KL, 0]; KL is
    identical on all ranks, so it is pre-divided and the sum restores it; ELBO is rebuilt from the reduced parts.
    Works on any backend (RCCL on the GPUs, gloo in the CPU tests)."""
    if world_size > 1:
        pre_reduce(grad, n, world_size)
        torch.distributed.all_reduce(grad, op=torch.distributed.ReduceOp.SUM, group=group)
        post_reduce(grad, n)
    return grad


class RcclUnavailable(RuntimeError):
    """The ABI communicator could not be built on at least one rank (raised on EVERY rank of the group, so that all of
    them take the same fallback branch)."""


COMM_TIMEOUT_S = 120.0     # default bound of the communicator bootstrap (RcclComm(timeout_s=...), ElboEngine(comm_timeout_s=...))
_COMM_GENERATION = {}      # communicators built per group (key = the group's global ranks), counted by the group's rank 0


def _store_rendezvous(tag, world_size, timeout_s):
    """All ranks of the default process group's store arrive at `tag`, or TimeoutError after timeout_s: a rank that died
    on the way makes the others FAIL here instead of hanging inside ncclCommInitRank (VERDICT r4 #5a)."""
    import time
    try:
        store = torch.distributed.distributed_c10d._get_default_store()
    except Exception:
        return                                   # no store to meet at (a hand-made group): nothing to bound with
    store.add(tag, 1)
    t0 = time.time()
    while int(store.add(tag, 0)) < world_size:
        if time.time() - t0 > timeout_s:
            raise TimeoutError("only %d of %d ranks reached %s within %.0f s" % (int(store.add(tag, 0)), world_size, tag, timeout_s))
        time.sleep(0.005)


class RcclComm:
    """The C ABI's collective (tgp_comm_* / tgp_allreduce_f64: RCCL bound at run time, all-reduce on the CALLER's stream).
    One per process.  The group's rank 0 draws the 128-byte id; with more than one rank it travels through
    torch.distributed's object broadcast (any backend: it is a host-side exchange), so the bootstrap needs no second
    rendezvous.  The bootstrap is GUARDED (VERDICT r4 #5a): (1) the local steps (dlopen of RCCL, the id) never
    raise on their own -- every rank first learns through one MIN all-reduce whether ALL of them succeeded, and on a
    failure every rank raises RcclUnavailable together (nobody is left waiting in a broadcast); (2) the id is broadcast
    from the GLOBAL rank of the group's rank 0; (3) before ncclCommInitRank the ranks meet at the process group's store
    with a timeout, and the init call itself runs under a watchdog: a rank that cannot complete it raises TimeoutError
    (tgp_last_error() in the message) instead of hanging -- the caller exits non-zero, nothing re-execs."""

    def __init__(self, world_size=1, rank=0, group=None, timeout_s=None, uid=None):
        import ctypes as C
        import threading
        timeout_s = COMM_TIMEOUT_S if timeout_s is None else float(timeout_s)
        self.comm = None
        self.lib = L.load()
        dist = torch.distributed
        multi = world_size > 1 and dist.is_initialized() and uid is None
        if world_size > 1 and not multi and uid is None:
            #  a hand-made world: nothing here can carry rank 0's id to the others, and ncclCommInitRank with an
            # all-zero id never returns -- say so instead of hanging
            raise RcclUnavailable("RcclComm(world_size=%d) needs torch.distributed initialised (the id travels through its "
                                  "object broadcast) or the 128-byte id of rank 0 passed as `uid`" % world_size)
        given = uid
        uid = (C.c_char * 128)()
        if given is not None:
            uid.raw = bytes(given)[:128].ljust(128, b"\0")
        err = None
        try:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")   # the RCCL this process already holds
            L.check(self.lib.tgp_comm_load(path.encode() if os.path.exists(path) else None), "tgp_comm_load")
            if rank == 0 and given is None:
                L.check(self.lib.tgp_comm_unique_id(C.cast(uid, C.c_void_p)), "tgp_comm_unique_id")
        except Exception as e:                      # (kept: all ranks must reach the agreement below)
            err = e
        if multi:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if float(ok.cpu()) < 0.5:
                raise RcclUnavailable("tgp_comm bootstrap failed on %s: %s" % ("this rank" if err is not None else "another rank",
                                                                              err if err is not None else "see its log"))
            src = dist.get_global_rank(group, 0) if group is not None else 0
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            key = "-".join(str(r) for r in ranks)
            # the rendezvous tag's generation is rank 0's count FOR THIS GROUP and travels with the id (a process-global
            # counter differs between ranks that are members of different sub-groups, and equal tags are what the ranks meet at)
            if rank == 0:
                _COMM_GENERATION[key] = _COMM_GENERATION.get(key, 0) + 1
            box = [bytes(uid.raw), _COMM_GENERATION.get(key, 0)]
            dist.broadcast_object_list(box, src=src, group=group)
            uid.raw = box[0]
            tag = "tgp_comm_init/%d/%s" % (int(box[1]), key)
            _store_rendezvous(tag, len(ranks), timeout_s)
        elif err is not None:
            raise RcclUnavailable(str(err))
        comm = C.c_void_p()
        res = {}

        def _init(device=torch.cuda.current_device() if torch.cuda.is_available() else None):
            try:
                if device is not None:
                    torch.cuda.set_device(device)    # (the watchdog thread starts on device 0: RCCL binds to the current one)
                res["rc"] = self.lib.tgp_comm_init(C.cast(uid, C.c_void_p), int(world_size), int(rank), C.byref(comm))
                res["err"] = self.lib.tgp_last_error().decode(errors="replace") if res["rc"] != 0 else ""
            except Exception as e:
                res["exc"] = e

        if world_size > 1:                           # any multi-rank init can block on a missing peer: always under the watchdog
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(timeout_s)
            if th.is_alive():
                raise TimeoutError("tgp_comm_init (ncclCommInitRank, %d ranks) did not return within %.0f s on rank %d; "
                                   "tgp_last_error(): %s" % (world_size, timeout_s, rank,
                                                             self.lib.tgp_last_error().decode(errors="replace")))
        else:
            _init(None)
        if "exc" in res:
            raise res["exc"]
        if res.get("rc", -1) != 0:
            raise RuntimeError("tgp_comm_init failed (%d): %s" % (res.get("rc", -1), res.get("err", "")))
        self.comm = comm
        self.world_size, self.rank = int(world_size), int(rank)

    def allreduce(self, buf, n=None):
        """In-place sum of the first n doubles of `buf` over the ranks, on the current stream (capturable)."""
        n = buf.numel() if n is None else int(n)
        L.check(self.lib.tgp_allreduce_f64(self.comm, L.ptr(buf), n, L.stream_ptr()), "tgp_allreduce_f64")
        return buf

    def close(self):
        if self.comm:
            L.check(self.lib.tgp_comm_destroy(self.comm), "tgp_comm_destroy")
            self.comm = None


def choose_collective(requested, world_size, backend):
    """Which all-reduce a data-parallel engine starts with, and whether it has to prove itself first.
    Returns (candidate, needs_selfcheck): candidate in {"torch", "abi"}.
      "torch": torch.distributed.all_reduce between two graphs (any backend; one Python iteration per step);
      "abi"  : tgp_allreduce_f64 on the compute stream, a node of ONE captured graph (U steps per launch).
    `requested`: None / "auto" = pick; "torch" / "abi" = the caller's choice, taken as is.  "auto" proposes the ABI
    collective wherever RCCL is the transport (backend "nccl") and more than one rank exists -- subject to the
    self-check of ElboEngine._selfcheck_collective -- and torch.distributed everywhere else (gloo rehearsals, one rank)."""
    if requested in ("torch", "abi"):
        return requested, False
    if requested not in (None, "auto"):
        raise ValueError("collective must be None, 'auto', 'torch', 'abi' or an RcclComm")
    if backend == "nccl" and (world_size > 1 or requested == "auto"):
        return "abi", True
    return "torch", False


def pre_reduce(grad, n, world_size):
    """KL is identical on every rank: pre-divide so that the sum restores it."""
    grad[n + 2].div_(world_size)


def post_reduce(grad, n):
    """ELBO from the reduced parts."""
    grad[n] = grad[n + 1] - grad[n + 2]


def shard_rows(N, world_size, rank):
    """Contiguous row shard [lo, hi) of rank `rank`."""
    return (N * rank) // world_size, (N * (rank + 1)) // world_size


class FlatParams:
    """[Z | raw_ls | raw_os | m | Lam | lvn | theta] as views of one buffer (plus same-shaped grads/moments)."""

    def __init__(self, tensors, device):
        self.shapes = {k: tuple(tensors[k].shape) for k in ORDER if tensors.get(k) is not None}
        self.sizes = {k: int(torch.tensor(self.shapes[k]).prod()) if len(self.shapes[k]) else 1 for k in self.shapes}
        n = sum(self.sizes.values())
        self.n = n
        self.extra = 4  # [ELBO, ELL, KL, 0] ride at the end of the gradient buffer so one all-reduce carries them
        self.data = torch.zeros(n, dtype=torch.float64, device=device)
        self.grad = torch.zeros(n + self.extra, dtype=torch.float64, device=device)
        self.exp_avg = torch.zeros(n, dtype=torch.float64, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float64, device=device)
        self.offsets = {}
        o = 0
        for k in ORDER:
            if k in self.shapes:
                self.offsets[k] = o
                self.view(k).copy_(tensors[k].to(device=device, dtype=torch.float64))
                o += self.sizes[k]

    def view(self, k, buf=None):
        buf = self.data if buf is None else buf
        o = self.offsets[k]
        return buf[o:o + self.sizes[k]].view(self.shapes[k])

    def gview(self, k):
        return self.view(k, self.grad)

    @property
    def out(self):
        return self.grad[self.n:self.n + self.extra]


class ElboEngine:
    def __init__(self, X, Y, params, N_total, flow_blocks=None, S=None, rowp=None, lr=0.01, betas=(0.9, 0.999),
                 eps=1e-8, device="cuda:0", world_size=1, rank=0, mb_global=None, process_group=None,
                 kernel="scale_rbf", mlp=None, mlp_weights=None, nn_weight_decay=1e-5, mlp_training=True,
                 jitter_ladder=1e-8, share=None, collective=None, plan=0, comm_timeout_s=None):
        """`mlp` (ops.MlpSpec) + `mlp_weights` (packed, nnets * weights_per_net): input-dependent flow (ID_TGP) whose
        per-row parameters come from the HIP MLP kernels inside the step; `nn_weight_decay` is the reference's Adam
        group for the 'NNets' parameters (main.py:276-288).  `share` = another ElboEngine of the same model whose flat
        parameter / gradient / Adam buffers and step counter this one uses (two batch sizes of one training run)."""
        self.lib = L.load()
        self.device = torch.device(device)
        self.world_size, self.rank, self.pg = int(world_size), int(rank), process_group
        # `collective`: "torch" (default) = torch.distributed.all_reduce between two graphs; "abi" = tgp_allreduce_f64 on the
        # compute stream, INSIDE the captured step (one graph, U steps per launch like a single rank's) -- an RcclComm or
        # the string (a communicator is then created here; torch.distributed only carries its 128-byte id).  With "abi" a
        # 1-rank engine runs the collective too (sum over one rank): the path the GPU tests can reach on one GPU.
        # Default (None / "auto"): with more than one rank over RCCL the engine builds the ABI communicator, reduces a seeded
        # buffer through BOTH paths and keeps the ABI collective only if every rank saw the same sums (choose_collective,
        # _selfcheck_collective); any other situation keeps torch.distributed.  `collective_info` records the decision.
        collective = collective if collective is not None else os.environ.get("TGP_COLLECTIVE") or None
        self.comm, self._own_comm = None, False
        self.collective_info = {"collective": "torch", "selfcheck": "skipped", "why": "one rank"}
        if isinstance(collective, RcclComm):
            self.comm = collective
            self.collective_info = {"collective": "abi", "selfcheck": "skipped", "why": "communicator passed in"}
        else:
            backend = torch.distributed.get_backend(process_group) if torch.distributed.is_initialized() else None
            cand, check = choose_collective(collective, self.world_size, backend)
            self.collective_info = {"collective": cand, "selfcheck": "skipped",
                                    "why": "requested" if collective in ("torch", "abi") else "backend %s, %d rank(s)" % (backend, self.world_size)}
            if cand == "abi":
                # (RCCL binds the communicator and the stream to the CURRENT device: make that this engine's)
                with torch.cuda.device(self.device):
                    # (the communicator spans the process group that carries its id, whatever `world_size` weights KL with)
                    cw = torch.distributed.get_world_size(process_group) if torch.distributed.is_initialized() else self.world_size
                    cr = torch.distributed.get_rank(process_group) if torch.distributed.is_initialized() else self.rank
                    try:
                        self.comm = RcclComm(cw, cr, process_group, timeout_s=comm_timeout_s)
                        self._own_comm = True
                    except RcclUnavailable as e:
                        # every rank raised together (RcclComm's agreement step): with the automatic choice all of them
                        # keep torch.distributed; an explicit "abi" request is an error
                        if not check:
                            raise
                        import warnings
                        warnings.warn("ABI collective unavailable (%s): keeping torch.distributed.all_reduce" % e)
                        self.comm, self._own_comm = None, False
                        self.collective_info.update(collective="torch", selfcheck="skipped", why="bootstrap failed: %s" % e)
                    if check and self.comm is not None:
                        ok, why = self._selfcheck_collective(process_group)
                        self.collective_info.update(selfcheck="pass" if ok else "fail", why=why)
                        if not ok:
                            import warnings
                            warnings.warn("tgp_allreduce_f64 failed its self-check (%s): keeping torch.distributed.all_reduce" % why)
                            self.close()
                            self.collective_info["collective"] = "torch"
        self.X = X.to(self.device, torch.float64).contiguous()
        self.Y = Y.reshape(-1).to(self.device, torch.float64).contiguous()
        self.N, self.D = self.X.shape
        names = {"Z": "Z", "raw_ls": "raw_lengthscale", "raw_os": "raw_outputscale", "m": "m", "Lam": "Lam",
                 "lvn": "log_var_noise", "theta": "theta"}
        tensors = {k: params.get(v, params.get(k)) for k, v in names.items()}
        self.mlp, self.mlp_training, self.nn_wd = mlp, bool(mlp_training), float(nn_weight_decay)
        if mlp is not None:
            tensors["nn"] = mlp_weights.reshape(-1)
            assert tensors["nn"].numel() == mlp.nnets * mlp.weights_per_net
            assert rowp is None, "per-row parameters come from the MLPs"
            rowp = torch.zeros(self.N, mlp.nnets, dtype=torch.float64)
        tensors["raw_ls"] = tensors["raw_ls"].reshape(-1)
        tensors["raw_os"] = tensors["raw_os"].reshape(-1)
        tensors["lvn"] = tensors["lvn"].reshape(-1)
        self.fp = FlatParams(tensors, self.device) if share is None else share.fp
        self.M = self.fp.sizes["m"]
        self.flow = None
        self.S = int(S) if S else 1
        P = self.fp.sizes.get("theta", 0)
        self.rowp = rowp.to(self.device, torch.float64).contiguous() if rowp is not None else None
        RP = self.rowp.shape[1] if self.rowp is not None else 0
        if flow_blocks is not None:
            self.flow = ops.FlowSpec(flow_blocks, P, RP, self.device)
        self.g_rowp = torch.zeros_like(self.rowp) if self.rowp is not None else None
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=self.device) if share is None else share.step_dev
        self.status = torch.zeros(8, dtype=torch.int32, device=self.device) if share is None else share.status
        self.pre_step = None        # callables launched (and captured) before / after the step: the minibatch gather
        self.post_step = None
        mbg = mb_global if mb_global is not None else self.N
        scale = float(N_total) / float(mbg)
        fp = self.fp
        self.md, self._keep = ops._model_struct(self.X, fp.view("Z"), fp.view("raw_ls"), fp.view("raw_os"), fp.view("m"),
                                                fp.view("Lam"), fp.view("lvn"), scale, 0.0, 1.0 / self.world_size,
                                                self.flow, fp.view("theta") if P else None, self.S, kernel, plan)
        # psd_safe_cholesky's retry ladder (dsp/utils.py:256-269) runs on the device inside the captured step (fused path);
        # `jitter_ladder` is its base value (the reference: 1e-8 in float64, or cg.global_jitter), 0 disables it
        self.md.jitter_ladder = float(jitter_ladder or 0.0)
        self._warned_jitter = 0
        self.gs = L.TgpGrads()
        self.gs.Z, self.gs.raw_ls, self.gs.raw_os = (L.ptr(fp.gview(k)) for k in ("Z", "raw_ls", "raw_os"))
        self.gs.m, self.gs.Lam, self.gs.log_var_noise = (L.ptr(fp.gview(k)) for k in ("m", "Lam", "lvn"))
        if P:
            self.gs.theta = L.ptr(fp.gview("theta"))
        if RP:
            self.gs.rowp = L.ptr(self.g_rowp)
        self.ws = ops.workspace(self.N, self.D, self.M, self.md.S, self.md.nblk, self.md.P, self.md.RP, self.device,
                                self.md.kernel, plan)
        self.mlp_ws = None
        if self.mlp is not None:
            d = self.mlp.struct(self.N, True)
            self.mlp_ws = torch.empty(self.lib.tgp_mlp_workspace_bytes(d) // 8 + 16, dtype=torch.float64, device=self.device)
        # One rank, shared flow parameters only: ELBO step + Adam in ONE C-ABI call (tgp_elbo_step_adam_f64: on the fused
        # path the update rides in the last two backward launches -- one launch and one pass over the buffers less)
        self.fused_adam = (self.world_size == 1 and self.comm is None and self.mlp is None
                           and os.environ.get("TGP_FUSED_ADAM", "1") != "0")
        self.ad = L.TgpAdamArgs()
        self.ad.params, self.ad.grads = L.ptr(fp.data), L.ptr(fp.grad)
        self.ad.exp_avg, self.ad.exp_avg_sq = L.ptr(fp.exp_avg), L.ptr(fp.exp_avg_sq)
        # (with per-row networks the flat buffer's tail is their weights: a group of its own -- weight decay, second stream;
        #  the call then updates the prefix only, from the rotated unit's backward phase)
        self.ad.n = fp.offsets["nn"] if self.mlp is not None else fp.n
        self.ad.step_dev, self.ad.maximize = L.ptr(self.step_dev), 1
        self._side = None
        self.pipeline_steps = True      # ID_TGP, one rank: capture the rotated unit (see capture())
        self._out = None                # redirected scalar output while the unrolled graph is being captured
        self.unroll = 1                 # steps per replay of self.gU (capture())
        self.gU = None
        self.hist_u = None
        self.unroll_long, self.gL, self.hist_l = 1, None, None      # a longer unrolled graph for long runs (capture())
        self.graph = None
        self._warm = False

    def _selfcheck_collective(self, group, n=4099, seed=1234):
        """First contact with the node: the same seeded buffer (different on every rank) summed through tgp_allreduce_f64
        and through torch.distributed.all_reduce; the ABI collective is kept only if EVERY rank got the same sums to
        1e-15 relative (a ring and a tree add in different orders: not bitwise).  One host sync, before any capture."""
        g = torch.Generator(device="cpu").manual_seed(seed + self.rank)
        ref = torch.randn(n, generator=g, dtype=torch.float64).to(self.device)
        a, b = ref.clone(), ref.clone()
        self.comm.allreduce(a, n)
        torch.distributed.all_reduce(b, op=torch.distributed.ReduceOp.SUM, group=group)
        err = float(((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).cpu())
        ok = torch.tensor([1.0 if err <= 1e-15 * max(self.world_size, 1) else 0.0], dtype=torch.float64, device=self.device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=group)
        good = bool(ok.item() > 0.5)
        return good, "max rel diff %.1e on rank %d, all ranks %s" % (err, self.rank, "agree" if good else "do NOT agree")

    def close(self):
        """Destroy the communicator this engine created (one it was handed stays its owner's)."""
        if self.comm is not None and self._own_comm:
            with torch.cuda.device(self.device):
                self.comm.close()
        if self._own_comm:
            self.comm, self._own_comm = None, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one step, eager launches ------------------------------------------------------------------
    def elbo(self, phases=7):
        out = self.fp.out if self._out is None else self._out      # (ELBO, ELL, KL) of this step: see capture(unroll)
        rc = self.lib.tgp_elbo_step_phases_f64(self.md, L.ptr(self.X), L.ptr(self.Y), L.ptr(self.rowp),
                                               L.ptr(out), self.gs, None, None, L.ptr(self.status),
                                               L.ptr(self.ws), self.ws.numel() * 8, phases, L.stream_ptr())
        L.check(rc, "tgp_elbo_step_phases_f64")
        self._warm = True

    def step_adam(self, phases=0):
        """forward_backward() + adam() as one call (fused_adam engines): same kernels' arithmetic, same results.
        `phases` != 0: only those phases (the update rides in the backward phase)."""
        out = self.fp.out if self._out is None else self._out
        self.ad.lr, self.ad.beta1, self.ad.beta2, self.ad.eps = self.lr, self.betas[0], self.betas[1], self.eps
        self.ad.phases = phases
        rc = self.lib.tgp_elbo_step_adam_f64(self.md, L.ptr(self.X), L.ptr(self.Y), L.ptr(self.rowp), L.ptr(out), self.gs,
                                             None, None, L.ptr(self.status), L.ptr(self.ws), self.ws.numel() * 8, self.ad,
                                             L.stream_ptr())
        L.check(rc, "tgp_elbo_step_adam_f64")
        self._warm = True

    def allreduce(self):
        if self.comm is not None:
            pre_reduce(self.fp.grad, self.fp.n, self.world_size)
            with torch.cuda.device(self.device):
                self.comm.allreduce(self.fp.grad, self.fp.n + self.fp.extra)
            post_reduce(self.fp.grad, self.fp.n)
            return
        allreduce_flat(self.fp.grad, self.fp.n, self.world_size, self.pg)

    def adam(self):
        n_plain = self.fp.offsets["nn"] if self.mlp is not None else self.fp.n
        rc = self.lib.tgp_adam_dev_groups_f64(L.ptr(self.fp.data), L.ptr(self.fp.grad), L.ptr(self.fp.exp_avg),
                                              L.ptr(self.fp.exp_avg_sq), self.fp.n, self.lr, self.betas[0], self.betas[1],
                                              self.eps, n_plain, self.nn_wd if self.mlp is not None else 0.0,
                                              L.ptr(self.step_dev), 1, L.stream_ptr())
        L.check(rc, "tgp_adam_dev_groups_f64")

    def mlp_forward(self, step=None):
        if self.mlp is not None:
            d = self.mlp.struct(self.N, self.mlp_training)
            L.check(self.lib.tgp_mlp_forward_f64(d, L.ptr(self.X), L.ptr(self.fp.view("nn")),
                                                 L.ptr(self.step_dev if step is None else step), L.ptr(self.rowp),
                                                 L.stream_ptr()), "tgp_mlp_forward_f64")

    def mlp_backward(self, step=None):
        if self.mlp is not None:
            d = self.mlp.struct(self.N, self.mlp_training)
            L.check(self.lib.tgp_mlp_backward_f64(d, L.ptr(self.X), L.ptr(self.fp.view("nn")),
                                                  L.ptr(self.step_dev if step is None else step), L.ptr(self.g_rowp),
                                                  L.ptr(self.fp.gview("nn")), L.ptr(self.mlp_ws), self.mlp_ws.numel() * 8,
                                                  L.stream_ptr()), "tgp_mlp_backward_f64")

    def mlp_backward_adam(self, step):
        """MLP backward + Adam on the network weights (their own group: weight decay, own step counter) in the launch that
        reduces the weight gradients (tgp_mlp_backward_adam_f64): one launch less on the side chain of the rotated unit."""
        fp = self.fp
        lo, hi = fp.offsets["nn"], fp.n
        ad = L.TgpAdamArgs()
        ad.params, ad.grads = L.ptr(fp.data[lo:hi]), L.ptr(fp.grad[lo:hi])
        ad.exp_avg, ad.exp_avg_sq = L.ptr(fp.exp_avg[lo:hi]), L.ptr(fp.exp_avg_sq[lo:hi])
        ad.n = hi - lo
        ad.lr, ad.beta1, ad.beta2, ad.eps = self.lr, self.betas[0], self.betas[1], self.eps
        ad.step_dev, ad.maximize, ad.phases = L.ptr(step), 1, 0
        d = self.mlp.struct(self.N, self.mlp_training)
        L.check(self.lib.tgp_mlp_backward_adam_f64(d, L.ptr(self.X), L.ptr(fp.data[lo:hi]), L.ptr(step), L.ptr(self.g_rowp),
                                                   L.ptr(fp.grad[lo:hi]), L.ptr(self.mlp_ws), self.mlp_ws.numel() * 8, ad,
                                                   float(self.nn_wd), L.stream_ptr()), "tgp_mlp_backward_adam_f64")

    def _adam_segment(self, lo, hi, weight_decay, step):
        fp = self.fp
        rc = self.lib.tgp_adam_dev_f64(L.ptr(fp.data[lo:hi]), L.ptr(fp.grad[lo:hi]), L.ptr(fp.exp_avg[lo:hi]),
                                       L.ptr(fp.exp_avg_sq[lo:hi]), hi - lo, self.lr, self.betas[0], self.betas[1], self.eps,
                                       float(weight_decay), L.ptr(step), 1, L.stream_ptr())
        L.check(rc, "tgp_adam_dev_f64")

    def forward_backward(self):
        """MLPs -> fused ELBO step -> MLP backward: every gradient of the flat buffer is written.

        With MLPs the two halves that do not depend on each other run on a side stream: the MLP forward (needs X and
        the weights only) under the M x M prepare phase (K_MM, Cholesky, KL), and the MLP backward (needs d/d rowp from
        the row kernel only) under the M x M adjoint.  Fork and join are event waits, valid under graph capture."""
        if self.mlp is None:
            self.elbo()
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        # (the main chain of each fork is issued first: it is the longer one, and under graph replay ready nodes are
        #  dispatched in creation order -- see capture())
        forked = torch.cuda.Event()
        forked.record(main)
        self.elbo(1)                 # prepare
        side.wait_event(forked)
        with torch.cuda.stream(side):
            self.mlp_forward()
        main.wait_stream(side)
        self.elbo(2)                 # rows: consumes rowp, produces g_rowp
        forked = torch.cuda.Event()
        forked.record(main)
        self.elbo(4)                 # M x M adjoint + gradient assembly
        side.wait_event(forked)
        with torch.cuda.stream(side):
            self.mlp_backward()
        main.wait_stream(side)

    def step(self):
        if self.graph == "rotated":      # the captured unit straddles two steps: eager steps would repeat half of one
            return self.replay()
        if self.pre_step is not None:
            self.pre_step()
        if self.fused_adam:
            self.step_adam()
        else:
            self.forward_backward()
            self.allreduce()
            self.adam()
        if self.post_step is not None:
            self.post_step()

    # ---- HIP graph ----------------------------------------------------------------------------------
    def capture(self, with_allreduce=None, unroll=None):
        """Capture one full step.  With world_size > 1 the all-reduce stays outside (two graphs) unless
        with_allreduce=True."""
        if not self._warm:
            self.forward_backward()          # first launch sets kernel attributes; must happen outside capture
        torch.cuda.synchronize()
        pre = self.pre_step if self.pre_step is not None else (lambda: None)
        post = self.post_step if self.post_step is not None else (lambda: None)
        if (self.mlp is not None and self.world_size == 1 and self.comm is None and self.pipeline_steps
                and self.pre_step is None):
            # Rotated unit.  The long pole of an ID_TGP step is the MLP backward (97 us), and nothing of step t depends on it
            # except the network weights' own Adam update and the NEXT step's MLP forward.  So the captured unit starts
            # at the row kernel:   main: rows(t) -> M x M adjoint(t) -> Adam(GP params) -> prepare(t+1)
            #                      side:            MLP backward(t) -> Adam(network weights) -> MLP forward(t+1)
            # joined at the end.  Same operations in the same per-step order as step(); prepare(0) and MLP forward(0)
            # run here, eagerly, once.  Two device step counters (one per Adam group; the dropout masks follow the
            # network group's) keep both launches graph-replayable.
            n_plain = self.fp.offsets["nn"]
            self.step_nn = self.step_dev.clone()
            self.elbo(1)
            self.mlp_forward(self.step_nn)
            torch.cuda.synchronize()
            def unit():
                main = torch.cuda.current_stream()
                side = self._side
                self.elbo(2)
                # The main chain is issued BEFORE the side branch: graph replay dispatches ready nodes in creation order,
                # and with the MLP backward (408 workgroups) created first the slab reduction that heads the critical
                # chain found every CU taken (5 -> 12 us; the whole step 172 -> 150 us with this order).
                forked = torch.cuda.Event()
                forked.record(main)
                if os.environ.get("TGP_FUSED_ADAM", "1") != "0":
                    self.step_adam(4)            # M x M adjoint + gradient assembly + Adam on the GP / flow parameters
                else:
                    self.elbo(4)
                    self._adam_segment(0, n_plain, 0.0, self.step_dev)
                self.elbo(1)
                side.wait_event(forked)
                with torch.cuda.stream(side):
                    if os.environ.get("TGP_FUSED_ADAM", "1") != "0":
                        self.mlp_backward_adam(self.step_nn)     # backward + Adam on the network weights, one launch less
                    else:
                        self.mlp_backward(self.step_nn)
                        self._adam_segment(n_plain, self.fp.n, self.nn_wd, self.step_nn)
                    self.mlp_forward(self.step_nn)
                main.wait_stream(side)
            self.g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, capture_error_mode=CAPTURE_MODE):
                unit()
            self._capture_unrolled(unit, unroll)
            self.graph = "rotated"
            return
        if self.world_size > 1 and not with_allreduce and self.comm is None:
            # [graph 1: step kernels + KL pre-division] -> RCCL all-reduce -> [graph 2: ELBO fix-up + Adam]
            self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, capture_error_mode=CAPTURE_MODE):
                pre()
                self.forward_backward()
                pre_reduce(self.fp.grad, self.fp.n, self.world_size)
            with torch.cuda.graph(self.g2, capture_error_mode=CAPTURE_MODE):
                post_reduce(self.fp.grad, self.fp.n)
                self.adam()
                post()
            self.graph = "split"
        else:
            def unit():
                pre()
                if self.fused_adam:
                    self.step_adam()
                else:
                    self.forward_backward()
                    self.allreduce()
                    self.adam()
                post()
            self.g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, capture_error_mode=CAPTURE_MODE):
                unit()
            if self.world_size == 1 or self.comm is not None:   # (the ABI's all-reduce is a node of the graph like any kernel)
                self._capture_unrolled(unit, unroll)
            self.graph = "full"

    def _capture_unrolled(self, unit, unroll):
        """A second graph of U consecutive steps (one rank only).  Launching a graph costs ~5 us of idle GPU between two
        replays of the launch chain (14 us with the two-stream ID_TGP unit): U steps per launch pay it once (Power TGP
        129.5 -> 125 us per step at U = 8-10, ID_TGP 148 -> 141).  The same kernels in the same order: results are
        bit-identical to U single-step replays.  The scalars of the first U - 1 steps go to self.hist_u (the step's `out`
        argument is redirected: no copy node), the last step's to fp.out as always.
        With the default U (no explicit `unroll`) a THIRD graph of 4 U steps serves long runs (round 5, Power TGP at 96 us per
        step: 10 406 steps/s at 10 steps per launch, 10 470 at 40); replay_many uses the longest graph that still fits, so a
        20-step run is two replays of the U-step graph as before."""
        U = int(unroll if unroll is not None else os.environ.get("TGP_GRAPH_UNROLL", "10"))
        self.unroll, self.gU = 1, None
        self.unroll_long, self.gL, self.hist_l = 1, None, None
        if U < 2 or (unroll is None and self.M > 128):      # general-M steps take milliseconds: nothing to gain, ~100 nodes each
            return
        # With a collective in the unit the step's scalars must stay where allreduce() / pre_reduce / post_reduce act on
        # them -- behind the gradients in fp.grad -- and reach hist_u by a captured 4-double copy AFTER the reduction; the
        # redirected `out` of the single-rank form would log rank-local, un-reduced values (and re-sum a stale slot).
        reduced = self.comm is not None or self.world_size > 1

        def capture_steps(nsteps):
            hist = torch.zeros(nsteps - 1, 4, dtype=torch.float64, device=self.device)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    for u in range(nsteps):
                        self._out = hist[u] if (u < nsteps - 1 and not reduced) else None
                        unit()
                        if reduced and u < nsteps - 1:
                            hist[u].copy_(self.fp.out)
            finally:
                self._out = None        # a failed capture must not leave later steps writing their scalars into the history
            return g, hist

        self.gU, self.hist_u = capture_steps(U)
        self.unroll = U
        UL = int(os.environ.get("TGP_GRAPH_UNROLL_LONG", str(4 * U))) if unroll is None else 0
        if UL > U and not reduced:
            self.gL, self.hist_l = capture_steps(UL)
            self.unroll_long = UL

    def replay_many(self, n, hist=None, row0=0):
        """n steps: replays of the longest unrolled graph that fits, then of the U-step one, the rest one by one.
        hist[row0 + i] <- (ELBO, ELL, KL) of step i (device-to-device copies between replays, nothing synchronises)."""
        k = 0
        for g, U, hu in ((self.gL, self.unroll_long, self.hist_l), (self.gU, self.unroll, self.hist_u)):
            while g is not None and n - k >= U:
                g.replay()
                if hist is not None:
                    hist[row0 + k:row0 + k + U - 1].copy_(hu[:, :3])
                    hist[row0 + k + U - 1].copy_(self.fp.out[:3])
                k += U
        while k < n:
            self.replay()
            if hist is not None:
                hist[row0 + k].copy_(self.fp.out[:3])
            k += 1

    def replay(self):
        if self.graph in ("full", "rotated"):
            self.g1.replay()
        else:
            self.g1.replay()
            torch.distributed.all_reduce(self.fp.grad, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            self.g2.replay()

    # ---- bookkeeping ---------------------------------------------------------------------------------
    def check_status(self):
        """Lazy Cholesky check (one sync): raises like the reference's psd_safe_cholesky would."""
        st = self.status.cpu()
        if int(st[0]) == ops.STATUS_SYNC_TIMEOUT or int(st[3]):    # st[3]: sticky count, survives the later steps of a replayed graph
            ops.raise_for_status(st)
        if int(st[1]):
            raise ops.NanError("cholesky: K_MM contains NaN")
        if int(st[2]) > self._warned_jitter:
            import warnings
            self._warned_jitter = int(st[2])
            warnings.warn("A not p.d., added jitter of %g to the diagonal" % (self.md.jitter_ladder * 10 ** (int(st[2]) - 1)),
                          ops.NumericalWarning)
        if int(st[0]):
            raise ops.NotPSDError("K_MM not positive definite at pivot %d even with the largest jitter of the ladder "
                                  "(%g), like psd_safe_cholesky's final raise (dsp/utils.py:268-269)"
                                  % (int(st[0]), self.md.jitter_ladder * 100))

    def scalars(self):
        o = self.fp.out.cpu()
        return float(o[0]), float(o[1]), float(o[2])

    def set_lr(self, lr):
        self.lr = float(lr)          # a launch argument: re-capture afterwards


class MinibatchEngine:
    """The reference's minibatch loop (trainers/trainer_base.py:322-349 over a DataLoader(batch_size, shuffle,
    drop_last=False), code/main.py:74) with the data set resident in HBM and every step replayed from a HIP graph.

    * X, Y stay on the device; one epoch's row order is an int32 index buffer written once per epoch (`set_order`);
    * a captured step = [tgp_gather_rows_f64: batch rows -> fixed batch buffers, device cursor advanced by the launch]
      -> the ElboEngine step (ELBO + backward + Adam) on those buffers;
    * the ragged last batch of an epoch (N mod B rows) has its own engine + graph on the same flat parameter / Adam
      buffers (the ELL scale N_total / MB follows the batch's own size, sparse_MF_SP.py:623-626);
    * the three logged scalars of every step are copied device-to-device into a history buffer; nothing syncs.
    With world_size > 1 each rank gathers and processes its row shard of every batch (one all-reduce per step)."""

    def __init__(self, X, Y, params, N_total, batch_size, device="cuda:0", world_size=1, rank=0, **engine_kw):
        self.device = torch.device(device)
        self.lib = L.load()
        self.X = X.to(self.device, torch.float64).contiguous()
        self.Y = Y.reshape(-1).to(self.device, torch.float64).contiguous()
        self.N, self.D = self.X.shape
        self.B = int(min(batch_size, self.N))
        self.nfull, self.rest = divmod(self.N, self.B)
        self.steps_per_epoch = self.nfull + (1 if self.rest else 0)
        self.world_size, self.rank = int(world_size), int(rank)
        self.index = torch.arange(self.N, dtype=torch.int32, device=self.device)
        self.cursor = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.has_order = False
        lo, hi = shard_rows(self.B, world_size, rank)
        self.Xb = torch.zeros(hi - lo, self.D, dtype=torch.float64, device=self.device)
        self.Yb = torch.zeros(hi - lo, dtype=torch.float64, device=self.device)
        self.full = ElboEngine(self.Xb, self.Yb, params, N_total, device=self.device, world_size=world_size, rank=rank,
                               mb_global=self.B, **engine_kw)
        self.full.pipeline_steps = False
        self.full.pre_step = self._gather(lo, hi - lo, self.B)
        self.last = None
        if self.rest:
            lo2, hi2 = shard_rows(self.rest, world_size, rank)
            kw = dict(engine_kw)
            kw.pop("mlp_weights", None)
            self.last = ElboEngine(self.Xb[:hi2 - lo2], self.Yb[:hi2 - lo2], params, N_total, device=self.device,
                                   world_size=world_size, rank=rank, mb_global=self.rest, share=self.full,
                                   mlp_weights=(self.full.fp.view("nn") if self.full.mlp is not None else None), **kw)
            self.last.pipeline_steps = False
            self.last.pre_step = self._gather(lo2, hi2 - lo2, self.rest)
        self.fp = self.full.fp

    def _gather(self, offset, nrows, advance):
        def run():
            rc = self.lib.tgp_gather_rows_f64(L.ptr(self.X), L.ptr(self.Y), self.N, self.D,
                                              L.ptr(self.index) if self.has_order else None, L.ptr(self.cursor), offset, nrows,
                                              advance, self.N, L.ptr(self.Xb), L.ptr(self.Yb), L.stream_ptr())
            L.check(rc, "tgp_gather_rows_f64")
        return run

    def set_order(self, perm=None):
        """Row order of the coming epoch (a permutation of range(N), host or device; None = stored order).  Must be
        called before capture() with the kind of order (permuted or not) the run will use: the index pointer is a launch
        argument of the captured gather."""
        if perm is None:
            if self.has_order:
                self.index.copy_(torch.arange(self.N, dtype=torch.int32, device=self.device))
        else:
            self.has_order = True
            self.index.copy_(perm.to(torch.int32), non_blocking=True)
        self.cursor.zero_()

    def capture(self):
        self.full.capture(unroll=1)      # steps are replayed one by one (per-step history copies, the ragged last batch)
        if self.last is not None:
            self.last.capture(unroll=1)

    def run_epoch(self, hist=None, row0=0, replay=True):
        """One pass over the data: nfull full batches + the ragged one.  hist[row0 + i] <- (ELBO, ELL, KL) of step i."""
        for i in range(self.steps_per_epoch):
            eng = self.full if i < self.nfull else self.last
            (eng.replay if (replay and eng.graph is not None) else eng.step)()
            if hist is not None:
                hist[row0 + i].copy_(eng.fp.out[:3])
        return self.steps_per_epoch

    def check_status(self):
        self.full.check_status()

    @property
    def lr(self):
        return self.full.lr

    def set_lr(self, lr):
        self.full.set_lr(lr)
        if self.last is not None:
            self.last.set_lr(lr)
