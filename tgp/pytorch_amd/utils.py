"""Small helpers with the reference's names (code/dsp/utils.py)."""
import torch

from . import config as cg


def inv_softplus(x):
    """gpytorch.utils.transforms.inv_softplus."""
    x = torch.as_tensor(x, dtype=torch.get_default_dtype())
    return x + torch.log(-torch.expm1(-x))


def positive_transform(x):
    """dsp/utils.py:39-46 ('exp' is the only transform main.py configures)."""
    if cg.positive_transform == "exp":
        return torch.exp(x)
    if cg.positive_transform == "softplus":
        return torch.log(torch.exp(x) + 1)
    raise NotImplementedError("positive_transform function %s is not implemented." % cg.positive_transform)


def inverse_positive_transform(x):
    if cg.positive_transform == "exp":
        return torch.log(x)
    if cg.positive_transform == "softplus":
        return torch.log(torch.exp(x) - 1.0)
    raise NotImplementedError("inverse_positive_transform for %s is not implemented." % cg.positive_transform)


def KMEANS(X, num_Z, n_init=1, seed=None, backend="hip", max_iter=300, tol=1e-4, return_info=False):
    """Inducing-point initialisation (dsp/utils.py:143-159).  The reference calls
    sklearn.cluster.KMeans(n_clusters, init='k-means++', n_init, random_state=seed).fit(X) on the host; `backend='hip'`
    runs the same algorithm on the GPU -- sklearn 1.x's sequence restated: centre X, k-means++ seeding with
    2 + log(k) local trials, Lloyd iterations until the labels repeat or the squared centre shift drops under
    tol * mean(var(X)), best of n_init by inertia -- with the random draws taken from the same
    numpy RandomState(seed) in the same order, so that the result agrees with sklearn's up to floating-point
    rounding of the distances (tests/test_gpu_models.py).  `backend='sklearn'` is the reference's own call."""
    if seed is None:
        seed = cg.config_seed
    if backend == "sklearn":
        from sklearn.cluster import KMeans
        km = KMeans(n_clusters=num_Z, init="k-means++", n_init=n_init, random_state=seed).fit(X.to("cpu").numpy())
        return torch.tensor(km.cluster_centers_, dtype=cg.dtype).to(cg.device)
    if backend != "hip":
        raise ValueError("KMEANS backend must be 'hip' or 'sklearn'")
    import numpy as np
    from . import lib as L
    from . import ops
    lib = L.load()
    if not torch.cuda.is_available():
        raise L.TgpError("KMEANS(backend='hip') needs a HIP device; backend='sklearn' is the reference's host path")
    dev = X.device if X.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Xd = X.detach().to(dev, torch.float64).contiguous()
    N, D = Xd.shape
    K = int(num_Z)
    if D > 16:
        raise L.TgpError("KMEANS(backend='hip'): D <= 16")
    st = L.stream_ptr
    tolv = float(Xd.var(dim=0, unbiased=False).mean()) * tol          # sklearn _tolerance (before centring)
    Xmean = Xd.mean(dim=0)
    Xc = (Xd - Xmean).contiguous()                                       # KMeans.fit: X -= X.mean(axis=0)
    rs = np.random.RandomState(seed)                                     # check_random_state(int)
    T = 2 + int(np.log(K))                                               # n_local_trials
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    mind2 = torch.empty(N, dtype=torch.float64, device=dev)
    trial = torch.empty(max(T, 1), N, dtype=torch.float64, device=dev)

    def assign(C):
        L.check(lib.tgp_kmeans_assign_f64(L.ptr(Xc), N, D, L.ptr(C), K, L.ptr(labels), L.ptr(mind2), st()), "tgp_kmeans_assign_f64")

    def seeding():
        """sklearn.cluster._kmeans._kmeans_plusplus with unit sample weights."""
        centers = torch.empty(K, D, dtype=torch.float64, device=dev)
        cid = int(rs.choice(N, p=np.full(N, 1.0 / N)))
        centers[0] = Xc[cid]
        cand = torch.tensor([cid], dtype=torch.int64, device=dev)
        L.check(lib.tgp_kmeans_pp_f64(L.ptr(Xc), N, D, L.ptr(cand), 1, None, L.ptr(trial), st()), "tgp_kmeans_pp_f64")
        closest = trial[0].clone()
        pot = closest.sum()
        for c in range(1, K):
            rv = torch.from_numpy(rs.uniform(size=T)).to(dev) * pot
            cand = torch.searchsorted(torch.cumsum(closest, 0), rv).clamp_(max=N - 1)
            L.check(lib.tgp_kmeans_pp_f64(L.ptr(Xc), N, D, L.ptr(cand), T, L.ptr(closest), L.ptr(trial), st()),
                    "tgp_kmeans_pp_f64")
            pots = trial[:T].sum(dim=1)
            best = torch.argmin(pots)
            pot = pots[best]
            closest = trial[best].clone()
            centers[c] = Xc[cand[best]]
        return centers

    def lloyd(C):
        """sklearn _kmeans_single_lloyd: returns (labels, inertia, centers, n_iter)."""
        C = C.clone()
        labels_old = torch.full((N,), -1, dtype=torch.int32, device=dev)
        strict = False
        it = 0
        for it in range(max_iter):
            assign(C)
            order = torch.argsort(labels, stable=True)
            counts = torch.bincount(labels, minlength=K)
            offs = torch.zeros(K + 1, dtype=torch.int64, device=dev)
            offs[1:] = torch.cumsum(counts, 0)
            sums = torch.empty(K, D, dtype=torch.float64, device=dev)
            L.check(lib.tgp_kmeans_segsum_f64(L.ptr(Xc), D, L.ptr(order), L.ptr(offs), K, L.ptr(sums), st()),
                    "tgp_kmeans_segsum_f64")
            w = counts.to(torch.float64)
            if int((counts == 0).sum()) > 0:
                # sklearn _relocate_empty_clusters_dense: the farthest points become the empty clusters' centres
                empty = torch.nonzero(counts == 0).reshape(-1)
                far = torch.argsort(mind2, descending=True)[: empty.numel()]
                for e, f in zip(empty.tolist(), far.tolist()):
                    old = int(labels[f])
                    sums[old] -= Xc[f]
                    sums[e] = Xc[f]
                    w[e] = 1.0
                    w[old] -= 1.0
            Cn = sums / w.reshape(-1, 1)
            shift2 = float(((Cn - C) ** 2).sum())
            C = Cn
            if torch.equal(labels, labels_old):
                strict = True
                break
            if shift2 <= tolv:
                break
            labels_old.copy_(labels)
        if not strict:
            assign(C)                      # E-step again so that the labels match the returned centres
        inertia = float(((Xc - C[labels.to(torch.int64)]) ** 2).sum())
        return labels.clone(), inertia, C, it + 1

    best = None
    for _ in range(int(n_init)):
        lab, inertia, C, nit = lloyd(seeding())
        if best is None or (inertia < best[1] and not _same_clustering(lab, best[0], K)):
            best = (lab, inertia, C, nit)
    centers = best[2] + Xmean
    Z = centers.to(cg.dtype).to(cg.device)
    if return_info:
        return Z, {"inertia": best[1], "n_iter": best[3], "labels": best[0]}
    return Z


def _same_clustering(a, b, K):
    """sklearn _is_same_clustering: equal up to a permutation of the labels."""
    m = torch.full((K,), -1, dtype=torch.int64, device=a.device)
    a64, b64 = a.to(torch.int64), b.to(torch.int64)
    m[a64] = b64                      # last write wins; consistent mapping <=> m[a] == b everywhere
    return bool(torch.equal(m[a64], b64))


def psd_safe_cholesky(A, upper=False, out=None, jitter=None):
    """dsp/utils.py:222-270 on the GPU (HIP blocked Cholesky + the reference's jitter ladder): (L, A_used)."""
    from . import ops
    if cg.constant_jitter is not None:
        A.diagonal(dim1=-2, dim2=-1).add_(cg.constant_jitter)
    squeeze = A.dim() == 3
    A2 = A[0] if squeeze else A
    L, Ap = ops.psd_safe_cholesky(A2, jitter)
    if upper:
        L = L.transpose(-1, -2)
    return (L.unsqueeze(0), Ap.unsqueeze(0)) if squeeze else (L, Ap)
