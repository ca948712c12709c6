"""Model, evaluation and OT grouping with the reference's names (method/utils.py of
the reference), backed by the HIP engine.

  seed_all     utils.py:21-25
  MF           utils.py:30-43    two embedding tables; forward = row-wise dot
  baseTest     utils.py:115-187  ensemble mean score, RMSE, HR@10, NDCG@10
  computeNDCG / computeDCG  utils.py:190-210
  ot_cluster   utils.py:628-656  OT balanced clustering (exact EMD)
  saveObject / loadObject / timefn  utils.py:319-326, 616-626

Training does not go through a `baseTrain(dataloader, model, loss_fn, opt, ...)` loop:
the per-batch forward / backward / SGD step of utils.py:58-91 is one fused kernel
launch per step, driven by engine.TrainJob from Scratch.train.
"""
import ctypes
import os
import pickle
import time
from functools import wraps

import numpy as np
import torch
from torch import nn

from .. import _native as nv
from .. import engine
from ..read import as_loader
from ..rng import seed_all  # noqa: F401  (re-exported under the reference's name)

STD = 1
OT_WARM_ITERS = 400      # dual-ascent steps that warm-start the exact OT solver (ure_ot_potentials) at large n


def ot_warm_iters(n):
    """Fewer steps for small problems: the solver finishes a rough start of a few thousand points in a millisecond, while
    every ascent step is two launches (n = 6,040, ten rounds, the ascent starting from the round before's potentials: at least 60 / 40 / 30 / 20 / 10 steps
    11.6 / 10.0 / 9.8 / 8.9 / 8.6 ms at k = 5, 19.3-20.4 / 18.6 / 18.3 / 18.3 / 18.7 at k = 16; the same labels)."""
    return int(min(OT_WARM_ITERS, max(20, n // 400)))


class MF(nn.Module):
    """utils.py:30-43.  Constructed on the CPU with exactly the reference's four normal
    fills (so the global torch stream advances identically); `.to('cuda')` moves the
    tables to HBM.  forward() scores (uid, iid) pairs with the HIP scoring kernel."""

    def __init__(self, n_user, n_item, k=16):
        super().__init__()
        self.k = k
        self.user_mat = nn.Embedding(n_user, k)
        self.item_mat = nn.Embedding(n_item, k)
        self.init_weight()

    def init_weight(self):
        nn.init.normal_(self.user_mat.weight, std=STD)
        nn.init.normal_(self.item_mat.weight, std=STD)

    @classmethod
    def from_tables(cls, U, V):
        """Wrap trained device tables without drawing from any generator."""
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.k = U.shape[1]
        m.user_mat = nn.Embedding(U.shape[0], U.shape[1], _weight=U)
        m.item_mat = nn.Embedding(V.shape[0], V.shape[1], _weight=V)
        m.requires_grad_(False)
        return m

    def forward(self, uid, iid):
        U, V, d = padded_tables(self)
        uid = uid.to(device=U.device, dtype=torch.int32).contiguous()
        iid = iid.to(device=U.device, dtype=torch.int32).contiguous()
        pred = torch.empty(uid.numel(), dtype=torch.float32, device=U.device)
        import ctypes
        Up = (ctypes.c_void_p * 1)(U.data_ptr())
        Vp = (ctypes.c_void_p * 1)(V.data_ptr())
        nv.check(nv.lib().ure_score(Up, Vp, 1, 1, 1, 1, nv.ptr(uid), nv.ptr(iid), None, uid.numel(), d,
                                    nv.ptr(pred), None, nv.stream_handle()), 'ure_score')
        return pred


def padded_tables(model):
    """(U, V, d) as contiguous device tensors whose width is the kernels' padded d."""
    U, V = model.user_mat.weight.detach(), model.item_mat.weight.detach()
    if not U.is_cuda:
        raise nv.NativeError('model tables are not on the HIP device (call model.to("cuda")): no CPU fallback')
    d = engine.pad_dim(U.shape[1])

    def fix(t):
        if t.shape[1] == d and t.is_contiguous() and t.dtype == torch.float32:
            return t
        out = torch.zeros(t.shape[0], d, dtype=torch.float32, device=t.device)
        out[:, :t.shape[1]] = t
        return out
    return fix(U), fix(V), d


def baseTest(dataloader, models, loss_fn=None, device=None, verbose=0, top_k=10):
    """utils.py:115-187: (rmse, ndcg, hr) of the mean-ensemble of `models` on a test
    loader.  loss_fn / device are accepted for signature compatibility."""
    ev = as_loader(dataloader).eval_set()
    tabs = [padded_tables(m) for m in models]
    d = tabs[0][2]
    rmse, ndcg, hr = ev.evaluate([(U, V) for U, V, _ in tabs], d, top_k=top_k)
    if verbose == 2:
        print(f'Test - RMSE: {rmse:>.4f}, NDCG: {ndcg:>.3f}, HR: {hr:>.3f}')
    return rmse, ndcg, hr


def computeNDCG(r, top_k):
    """utils.py:190-207 (host helper; the device kernel computes the same value)."""
    r = np.asarray(r, dtype=np.float64)
    if len(r) == 0:
        return 0
    r = np.concatenate([r, np.zeros(top_k - len(r))])
    return computeDCG(r) / computeDCG(np.ones(top_k))


def computeDCG(r):
    """utils.py:209-210."""
    return r[0] + np.sum(r[1:] / np.log2(np.arange(2, len(r) + 1)))


def dist_rank():
    """(rank, torch.distributed module or None): who writes artifacts when several ranks run the same call."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist
    return 0, None


def atomic_save(path, writer):
    """Write a file under a temporary name and rename it into place: a reader on another rank never sees half of it."""
    tmp = f'{path}.tmp{os.getpid()}'
    writer(tmp)
    os.replace(tmp, path)


def saveObject(filename, obj):
    def write(tmp):
        with open(tmp, 'wb') as output:
            pickle.dump(obj, output, pickle.HIGHEST_PROTOCOL)
    atomic_save(filename + '.pkl', write)


def loadObject(filename):
    with open(filename + '.pkl', 'rb') as input:
        return pickle.load(input)


def timefn(fn):
    """utils.py:616-626."""
    @wraps(fn)
    def measure_time(*args, **kwargs):
        t1 = time.time()
        result = fn(*args, **kwargs)
        print(f"@time: {time.time() - t1: .5f} s")
        return result
    return measure_time


def _ot_round(Xd, centroid, n, k, d, dist_d, label_d, cent_d, counts_d, pi=None, mfma_check=None, timing=None):
    """One round of utils.py:637-648 on the device + the exact host LP.  mfma_check (a list): the round is also
    solved on the MFMA form of the cost matrix (ure_ot_cost_mfma) and the number of labels that differ from the
    exact path's is appended -- the cross-check that form needs before anyone may rely on it.
    timing (a list): the round's parts are timed on the host's clock with a synchronisation between them (bench.py's OT leg) and
    appended as a dict of milliseconds."""
    L, st = nv.lib(), nv.stream_handle()
    tick = None
    if timing is not None:
        marks = []

        def tick(name):
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
        tick('start')
    cd = torch.from_numpy(np.ascontiguousarray(centroid, dtype=np.float32)).to(Xd.device)
    fast_label = None
    if mfma_check is not None:
        nv.check(L.ure_ot_cost_mfma(nv.ptr(Xd), nv.ptr(cd), n, k, d, nv.ptr(dist_d), st), 'ure_ot_cost_mfma')
        fpi = np.zeros(k, dtype=np.float64) if pi is None else pi.copy()
        nv.check(L.ure_ot_potentials(nv.ptr(dist_d), n, k, ot_warm_iters(n), fpi.ctypes.data, None, st), 'ure_ot_potentials')
        fast_label, _, _, _ = nv.ot_assign_warm(dist_d.cpu().numpy(), fpi, want_plan=False)
    if tick:
        tick('centroid_upload')
    nv.check(L.ure_ot_cost(nv.ptr(Xd), nv.ptr(cd), n, k, d, nv.ptr(dist_d), st), 'ure_ot_cost')
    if tick:
        tick('cost_kernel')
    # cluster potentials by dual ascent on the device (a warm start only: the LP below is solved exactly for any
    # potentials), while the cost matrix travels to the host
    pi = np.zeros(k, dtype=np.float64) if pi is None else pi         # in: the previous round's, out: this round's
    nv.check(L.ure_ot_potentials(nv.ptr(dist_d), n, k, ot_warm_iters(n), pi.ctypes.data, None, st), 'ure_ot_potentials')
    if tick:
        tick('device_potentials')
    dist = dist_d.cpu().numpy()                                       # [k, n] fp32 (synchronises)
    if tick:
        tick('cost_to_host')
    label, _, _, _ = nv.ot_assign_warm(dist, pi, want_plan=False)     # exact EMD + argmax (host)
    if tick:
        tick('host_solver')
    if fast_label is not None:
        mfma_check.append(int((fast_label != label).sum()))
    # utils.py:648 from member lists: a stable sort of the labels (ascending id inside a cluster = numpy's order of addition)
    # (labels as the narrowest unsigned type: numpy's stable sort of 8- and 16-bit keys is a radix sort -- 0.3 ms instead of 3 at n = 162,000)
    keys = label.astype(np.uint8 if k <= 256 else np.uint16 if k <= 65536 else np.int64)
    order = torch.from_numpy(np.argsort(keys, kind='stable').astype(np.int32)).to(Xd.device)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(label, minlength=k))]).astype(np.int64)).to(Xd.device)
    nv.check(L.ure_ot_centroids_members(nv.ptr(Xd), nv.ptr(order), nv.ptr(off), n, k, d, nv.ptr(cent_d), nv.ptr(counts_d), st),
             'ure_ot_centroids_members')
    new_centroid = cent_d.cpu().numpy()
    if tick:
        tick('centroids')
        timing.append({b[0] + '_ms': round((b[1] - a[1]) * 1e3, 4) for a, b in zip(marks[:-1], marks[1:])})
    return dist, label, new_centroid


@timefn
def ot_cluster(X, k, max_iters=10, timing=None):
    """utils.py:628-656.  Initial centroids come from the global numpy generator, as in
    the reference.  Returns (inertia, label[int64]).  timing (a list, optional): every round's parts in milliseconds (_ot_round)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, d = X.shape
    if k < 1 or k > n:
        raise ValueError('need 1 <= k <= n clusters')
    centroid = X[np.random.choice(n, size=k, replace=False)]
    dev = engine._device()
    Xd = torch.from_numpy(X).to(dev)
    dist_d = torch.empty(k, n, dtype=torch.float32, device=dev)
    label_d = torch.empty(n, dtype=torch.int32, device=dev)
    cent_d = torch.empty(k, d, dtype=torch.float32, device=dev)
    counts_d = torch.empty(k, dtype=torch.int32, device=dev)
    pi = np.zeros(k, dtype=np.float64)
    # URE_OT_MFMA=1: every round is solved a second time on the MFMA cost matrix and compared (ot_cluster.mfma_mismatches);
    # the labels returned are always the exact path's
    check = [] if os.environ.get('URE_OT_MFMA', '0') == '1' else None
    ot_cluster.mfma_mismatches = check
    for _ in range(max_iters):
        dist, label, new_centroid = _ot_round(Xd, centroid, n, k, d, dist_d, label_d, cent_d, counts_d, pi, check, timing)
        inertia = np.min(dist, axis=0).sum()
        if np.allclose(centroid, new_centroid):
            break
        centroid = new_centroid
    print(f'{inertia:.3f}', end=' ')
    return inertia, label.astype(np.int64)


# ---------------------------------------------------------------------------
# Comparison clusterers (utils.py:354-418): k-means / balanced k-means on the embedding the
# reference's notebook compared OT grouping against.  Distances and centroid updates run on the
# GPU in scipy's csr arithmetic (labels identical to the reference's), the assignment -- a global
# sort of n*k distances and a greedy fill -- on the host.
# ---------------------------------------------------------------------------
def _dense_f32(sp_mat):
    return np.ascontiguousarray(sp_mat.toarray() if hasattr(sp_mat, 'toarray') else sp_mat, dtype=np.float32)


def singleKmeans(k, n_user, sp_mat, balanced, max_iter):
    """utils.py:354-404.  sp_mat: csr_matrix or array [n_user, n_embedding]; initial centroids from
    numpy's global generator.  Returns (label int64 [n_user], inertia)."""
    X = _dense_f32(sp_mat)
    n, d = X.shape
    assert n == n_user
    if k < 1 or k > n:
        raise ValueError('need 1 <= k <= n clusters')
    L, st, dev = nv.lib(), nv.stream_handle(), engine._device()
    group_len = int(np.ceil(n_user / k))
    cen_idx = np.random.choice(n_user, k, replace=False)
    Xd = torch.from_numpy(X).to(dev)
    cent_d = Xd[torch.from_numpy(cen_idx).to(dev)].contiguous()
    dist_d = torch.empty(n, k, dtype=torch.float32, device=dev)
    label_d = torch.empty(n, dtype=torch.int32, device=dev)
    counts_d = torch.empty(k, dtype=torch.int32, device=dev)
    label = np.zeros(n, dtype=np.int32)
    new_label = np.empty(n, dtype=np.int32)
    inertia = ctypes.c_double(0.0)
    for _ in range(max_iter):
        nv.check(L.ure_kmeans_cost(nv.ptr(Xd), nv.ptr(cent_d), n, k, d, nv.ptr(dist_d), st), 'ure_kmeans_cost')
        dist = dist_d.cpu().numpy()
        nv.check(L.ure_host_kmeans_assign(dist.ctypes.data, n, k, group_len if balanced else 0, new_label.ctypes.data,
                                          ctypes.byref(inertia)), 'ure_host_kmeans_assign')
        if (new_label == label).all():
            break
        label = new_label.copy()
        label_d.copy_(torch.from_numpy(label))
        nv.check(L.ure_kmeans_centroids(nv.ptr(Xd), nv.ptr(label_d), n, k, d, nv.ptr(cent_d), nv.ptr(counts_d), st),
                 'ure_kmeans_centroids')
        if int(counts_d.min().item()) == 0:
            raise ZeroDivisionError('a cluster lost all its members (utils.py:403 divides by its size)')
    return label.astype(np.int64), float(inertia.value)


def kmeans(n_group, n_user, sp_mat, balanced=False, n_init=5, max_iter=10):
    """utils.py:406-418: the labels of the best of n_init runs (smallest inertia)."""
    tmp_inertia, fin_label = 1e10, None
    for _ in range(n_init):
        label, inertia = singleKmeans(n_group, n_user, sp_mat, balanced, max_iter)
        if inertia < tmp_inertia:
            tmp_inertia = inertia
            fin_label = label
    return fin_label
