"""oracle/cpu_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + the C file mf_oracle.c + torch's *public* CPU RNG) of the
reference's SISA hot path, written from SURVEY.md and the reference sources read as
text.  Citations are file:line under /root/reference.  Checked against the golden
vectors in tests/golden/ by tests/test_oracle_golden.py.

What is restated
  read.py:22-33      uniform grouping                 -> uniform_groups
  read.py:52-70      shard partition + user deletion  -> partition
  read.py:73-106     shard ordering by rating count   -> order_by_count
  utils.py:31-40     MF init (4 normal fills)          -> mf_init           
  read.py:108-133    shuffled batches                  -> draw_seed / epoch_perm
  utils.py:46-111    baseTrain                         -> train_epoch
  utils.py:115-210   baseTest, computeNDCG/DCG         -> eval_metrics
  scratch.py:51-148  Scratch.train                     -> scratch_train
  sisa.py:25-118     Sisa.learn / unlearn / test       -> sisa_learn / sisa_unlearn
  utils.py:628-656   ot_cluster                        -> ot_cluster (LP solver: see below)
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import build as _build

_LIB = None
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(_build.build())
        L.ure_oracle_train_epoch.restype = ctypes.c_double
        L.ure_oracle_train_epoch.argtypes = [_f32p, _f32p, _f32p, _f32p, _i32p, _i32p, _f32p, _i32p,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i64p]
        L.ure_oracle_score.restype = None
        L.ure_oracle_score.argtypes = [ctypes.POINTER(_f32p), ctypes.POINTER(_f32p), ctypes.c_int32, _i32p, _i32p,
                                       ctypes.c_int64, ctypes.c_int32, _f32p]
        L.ure_oracle_ot_cost.restype = None
        L.ure_oracle_ot_cost.argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _f32p]
        L.ure_oracle_centroids.restype = None
        L.ure_oracle_centroids.argtypes = [_f32p, _i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _f32p]
        _LIB = L
    return _LIB


def _p(a, t):
    # the C side reads raw memory: an array of another element type (int64 ids where int32 are expected ...) would be
    # misread silently, so the element type is checked here
    want = {ctypes.c_float: np.float32, ctypes.c_double: np.float64, ctypes.c_int32: np.int32, ctypes.c_int64: np.int64}.get(t._type_)
    if want is not None and (a.dtype != want or not a.flags['C_CONTIGUOUS']):
        raise TypeError(f'oracle: expected a contiguous {np.dtype(want)} array, got {a.dtype}')
    return a.ctypes.data_as(t)


@dataclass
class Hyper:
    """config.py:17-31 (InsParam)."""
    k: int = 16
    lam: float = 0.1
    seed: int = 42
    batch: int = 30000
    lr: float = 0.001
    lr_decay: float = 0.95
    momentum: float = 0.9
    epochs: int = 50
    max_rating: float = 5


# ----------------------------------------------------------------------------
# data plumbing
# ----------------------------------------------------------------------------
def load_csv(path):
    """`uid,iid,rating` rows without header (read.py:37)."""
    a = np.loadtxt(path, delimiter=',', dtype=np.float64, ndmin=2)
    return a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].copy()


def uniform_groups(n_user, n_group):
    """read.py:22-33: seed(0) shuffle of the id *list*, ceil-sized consecutive slices."""
    org = np.arange(n_user).tolist()
    if n_group == 1:
        return [org]
    group_len = int(np.ceil(n_user / n_group))
    np.random.seed(0)
    np.random.shuffle(org)
    return [org[i * group_len:(i + 1) * group_len] for i in range(n_group)]


def order_by_count(uid, group_index):
    """read.py:40-50,77-81,102: shards reordered ascending by their row count."""
    counts = [int(np.isin(uid, g).sum()) for g in group_index]
    order = np.argsort(counts)
    return [group_index[i] for i in order]


def partition(uid, iid, raw, group_index, del_user=(), max_rating=5):
    """read.py:52-70: shard s = rows (file order) whose uid is in group_index[s] and
    not deleted; rating / max_rating in float64, later cast to float32
    (read.py:113,124)."""
    dels = set(int(x) for x in del_user)
    out = []
    for g in group_index:
        keep = np.array(sorted(set(int(x) for x in g) - dels), dtype=np.int64)
        loc = np.isin(uid, keep)
        out.append((uid[loc].astype(np.int32), iid[loc].astype(np.int32),
                    (raw[loc] / max_rating).astype(np.float32)))
    return out


def hstack(shards):
    """config.py:144-148 test_total = hstack of the per-shard test arrays."""
    return tuple(np.concatenate([s[c] for s in shards]) for c in range(3))


# ----------------------------------------------------------------------------
# RNG stream -- torch's process-global CPU generator
# ----------------------------------------------------------------------------
def mf_init(n_user, n_item, k):
    """utils.py:31-40: two nn.Embedding constructors (N(0,1) fills, overwritten) then
    init_weight's two normal_(std=1) fills, all on the global CPU generator."""
    torch.empty(n_user, k).normal_(0, 1)
    torch.empty(n_item, k).normal_(0, 1)
    U = torch.empty(n_user, k).normal_(0, 1)
    V = torch.empty(n_item, k).normal_(0, 1)
    return U.numpy().copy(), V.numpy().copy()


def draw_seed():
    """One `torch.empty((), dtype=int64).random_()` draw (DataLoader base seed /
    RandomSampler seed)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def epoch_perm(seed, n):
    """RandomSampler: fresh Generator seeded with the drawn seed, randperm(n)."""
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).numpy().astype(np.int32)


# ----------------------------------------------------------------------------
# arithmetic
# ----------------------------------------------------------------------------
class MFState:
    """One model being trained: tables, momentum, optimizer step count."""

    def __init__(self, U, V):
        self.U, self.V = np.ascontiguousarray(U), np.ascontiguousarray(V)
        self.mU, self.mV = np.zeros_like(self.U), np.zeros_like(self.V)
        self.steps = np.zeros(1, dtype=np.int64)


def train_epoch(st, data, perm, B, lr, lam, mu):
    """utils.py:58-111: returns (train_loss, rmse) = sqrt(sum batch losses / N) twice."""
    uid, iid, r = data
    n = len(uid)
    tot = lib().ure_oracle_train_epoch(_p(st.U, _f32p), _p(st.V, _f32p), _p(st.mU, _f32p), _p(st.mV, _f32p),
                                       _p(uid, _i32p), _p(iid, _i32p), _p(r, _f32p),
                                       _p(perm, _i32p) if perm is not None else None,
                                       n, B, st.U.shape[0], st.V.shape[0], st.U.shape[1],
                                       lr, lam, mu, _p(st.steps, _i64p))
    v = float(np.sqrt(tot / n))
    return v, v


def score(models, uid, iid):
    """utils.py:140-145: stack(preds).mean(0), fp32."""
    S = len(models)
    keep = [np.ascontiguousarray(m[0]) for m in models] + [np.ascontiguousarray(m[1]) for m in models]
    Us = (_f32p * S)(*[_p(a, _f32p) for a in keep[:S]])
    Vs = (_f32p * S)(*[_p(a, _f32p) for a in keep[S:]])
    pred = np.empty(len(uid), dtype=np.float32)
    lib().ure_oracle_score(Us, Vs, S, _p(uid, _i32p), _p(iid, _i32p), len(uid), keep[0].shape[1], _p(pred, _f32p))
    return pred


_LOG2 = np.log2(np.arange(2, 11))


def dcg(r):
    """utils.py:209-210."""
    return r[0] + np.sum(r[1:] / np.log2(np.arange(2, len(r) + 1)))


def ndcg_at_k(r, top_k=10):
    """utils.py:190-207."""
    n = len(r)
    if n == 0:
        return 0
    r = np.concatenate([r, np.zeros(top_k - n)])
    return dcg(r) / dcg(np.ones(top_k))


def eval_from_pred(uid, r, pred, batch, top_k=10):
    """utils.py:127-187 given the ensemble prediction: RMSE over batch-wise fp32
    losses, per-user HR@k and the reference's positional NDCG@k, users in
    first-appearance order, argsort forced stable (SURVEY 7 'NDCG tie-breaking')."""
    n = len(uid)
    loss = 0.0
    for b0 in range(0, n, batch):
        e = pred[b0:b0 + batch] - r[b0:b0 + batch]
        loss += float(np.float32((e * e).astype(np.float64).sum()))
    rmse = float(np.sqrt(loss / n))
    _, first = np.unique(uid, return_index=True)
    users = uid[np.sort(first)]
    order = np.argsort(uid, kind='stable')
    su = uid[order]
    starts = np.searchsorted(su, users, side='left')
    ends = np.searchsorted(su, users, side='right')
    ndcg, hr = [], []
    for s, e in zip(starts, ends):
        idx = order[s:e]
        ur = r[idx].astype(np.float64)
        up = pred[idx].astype(np.float64)
        top_r = np.argsort(ur, kind='stable')[::-1][:top_k]
        top_p = np.argsort(up, kind='stable')[::-1][:top_k]
        rel = ur[top_p]
        hr.append(sum(rel >= (4 / 5)) / top_k)
        common = np.isin(top_r, top_p)
        rel = rel * (rel >= (4 / 5))
        ndcg.append(ndcg_at_k(rel * common, top_k))
    return rmse, float(np.mean(ndcg)), float(np.mean(hr))


def eval_metrics(data, models, batch, top_k=10):
    uid, iid, r = data
    return eval_from_pred(uid, r, score(models, uid, iid), batch, top_k)


# ----------------------------------------------------------------------------
# drivers
# ----------------------------------------------------------------------------
def scratch_train(h, n_user, n_item, train, test, test_total=None, prev_models=(), on_epoch=None, with_eval=True):
    """scratch.py:51-148.  Consumes the global torch CPU stream exactly as SURVEY 3.4
    lists.  Returns (U, V, log)."""
    U0, V0 = mf_init(n_user, n_item, h.k)
    st = MFState(U0, V0)
    log = {k: [] for k in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr')}
    for t in range(h.epochs):
        lr = h.lr * (h.lr_decay ** (t // 50))                       # StepLR(50, gamma), scratch.py:69,80
        draw_seed()                                                  # 5a DataLoader base seed
        perm = epoch_perm(draw_seed(), len(train[0]))                # 5b sampler seed -> randperm
        tl, _ = train_epoch(st, train, perm, h.batch, lr, h.lam, h.momentum)
        models = list(prev_models) + [(st.U, st.V)]
        draw_seed()                                                  # 5c group-test loader base seed
        nan = (float('nan'),) * 3                                    # with_eval=False: draws only (tests at scale)
        g = eval_metrics(test, models, h.batch) if with_eval else nan
        if test_total is None:
            tot = g
        else:
            draw_seed()                                              # 5d total-test loader base seed
            tot = eval_metrics(test_total, models, h.batch) if with_eval else nan
        log['train_loss'].append(tl)
        for name, val in zip(('test_rmse', 'test_ndcg', 'test_hr'), g):
            log[name].append(val)
        if test_total is not None:
            for name, val in zip(('total_rmse', 'total_ndcg', 'total_hr'), tot):
                log[name].append(val)
        if on_epoch:
            on_epoch(t, st)
    return st.U, st.V, log


def sisa_learn(h, n_user, n_item, group_index, train_list, test_list, test_total, with_eval=True):
    """sisa.py:25-63."""
    models, logs = [], []
    for i in range(len(group_index)):
        U, V, log = scratch_train(h, n_user, n_item, train_list[i], test_list[i], test_total, prev_models=models,
                                  with_eval=with_eval)
        models.append((U, V))
        logs.append(log)
    pre = [m[0].copy() for m in models]
    merged = np.zeros_like(models[0][0])
    for i, g in enumerate(group_index):
        merged[np.asarray(g, dtype=np.int64)] = models[i][0][np.asarray(g, dtype=np.int64)]
    models = [(merged, m[1]) for m in models]
    draw_seed()                                                      # SURVEY 3.4 item 6 (Sisa.test loader)
    log0 = eval_metrics(test_total, models, h.batch)
    return {'models': models, 'U_pre': pre, 'merged': merged, 'log0': log0, 'logs': logs}


def sisa_unlearn(h, n_user, n_item, group_index, models, train_list, test_list, test_total, del_user, with_eval=True):
    """sisa.py:66-118.  `models` = list of (merged U, V_i) from sisa_learn."""
    retrain = set()
    for u in del_user:
        for i, g in enumerate(group_index):
            if u in g:
                retrain.add(i)
                break
    models = list(models)
    before = models[0][0]
    logs = {}
    for i in retrain:                                                # python set iteration order
        U, V, log = scratch_train(h, n_user, n_item, train_list[i], test_list[i], test_total, prev_models=models,
                                  with_eval=with_eval)
        models[i] = (U, V)
        logs[i] = log
    merged = before.copy()
    for i in retrain:
        g = np.asarray(group_index[i], dtype=np.int64)
        merged[g] = models[i][0][g]
    models = [(merged, m[1]) for m in models]
    draw_seed()
    log0 = eval_metrics(test_total, models, h.batch)
    return {'models': models, 'merged': merged, 'log0': log0, 'logs': logs, 'retrained': sorted(retrain)}


# ----------------------------------------------------------------------------
# OT balanced clustering
# ----------------------------------------------------------------------------
def ot_cost(X, C):
    """utils.py:637, bit-exact numpy fp32 order -> dist [k,n]."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    C = np.ascontiguousarray(C, dtype=np.float32)
    dist = np.empty((C.shape[0], X.shape[0]), dtype=np.float32)
    lib().ure_oracle_ot_cost(_p(X, _f32p), _p(C, _f32p), X.shape[0], C.shape[0], X.shape[1], _p(dist, _f32p))
    return dist


def centroids(X, label, k):
    """utils.py:648, bit-exact."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    label = np.ascontiguousarray(label, dtype=np.int64)
    C = np.empty((k, X.shape[1]), dtype=np.float32)
    lib().ure_oracle_centroids(_p(X, _f32p), _p(label, _i64p), X.shape[0], k, X.shape[1], _p(C, _f32p))
    return C


def emd_exact(M):
    """Exact OT plan for uniform marginals, cost M [n,k] (float64).

    The reference calls POT 0.9.0 `ot.emd(a, b, dist.T, 1e-3)` (utils.py:640-644;
    the 4th positional is numItermax and truncates to 0 = no cap).  POT is
    a PyPI dependency absent from /root/reference and from this image, so the
    published algorithm's *result* is restated instead: the optimum of
        min <G, M>  s.t.  G 1 = 1/n,  G^T 1 = 1/k,  G >= 0
    computed by HiGHS dual simplex.  For costs in general position the optimum is
    unique, hence solver-independent (SURVEY 3.3 'Key structural fact')."""
    from scipy.optimize import linprog
    from scipy.sparse import coo_matrix
    M = np.ascontiguousarray(M, dtype=np.float64)
    n, k = M.shape
    nv = n * k
    rows = np.concatenate([np.repeat(np.arange(n), k), n + np.tile(np.arange(k), n)])
    cols = np.concatenate([np.arange(nv), np.arange(nv)])
    A = coo_matrix((np.ones(2 * nv), (rows, cols)), shape=(n + k, nv)).tocsr()
    rhs = np.concatenate([np.ones(n) / n, np.ones(k) / k])
    res = linprog(M.reshape(-1), A_eq=A[:-1], b_eq=rhs[:-1], bounds=(0, None), method='highs-ds',
                  options={'primal_feasibility_tolerance': 1e-10, 'dual_feasibility_tolerance': 1e-10})
    assert res.status == 0, res.message
    return res.x.reshape(n, k)


def ot_fixed_point(dist_kn):
    """fp32 costs [k, n] -> int64 on a common power-of-two scale (exact: every float32 is m * 2^e)."""
    d = np.ascontiguousarray(dist_kn, dtype=np.float32).astype(np.float64)
    nz = d[d > 0]
    if nz.size == 0:
        return np.zeros(d.shape, dtype=np.int64)
    _, e = np.frexp(nz)
    shift = 24 - int(e.min())
    assert int(e.max()) + shift < 56, 'dynamic range of the costs too wide for an exact int64 certificate'
    return np.ldexp(d, shift).astype(np.int64)


def ot_certificate(dist_kn, plan_nk):
    """Proof that an integer plan (units of 1/(n k): every point ships k units, every cluster takes n)
    is an exact optimum of the transportation LP of utils.py:640-644: it is feasible and the residual
    graph has no negative cycle.  All arithmetic is integer (fixed-point costs), so the verdict has no
    tolerance.  A cycle through points reduces to a cycle over clusters whose edge a -> b costs
    min over points i holding units in a of cost[i, b] - cost[i, a].
    -> dict(feasible, optimal, objective (units), tight_cycles: a zero-cost cycle exists = the optimum
    is not unique)."""
    c = ot_fixed_point(dist_kn).T.copy()                  # [n, k]
    x = np.ascontiguousarray(plan_nk, dtype=np.int64)
    n, k = x.shape
    feasible = bool((x >= 0).all() and (x.sum(axis=1) == k).all() and (x.sum(axis=0) == n).all())
    INF = np.iinfo(np.int64).max // 4
    w = np.full((k, k), INF, dtype=np.int64)
    for a in range(k):
        rows = np.flatnonzero(x[:, a] > 0)
        if len(rows):
            w[a] = (c[rows] - c[rows, a:a + 1]).min(axis=0)
        w[a, a] = INF
    # Floyd-Warshall on Python ints (k <= a few dozen)
    D = [[int(v) if v < INF else None for v in row] for row in w]
    for m in range(k):
        for a in range(k):
            if D[a][m] is None:
                continue
            for b in range(k):
                if D[m][b] is None:
                    continue
                t = D[a][m] + D[m][b]
                if D[a][b] is None or t < D[a][b]:
                    D[a][b] = t
    diag = [D[a][a] for a in range(k) if D[a][a] is not None]
    optimal = feasible and all(v >= 0 for v in diag)
    return {'feasible': feasible, 'optimal': bool(optimal), 'objective': int((x * c).sum()),
            'tight_cycles': any(v == 0 for v in diag)}


def ot_cluster(X, k, max_iters=10, trace=None):
    """utils.py:628-656 with the global numpy RNG for the initial centroids."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n = X.shape[0]
    centroid = X[np.random.choice(n, size=k, replace=False)]
    for _ in range(max_iters):
        dist = ot_cost(X, centroid)
        inertia = np.min(dist, axis=0).sum()
        trans = emd_exact(dist.T)
        label = np.argmax(trans, axis=1)
        new_c = centroids(X, label, k)
        if trace is not None:
            trace.append({'dist': dist, 'label': label.copy(), 'plan': trans})
        if np.allclose(centroid, new_c):
            break
        centroid = new_c
    return inertia, label


# ---------------------------------------------------------------------------
# Comparison clusterers: k-means / balanced k-means (utils.py:354-418)
# ---------------------------------------------------------------------------
def kmeans_dist(X, C):
    """utils.py:373-375 on a csr input: dist = -2 X C^T + |x|^2 + |c|^2 in float32, every inner sum in
    scipy's csr order (columns ascending, one multiply and one add per term), then
    ((-2 dot) + |x|^2) + |c|^2."""
    n, d = X.shape
    dot = np.zeros((n, C.shape[0]), np.float32)
    esq = np.zeros(n, np.float32)
    csq = np.zeros(C.shape[0], np.float32)
    for j in range(d):
        dot += X[:, j:j + 1] * C[None, :, j]
        esq += X[:, j] * X[:, j]
        csq += C[:, j] * C[:, j]
    dist = np.float32(-2) * dot
    dist += esq[:, None]
    dist += csq[None, :]
    return dist


def kmeans_assign(dist, balanced):
    """utils.py:377-394: argmin, or the greedy fill in ascending order of distance with at most
    ceil(n/k) users per group (np.argsort's default kind; exact ties are broken by flat index here)."""
    n, k = dist.shape
    if not balanced:
        return dist.argmin(axis=1)
    label = np.zeros(n, dtype=np.int64)
    left = [int(np.ceil(n / k))] * k
    done = np.zeros(n, dtype=bool)
    n_done = 0
    for f in np.argsort(dist, axis=None, kind='stable'):
        if n_done == n:
            break
        u, c = divmod(int(f), k)
        if not done[u] and left[c] > 0:
            label[u], done[u] = c, True
            n_done += 1
            left[c] -= 1
    return label


def kmeans_centroids(X, label, k):
    """utils.py:402-403: centroid[j] = csr_matrix(sp_mat[label == j].mean(axis=0)) -- scipy's sparse mean:
    every member row times float32(1/count), summed in ascending row order in float32."""
    C = np.zeros((k, X.shape[1]), np.float32)
    for c in range(k):
        idx = np.nonzero(label == c)[0]
        inv = np.float32(1.0 / len(idx))
        acc = np.zeros(X.shape[1], np.float32)
        for i in idx:
            acc += X[i] * inv
        C[c] = acc
    return C


def single_kmeans(k, X, balanced, max_iter, cen_idx=None):
    """utils.py:354-404 (singleKmeans); initial centroids from numpy's global RNG unless given."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n = X.shape[0]
    label = np.zeros(n, dtype=np.int64)
    if cen_idx is None:
        cen_idx = np.random.choice(n, k, replace=False)
    C = X[cen_idx].copy()
    inertia = 0.0
    for _ in range(max_iter):
        dist = kmeans_dist(X, C)
        new = kmeans_assign(dist, balanced)
        inertia = float(np.sum(dist[np.arange(n), new]))
        if (new == label).all():
            break
        label = new
        C = kmeans_centroids(X, label, k)
    return label, inertia


def kmeans(k, X, balanced=False, n_init=5, max_iter=10):
    """utils.py:406-418: best of n_init runs by inertia."""
    best, fin = 1e10, None
    for _ in range(n_init):
        label, inertia = single_kmeans(k, X, balanced, max_iter)
        if inertia < best:
            best, fin = inertia, label
    return fin
