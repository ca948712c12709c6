"""Synthetic rating sets shaped like the reference's inputs.

`data/ml1m/ratings.dat` is not shipped with the reference (.MISSING_LARGE_BLOBS) and
there is no network, so benchmarks and scale tests run on generated data with the
same shape statistics as the preprocessed ml-1m files (data/ml1m/pro.ipynb):
  * per-user rating counts ~ max(20, lognormal(4.6, 1.0)), rescaled to the exact total;
  * items ~ Zipf(s=1) over a random permutation of ids, no duplicate (user, item);
  * ratings in {1..5} with p = (.056, .107, .261, .349, .226) (stored raw, /5 at read);
  * rows sorted by user id (as the CSV is); per-user ~10 % hold-out as the test split.
"""
import numpy as np

ML1M = dict(n_user=6040, n_item=3416, n_train=896914, n_test=102697)
ML25M = dict(n_user=162000, n_item=60000, n_train=22500000, n_test=2500000)
SEED = 20231210
_RATING_P = np.array([.056, .107, .261, .349, .226])
_RATING_P = _RATING_P / _RATING_P.sum()


def _exact_counts(rs, n_user, total, lo, hi):
    c = np.maximum(lo, np.rint(rs.lognormal(4.6, 1.0, n_user))).astype(np.int64)
    c = np.minimum(c, hi)
    for _ in range(200):
        diff = total - int(c.sum())
        if diff == 0:
            break
        scale = total / c.sum()
        c = np.clip(np.rint(c * scale), lo, hi).astype(np.int64)
        diff = total - int(c.sum())
        if abs(diff) <= n_user:
            room = (c < hi) if diff > 0 else (c > lo)
            idx = rs.choice(np.flatnonzero(room), size=min(abs(diff), int(room.sum())), replace=False)
            c[idx] += 1 if diff > 0 else -1
    assert c.sum() == total, 'could not hit the requested number of ratings'
    return c


def make_dataset(n_user, n_item, n_train, n_test, seed=SEED):
    """-> dict(train=(uid, iid, rating), test=(uid, iid, rating)), int64/int64/float64,
    ratings raw in 1..5, rows sorted by (uid, iid)."""
    rs = np.random.RandomState(seed)
    total = n_train + n_test
    counts = _exact_counts(rs, n_user, total, 20, n_item)
    w = 1.0 / np.arange(1, n_item + 1)
    cdf = np.cumsum(w / w.sum())
    item_of_rank = rs.permutation(n_item)
    logw = np.log(w)
    # users who rate a large share of the catalogue: exact weighted sampling without
    # replacement by the Gumbel top-k trick (rejection sampling would crawl for them)
    dense = np.flatnonzero(counts > n_item // 8)
    parts = []
    for u in dense:
        g = logw - np.log(-np.log(rs.random_sample(n_item)))
        top = np.argpartition(-g, counts[u] - 1)[:counts[u]]
        parts.append(u * n_item + item_of_rank[top])
    have_keys = np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64)
    need = counts.copy()
    need[dense] = 0
    while need.sum() > 0:
        draw = ((need * 1.5).astype(np.int64) + 8) * (need > 0)
        users = np.repeat(np.arange(n_user), draw)
        ranks = np.minimum(np.searchsorted(cdf, rs.random_sample(len(users))), n_item - 1)
        keys = np.unique(users * n_item + item_of_rank[ranks])
        keys = keys[~np.isin(keys, have_keys, assume_unique=True)]
        ku = keys // n_item
        # keep at most need[u] new keys per user, chosen at random
        pri = rs.random_sample(len(keys))
        order = np.lexsort((pri, ku))
        keys, ku = keys[order], ku[order]
        start = np.searchsorted(ku, np.arange(n_user))
        within = np.arange(len(keys)) - start[ku]
        keep = within < need[ku]
        have_keys = np.union1d(have_keys, keys[keep])
        need = counts - np.bincount(have_keys // n_item, minlength=n_user)
    uid = have_keys // n_item
    iid = have_keys % n_item
    rating = rs.choice(np.arange(1, 6), size=len(uid), p=_RATING_P).astype(np.float64)
    # per-user hold-out
    n_te = np.floor(0.1 * counts + 0.5).astype(np.int64)
    n_te = np.clip(n_te, 1, counts - 1)
    diff = n_test - int(n_te.sum())
    while diff != 0:
        room = (n_te < counts - 1) if diff > 0 else (n_te > 1)
        idx = rs.choice(np.flatnonzero(room), size=min(abs(diff), int(room.sum())), replace=False)
        n_te[idx] += 1 if diff > 0 else -1
        diff = n_test - int(n_te.sum())
    pri = rs.random_sample(len(uid))
    order = np.lexsort((pri, uid))
    start = np.searchsorted(uid, np.arange(n_user))
    within = np.empty(len(uid), dtype=np.int64)
    within[order] = np.arange(len(uid)) - start[uid[order]]
    is_test = within < n_te[uid]
    tr, te = ~is_test, is_test
    return {'train': (uid[tr], iid[tr], rating[tr]), 'test': (uid[te], iid[te], rating[te]),
            'n_user': n_user, 'n_item': n_item}


def write_csv(path, triple):
    u, i, r = triple
    with open(path, 'w') as f:
        for a, b, c in zip(u.tolist(), i.tolist(), r.tolist()):
            f.write(f'{a},{b},{c}\n')


def uniform_shards(n_user, n_group, seed=0):
    """The reference's uniform grouping (read.py:22-33) as an array user -> shard."""
    org = np.arange(n_user).tolist()
    if n_group == 1:
        return np.zeros(n_user, dtype=np.int64), [org]
    group_len = int(np.ceil(n_user / n_group))
    st = np.random.get_state()
    np.random.seed(seed)
    np.random.shuffle(org)
    np.random.set_state(st)
    groups = [org[i * group_len:(i + 1) * group_len] for i in range(n_group)]
    shard_of = np.empty(n_user, dtype=np.int64)
    for s, g in enumerate(groups):
        shard_of[np.asarray(g, dtype=np.int64)] = s
    return shard_of, groups


def split_shards(triple, shard_of, n_group, max_rating=5):
    """Partition (uid, iid, raw rating) by the shard of the user; ratings / max_rating, fp32."""
    u, i, r = triple
    s = shard_of[u]
    out = []
    for g in range(n_group):
        m = s == g
        out.append((u[m].astype(np.int32), i[m].astype(np.int32), (r[m] / max_rating).astype(np.float32)))
    return out


def ot_embedding(n, d, seed):
    """A user embedding of 12 blurred clusters for the OT grouping legs (the generator tests/golden/make_golden.py used for ot_ml1m.npz:
    the fixture stores its checksum).  float32 [n, d]."""
    rs = np.random.RandomState(seed)
    centers = rs.standard_normal((12, d)) * 0.8
    which = rs.randint(0, 12, n)
    X = centers[which] + rs.standard_normal((n, d)) * 0.6
    return X.astype(np.float32)
