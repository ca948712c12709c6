"""`ratings.dat` -> `squ0_train.csv` / `squ0_test.csv`, the input format of the SISA path.

Restates the reference's preprocessing notebook (data/ml1m/pro.ipynb) as a function:
  cell 0-1   read `uid::iid::rating::timestamp`
  cell 4     iterative 5-core filter: drop items with < low ratings, then users with < low
             ratings, until nothing is dropped
  cell 7     `uid -= 1`, then squeeze user and item ids to 0..n-1 in order of first appearance
             (the sorted copy it writes, ratings.csv, is not read again)
  cell 10    per user, in id order: n_train = int(total * 0.9) rows drawn with
             `random.sample` (Python's generator; the notebook never seeds it, so a run is
             reproducible only if the caller passes a seed), the rest is the test split; both
             splits keep the FILE order of the rows (ml-1m's ratings.dat is grouped by user);
             ratings written as float16 values

    python -m ultrare_amd.preprocess ratings.dat out_dir [--seed 0]
"""
import argparse
import os
import random

import numpy as np


def five_core(uid, iid, low=5):
    """Boolean mask of the rows that survive the notebook's alternating item / user filter."""
    keep = np.ones(len(uid), dtype=bool)
    while True:
        dropped = 0
        ci = np.bincount(iid[keep], minlength=int(iid.max()) + 1)
        bad = keep & (ci[iid] < low)
        dropped += len(np.unique(iid[bad]))
        keep &= ~bad
        cu = np.bincount(uid[keep], minlength=int(uid.max()) + 1)
        bad = keep & (cu[uid] < low)
        dropped += len(np.unique(uid[bad]))
        keep &= ~bad
        if dropped == 0:
            return keep


def squeeze(ids):
    """ids -> 0..n-1 in order of first appearance (pandas `unique()` order), plus the dict."""
    _, first, inv = np.unique(ids, return_index=True, return_inverse=True)
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind='stable')] = np.arange(len(first))
    new = rank[inv]
    return new, {int(o): int(n) for o, n in zip(ids[np.sort(first)], range(len(first)))}


def preprocess(ratings_dat, out_dir, low=5, train_ratio=0.9, seed=None):
    rows = []
    with open(ratings_dat) as f:
        for line in f:
            p = line.strip().split('::')
            if len(p) >= 3:
                rows.append((int(p[0]), int(p[1]), float(p[2])))
    a = np.asarray(rows, dtype=np.float64)
    uid, iid, rating = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
    keep = five_core(uid, iid, low)
    uid, iid, rating = uid[keep] - 1, iid[keep], rating[keep]
    uid, user_dict = squeeze(uid)
    iid, item_dict = squeeze(iid)
    # cell 10 works on the frame in FILE order (cell 7 sorts only the copy it writes to ratings.csv): a user's
    # candidate list is its row numbers ascending, and both splits keep the file order of the rows
    rng = random.Random(seed) if seed is not None else random
    by_user = np.argsort(uid, kind='stable')
    start = np.searchsorted(uid[by_user], np.arange(int(uid.max()) + 2))
    is_train = np.zeros(len(uid), dtype=bool)
    for u in range(int(uid.max()) + 1):
        idx = by_user[int(start[u]):int(start[u + 1])].tolist()
        is_train[rng.sample(idx, int(len(idx) * train_ratio))] = True
    os.makedirs(out_dir, exist_ok=True)
    val = rating.astype(np.float16)
    for name, m in (('squ0_train.csv', is_train), ('squ0_test.csv', ~is_train)):
        with open(os.path.join(out_dir, name), 'w') as f:
            for u, i, v in zip(uid[m].tolist(), iid[m].tolist(), val[m].tolist()):
                f.write(f'{u},{i},{v}\n')
    np.save(os.path.join(out_dir, 'user_dict'), user_dict)
    np.save(os.path.join(out_dir, 'item_dict'), item_dict)
    return {'n_user': int(uid.max()) + 1, 'n_item': int(iid.max()) + 1, 'n_train': int(is_train.sum()),
            'n_test': int((~is_train).sum())}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('ratings_dat')
    ap.add_argument('out_dir')
    ap.add_argument('--low', type=int, default=5)
    ap.add_argument('--seed', type=int, default=None)
    a = ap.parse_args()
    print(preprocess(a.ratings_dat, a.out_dir, a.low, seed=a.seed))
