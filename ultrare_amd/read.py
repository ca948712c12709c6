"""Data plumbing with the reference's names and meanings (read.py of the reference),
feeding the HBM-resident engine instead of a per-sample torch DataLoader.

  readRating     read.py:9-70     CSV -> per-shard [3, N_s] float64 arrays (+ group index)
  sort_group     read.py:73-106   shard order = ascending rating count
  RatingData     read.py:108-124  holder of one shard's (users, items, ratings)
  loadData       read.py:127-133  -> ShardLoader (batch, shuffle); n_worker is accepted
                                  and ignored: there are no worker processes, batches
                                  are formed on the device from the epoch permutation
  readSparseMat  read.py:136-145  CSR rating matrix (only used by 'rating-ot' grouping)

Reference defects handled here: D9 debug prints dropped; D11 the
'a' ordering no longer needs 'ml1m' inside the path.
"""
import numpy as np
import torch


def _read_csv(path):
    """`uid,iid,rating` rows, no header (read.py:37) -> (uid int64, iid int64, rating float64),
    parsed by the library's threaded reader (ure_host_read_csv)."""
    from . import _native as nv
    u, i, r = nv.read_csv(path)
    return u.astype(np.int64), i.astype(np.int64), r


def sort_group(order='a', group_index=(), var='count', ratings0=(), dataset=''):
    """read.py:73-106 with var='count' (the only key reachable from the CLI)."""
    assert var in ['count'], "only the rating-count key is supported (density needs files the reference does not ship)"
    ratings0 = np.asarray(ratings0)
    sort_value = [int(np.isin(ratings0, np.asarray(index)).sum()) for index in group_index]
    sorted_index = np.argsort(sort_value)
    return sorted_index if order == 'a' else sorted_index[::-1]


def readRating(dir, n_user, max_rating=5, del_user=[], del_rating=[], n_group=1, group_index=[], sort='r'):
    """Same contract as read.py:9-70.

    Returns
    -------
        rating_lists:   list [n_group] of array [3, n_rating]  (uid, iid, rating / max_rating)
        group_index:    list [n_group] of user-id lists (reordered when sort in 'a','d')
    """
    if len(group_index) == 0:
        group_len = int(np.ceil(n_user / n_group))
        org_index = np.arange(n_user).tolist()
        if n_group == 1:
            group_index = [org_index]
        else:
            np.random.seed(0)
            np.random.shuffle(org_index)
            group_index = [org_index[i * group_len:(i + 1) * group_len] for i in range(n_group)]
    group_index = list(group_index)

    from . import _native as nv
    uid, iid, raw = nv.read_csv(dir)               # int32 ids, float64 ratings (the library's threaded reader)

    if sort in ['d', 'a']:
        sorted_index = sort_group(order='a', group_index=group_index, var='count', ratings0=uid)   # read.py:45: always ascending
        group_index = [group_index[i] for i in sorted_index]

    n_ids = max(n_user, int(uid.max()) + 1 if len(uid) else 0)
    shard_of = np.full(n_ids, -1, dtype=np.int32)
    overlapping = False
    for s, g in enumerate(group_index[:n_group]):
        g = np.asarray(g, dtype=np.int64)
        overlapping = overlapping or bool((shard_of[g] >= 0).any()) or len(np.unique(g)) != len(g)
        shard_of[g] = s
    if not overlapping and len(del_rating) == 0:
        # one native pass over the rows (ure_host_partition): shard = group of the row's user, deleted users dropped
        if len(del_user):
            shard_of[np.asarray(list(del_user), dtype=np.int64)] = -1
        return nv.partition64(uid, iid, raw, shard_of, n_group, max_rating), group_index

    # general form of read.py:52-70 (a user listed in two groups, or single ratings deleted): boolean passes
    deleted = np.zeros(n_ids, dtype=bool)
    if len(del_user):
        deleted[np.asarray(list(del_user), dtype=np.int64)] = True
    # del_rating (single-rating deletion, config.py:36) is always [] on the published path
    drop = np.zeros(len(uid), dtype=bool)
    for pair in np.asarray(del_rating).reshape(-1, 2) if len(del_rating) else ():
        drop |= (uid == int(pair[0])) & (iid == int(pair[1]))

    rating_lists = []
    for i in range(n_group):
        member = np.zeros(len(deleted), dtype=bool)
        member[np.asarray(group_index[i], dtype=np.int64)] = True
        loc = member[uid] & ~deleted[uid] & ~drop
        rating_lists.append(np.vstack([uid[loc].astype(np.float64), iid[loc].astype(np.float64), raw[loc] / max_rating]))
    return rating_lists, group_index


class RatingData(torch.utils.data.Dataset):
    """One shard's interactions (read.py:108-124)."""

    def __init__(self, rating_array):
        super().__init__()
        self.users = np.asarray(rating_array[0]).astype(int)
        self.items = np.asarray(rating_array[1]).astype(int)
        self.ratings = np.asarray(rating_array[2]).astype(float)

    def __len__(self):
        return len(self.users)

    def __getitem__(self, idx):
        return (torch.tensor(self.users[idx], dtype=torch.long),
                torch.tensor(self.items[idx], dtype=torch.long),
                torch.tensor(self.ratings[idx], dtype=torch.float32))

    def triples(self):
        """(uid int32, iid int32, rating float32) as the device engine wants them."""
        return (self.users.astype(np.int32), self.items.astype(np.int32), self.ratings.astype(np.float32))


class ShardLoader:
    """What loadData returns: the dataset plus batch size / shuffle flag.  The engine
    keeps the shard resident in HBM and forms batch s of an epoch as
    perm[s*batch:(s+1)*batch] on the device, so nothing is iterated on the host."""

    def __init__(self, dataset, batch_size, shuffle, num_workers=0):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.num_workers = num_workers
        self._cache = {}

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def shard_data(self, n_user, n_item, device=None):
        """The shard's HBM layout (built and uploaded on first use, then kept on the loader)."""
        return shard_layouts([self], n_user, n_item, device)[0]

    def eval_set(self):
        from .engine import EvalSet
        if 'eval' not in self._cache:
            self._cache['eval'] = EvalSet(*self.dataset.triples())
        return self._cache['eval']


def shard_layouts(loaders, n_user, n_item, device=None, units_for=None):
    """The HBM layouts of several shards' loaders; the missing ones are built together (engine.build_shards: one native
    call for all of them) and kept on their loaders.  device: required when called from a worker thread, whose current HIP
    device is not the caller's.  units_for: a table width k whose work units are prepared right away as well."""
    return plan_shard_layouts(loaders, n_user, n_item, device, units_for)()


def plan_shard_layouts(loaders, n_user, n_item, device=None, units_for=None):
    """shard_layouts in pieces (engine.LayoutPlan): the function this returns builds the missing layouts (a worker may call it) and returns
    every loader's layout; its attribute `allocate` -- to be called on the request's own thread right after the worker was started --
    makes their device allocations meanwhile."""
    import contextlib
    from .engine import LayoutPlan, pad_dim
    key = ('train', n_user, n_item)
    todo = [l for l in loaders if key not in l._cache]
    on = lambda: torch.cuda.device(device) if device is not None and torch.device(device).type == 'cuda' else contextlib.nullcontext()
    plan = None
    if todo:
        with on():
            plan = LayoutPlan([(l.dataset.users, l.dataset.items, l.dataset.ratings) for l in todo], n_user, n_item, device, units_for=units_for)

    def allocate():
        if plan is not None:
            with on():
                plan.allocate()

    def finish():
        with on():
            if plan is not None:
                for l, sh in zip(todo, plan.build()):
                    l._cache[key] = sh
            out = [l._cache[key] for l in loaders]
            if units_for is not None:
                # the work units of this table width for every shard that lacks them: built side by side (native, outside the GIL),
                # uploaded in one copy
                from .engine import upload_many
                d = pad_dim(int(units_for))
                need = [sh for sh in out if (d, False) not in sh._units]
                if need:
                    built = list(_unit_pool().map(lambda sh: sh.units_host(d), need)) if len(need) > 1 else [need[0].units_host(d)]
                    for sh, (u, n_units, n_rows), dev_u in zip(need, built, upload_many([b[0] for b in built], need[0].device)):
                        sh._units[(d, False)] = (dev_u, n_units, n_rows)
        from .engine import mark
        mark('w: shard_layouts returns')
        return out
    finish.allocate = allocate              # the caller runs this after handing `finish` to a worker (or not at all: finish then does it)
    return finish


_UNIT_POOL = None


def _unit_pool():
    """Host threads for per-shard native calls made from inside a worker of rng.worker_pool() (a pool of its own: a task
    that waited for tasks of its own pool could starve it)."""
    global _UNIT_POOL
    if _UNIT_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        from . import rng
        _UNIT_POOL = ThreadPoolExecutor(max_workers=max(2, min(16, rng.host_cpus())), thread_name_prefix='ure-units')
    return _UNIT_POOL


def as_loader(obj):
    """Accept a ShardLoader or a real torch DataLoader over a RatingData-like dataset."""
    if isinstance(obj, ShardLoader):
        return obj
    ds = getattr(obj, 'dataset', None)
    if ds is not None and all(hasattr(ds, a) for a in ('users', 'items', 'ratings')):
        cached = getattr(obj, '_ure_loader', None)
        if cached is None:
            shuffle = obj.sampler.__class__.__name__ == 'RandomSampler'
            wrapped = ds if isinstance(ds, RatingData) else RatingData([ds.users, ds.items, ds.ratings])
            cached = ShardLoader(wrapped, obj.batch_size, shuffle)
            try:
                obj._ure_loader = cached
            except Exception:
                pass
        return cached
    raise TypeError('expected the result of loadData(RatingData(...)) or a DataLoader over RatingData')


def loadData(data, batch=30000, n_worker=24, shuffle=True):
    """read.py:127-133."""
    return ShardLoader(data, batch, shuffle, n_worker)


def readSparseMat(dir, n_user, n_item, max_rating=5):
    """read.py:136-145 (float16 CSR of ratings; consumed only by 'rating-ot')."""
    from scipy.sparse import coo_matrix
    row, col, raw = _read_csv(dir)
    val = raw / max_rating
    return coo_matrix((val, (row, col)), shape=(n_user, n_item), dtype=np.float16).tocsr()
