"""ctypes binding of libultrare_hip.so (include/ultrare_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module
raises.  Device memory, streams and process groups come from torch (plumbing);
all arithmetic of the hot path happens inside the library's HIP kernels.
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('URE_LIB') or os.path.join(_PKG, 'libultrare_hip.so')      # URE_LIB: experiment builds (tools/) only
ABI_VERSION = 9
MAX_MODELS_PER_CALL = 32

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64


class UreShard(ctypes.Structure):
    """struct ure_shard (include/ultrare_hip.h)."""
    _fields_ = [
        ('ent_oid', _vp), ('ent_r', _vp), ('ent_tag', _vp), ('ent_src', _vp), ('file_tag', _vp), ('inv_stage', _vp), ('inv_off', _vp),
        ('sched', _vp), ('units', _vp), ('n_units', _i32), ('n_active', _i32), ('n_slots', _i64),
        ('U', _vp * 2), ('V', _vp * 2), ('mU', _vp), ('mV', _vp),
        ('U0', _vp), ('V0', _vp), ('lr_host', _vp), ('lazy_rows', _i32),
        ('snapU', _vp), ('snapV', _vp), ('snap_a', _vp), ('snap', _vp), ('row_slot', _vp),
        ('perm', _vp), ('lr', _vp), ('sse', _vp),
        ('N', _i32), ('n_user', _i32), ('n_item', _i32), ('d', _i32),
        ('batch', _i32), ('epochs', _i32),
        ('lam', ctypes.c_float), ('mu', ctypes.c_float),
        ('touch_mode', _i32), ('n_multi', _i32),
        ('file_tags', _vp), ('n_split', _i32),
    ]


class NativeError(RuntimeError):
    pass


_PROTOTYPES = {
    'ure_abi_version': (ctypes.c_int, []),
    'ure_source_hash': (ctypes.c_char_p, []),
    'ure_last_error': (ctypes.c_char_p, []),
    'ure_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_char_p, ctypes.c_int]),
    'ure_job_create': (ctypes.c_int, [ctypes.POINTER(UreShard), ctypes.c_int, ctypes.POINTER(_vp)]),
    'ure_job_destroy': (ctypes.c_int, [_vp]),
    'ure_copy_rows_batch': (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    'ure_epoch_sse_batch': (ctypes.c_int, [_i32, _vp, _vp, _i32, _vp, _vp]),
    'ure_job_shard_steps': (_i64, [_vp, ctypes.c_int]),
    'ure_job_ticks': (_i64, [_vp]),
    'ure_job_train': (ctypes.c_int, [_vp, _i64, _i64, _vp]),
    'ure_job_materialize': (ctypes.c_int, [_vp, _i64, _vp]),
    'ure_job_touch_rows': (ctypes.c_int, [_vp, _vp, _vp]),
    'ure_job_index_read': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _i64, ctypes.POINTER(_i64)]),
    'ure_job_train_profiled': (ctypes.c_int, [_vp, _i64, _i64, _vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
    'ure_host_randperm': (ctypes.c_int, [_vp, ctypes.c_int, _i64, _vp, ctypes.c_int]),
    'ure_host_randperm_tags': (ctypes.c_int, [_vp, ctypes.c_int, _i64, _i32, _vp, ctypes.c_int]),
    'ure_device_randperm_tags_scratch': (_i64, [_i64, _i32]),
    'ure_device_randperm_tags': (ctypes.c_int, [_vp, _i32, _i64, _vp, _i64, _i32, _vp]),
    'ure_device_shuffle_tags_scratch': (_i64, [_i64, _i32]),
    'ure_device_shuffle_tags_flag': (_i64, [_i64, _i32]),
    'ure_device_shuffle_tags': (ctypes.c_int, [_vp, _i32, _i64, _vp, _i64, _i32, _vp]),
    'ure_host_mt_advance': (ctypes.c_int, [_vp, _i64, _i64]),
    'ure_host_mt_jump_blocks': (ctypes.c_int, [_vp, _i64]),
    'ure_host_mt_jump_support': (ctypes.c_int, [_i64, _vp, _i32, ctypes.POINTER(_i32)]),
    'ure_host_mt_charpoly': (ctypes.c_int, [_vp, _i32]),
    'ure_host_draw_int64': (ctypes.c_int, [_vp, _i64, _i64, _i64, _vp]),
    'ure_host_mf_init': (ctypes.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, ctypes.c_int]),
    'ure_host_mf_init_batch': (ctypes.c_int, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, ctypes.c_int]),
    'ure_device_mf_init_scratch': (_i64, [_i32, _i64, _i64]),
    'ure_device_mf_init': (ctypes.c_int, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, ctypes.c_int, _vp]),
    'ure_host_normal_blocks': (ctypes.c_int, [_vp, _i64, ctypes.c_float, ctypes.c_float]),
    'ure_host_normal_blocks_scalar': (ctypes.c_int, [_vp, _i64, _i32]),
    'ure_host_read_csv': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                         ctypes.POINTER(_i64), ctypes.c_int]),
    'ure_host_free': (None, [_vp]),
    'ure_host_partition': (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, ctypes.c_double, _vp, _vp, _vp, _vp, _vp]),
    'ure_host_partition64': (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, ctypes.c_double, _vp, _vp]),
    'ure_host_build_layout': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp,
                                             ctypes.POINTER(_i64), ctypes.POINTER(_i32), _vp, _vp]),
    'ure_host_build_layouts': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _i32, _i32,
                                              ctypes.POINTER(_vp), _vp, _vp, ctypes.c_int]),
    'ure_host_build_layouts_units': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _i32, _i32,
                                                    ctypes.POINTER(_vp), _vp, _vp, _vp, _i32, _vp, ctypes.c_int]),
    'ure_host_build_layouts_units_start': (_i64, [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _i32, _i32,
                                                  ctypes.POINTER(_vp), _vp, _vp, _vp, _i32, _vp, ctypes.c_int, ctypes.POINTER(_vp), _i32, _vp]),
    'ure_host_build_layouts_units_wait': (ctypes.c_int, [_i64]),
    'ure_host_build_units': (ctypes.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, ctypes.POINTER(_i64)]),
    'ure_score': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, _vp, _vp, _vp, _i64, ctypes.c_int, _vp, _vp, _vp]),
    'ure_eval_users': (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    'ure_eval_rank_ratings': (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp]),
    'ure_eval_reduce': (ctypes.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _vp]),
    'ure_eval_series': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, _vp, _vp, _i64, _i64, ctypes.c_int, _vp, _vp,
                                       _vp, _i64, ctypes.c_int, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    'ure_eval_series_compact': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _i32,
                                               ctypes.c_int, _vp, _vp, _vp, _i64, ctypes.c_int, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _i32, _i32, _vp]),
    'ure_score_own_compact': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, ctypes.c_int, _vp, _vp, _i64, ctypes.c_int, _vp, _vp]),
    'ure_eval_series_own': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _i64, ctypes.c_int,
                                           _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    'ure_sum_vectors': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, _i64, _vp, _vp]),
    'ure_eval_subset': (ctypes.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, ctypes.c_int, _vp, _vp]),
    'ure_merge_rows': (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int, _vp]),
    'ure_ot_cost': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'ure_ot_cost_mfma': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'ure_ot_centroids': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    'ure_ot_centroids_members': (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    'ure_kmeans_cost': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'ure_kmeans_centroids': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    'ure_host_kmeans_assign': (ctypes.c_int, [_vp, _i64, _i32, _i64, _vp, ctypes.POINTER(ctypes.c_double)]),
    'ure_ot_assign': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp, ctypes.POINTER(ctypes.c_double)]),
    'ure_ot_potentials': (ctypes.c_int, [_vp, _i64, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_i64), _vp]),
    'ure_ot_assign_warm': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
}
EXPORTS = tuple(_PROTOTYPES)

_lib = None


def lib():
    """Load the library once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f'{LIB_PATH} is missing: run `python -m ultrare_amd.build` '
                              '(there is no CPU fallback for the SISA hot path)')
        # torch first: it bundles its own libamdhip64 / libhsa-runtime64, and the process must
        # end up with ONE HIP runtime -- the one that owns the device pointers torch hands us.
        # Loading this library first would pull /opt/rocm's copy and split the process in two.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.ure_abi_version() != ABI_VERSION:
            raise NativeError(f'ABI mismatch: library {L.ure_abi_version()} != binding {ABI_VERSION}')
        # the library must be the build of THIS tree's sources (they travel together): a stale .so is an error,
        # not something to run silently.  URE_ALLOW_STALE_LIB=1 is for experiment builds (tools/) only.
        from . import build as _build
        have, want = L.ure_source_hash().decode().replace('URE_SRC_HASH=', ''), _build.source_hash()
        if have != want and os.environ.get('URE_ALLOW_STALE_LIB', '0') != '1' and LIB_PATH == _build.LIB:
            raise NativeError(f'{LIB_PATH} was built from other sources (hash {have}, tree {want}): run `python -m ultrare_amd.build`')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ure_last_error().decode(errors='replace')
        err = NativeError(f'{what} failed (code {rc}): {msg}')
        err.code = rc
        raise err


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags['C_CONTIGUOUS']
        return t.ctypes.data
    assert t.is_contiguous()
    return t.data_ptr()


def stream_handle(stream=None):
    """hipStream_t of a torch stream (default: the current stream of the current device)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def ot_assign(dist_kn):
    """Exact balanced OT on the host (ure_ot_assign): dist [k][n] fp32 ->
    (label int32 [n], plan int32 [n][k] in units of 1/(n k), objective)."""
    dist_kn = np.ascontiguousarray(dist_kn, dtype=np.float32)
    k, n = dist_kn.shape
    label = np.empty(n, dtype=np.int32)
    plan = np.empty((n, k), dtype=np.int32)
    obj = ctypes.c_double()
    check(lib().ure_ot_assign(dist_kn.ctypes.data, n, k, label.ctypes.data, plan.ctypes.data, ctypes.byref(obj)),
          'ure_ot_assign')
    return label, plan, obj.value


def ot_assign_warm(dist_kn, pi, want_plan=True):
    """ure_ot_assign_warm: as ot_assign, started from the cluster potentials pi (float64 [k], or None = cold).
    -> (label, plan or None, objective, augmentations made; -1 = the cold solver ran)."""
    dist_kn = np.ascontiguousarray(dist_kn, dtype=np.float32)
    k, n = dist_kn.shape
    label = np.empty(n, dtype=np.int32)
    plan = np.empty((n, k), dtype=np.int32) if want_plan else None
    obj, aug = ctypes.c_double(), _i64()
    pi = None if pi is None else np.ascontiguousarray(pi, dtype=np.float64)
    check(lib().ure_ot_assign_warm(dist_kn.ctypes.data, n, k, None if pi is None else pi.ctypes.data, label.ctypes.data,
                                   plan.ctypes.data if want_plan else None, ctypes.byref(obj), ctypes.byref(aug)), 'ure_ot_assign_warm')
    return label, plan, obj.value, aug.value


def read_csv(path, threads=0):
    """ure_host_read_csv -> (uid int32, iid int32, rating float64) numpy arrays."""
    pu, pi, pr, n = _vp(), _vp(), _vp(), _i64()
    check(lib().ure_host_read_csv(os.fsencode(path), ctypes.byref(pu), ctypes.byref(pi), ctypes.byref(pr), ctypes.byref(n), threads),
          'ure_host_read_csv')
    # the arrays own the library's buffers: they are freed when the last array is (no copy of 14 MB per file)
    m = n.value
    out = []
    for p, ct in ((pu, ctypes.c_int32), (pi, ctypes.c_int32), (pr, ctypes.c_double)):
        if not p or m == 0:
            if p:
                lib().ure_host_free(p)
            out.append(np.zeros(0, dtype=np.dtype(ct)))
            continue
        buf = (ct * m).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.dtype(ct))
        import weakref
        weakref.finalize(buf, lib().ure_host_free, ctypes.c_void_p(p.value))
        out.append(arr)
    return tuple(out)


def partition(uid, iid, rating, shard_of_user, n_shards, max_rating):
    """ure_host_partition: one pass over the rows -> per shard (uid int32, iid int32, rating / max_rating
    float64), file order kept; shard_of_user[u] = -1 drops user u (deleted, or in no group)."""
    uid = np.ascontiguousarray(uid, dtype=np.int32)
    iid = np.ascontiguousarray(iid, dtype=np.int32)
    rating = np.ascontiguousarray(rating, dtype=np.float64)
    shard_of_user = np.ascontiguousarray(shard_of_user, dtype=np.int32)
    counts = np.zeros(n_shards, dtype=np.int64)
    args = (uid.ctypes.data, iid.ctypes.data, rating.ctypes.data, len(uid), shard_of_user.ctypes.data, len(shard_of_user),
            n_shards, float(max_rating), counts.ctypes.data)
    check(lib().ure_host_partition(*args, None, None, None, None), 'ure_host_partition')
    tot = int(counts.sum())
    ou, oi, orr = np.empty(tot, np.int32), np.empty(tot, np.int32), np.empty(tot, np.float64)
    check(lib().ure_host_partition(*args, ou.ctypes.data, oi.ctypes.data, None, orr.ctypes.data), 'ure_host_partition')
    off = np.concatenate([[0], np.cumsum(counts)])
    return [(ou[off[s]:off[s + 1]], oi[off[s]:off[s + 1]], orr[off[s]:off[s + 1]]) for s in range(n_shards)]


def layout_region_words(n, n_user, n_item):
    """int32 words of a packed layout region (ure_host_build_layouts) for n interactions."""
    return 3 * layout_capacity(n, n_user, n_item) + 5 * (n_user + n_item)


def units_capacity_words(n, n_user, n_item, d):
    """int32 words to leave behind a packed layout region for the work units of table width d (ure_host_build_layouts_units): a
    generous bound -- a row has ceil(slots / (8 lanes)) units, workgroups are padded to 256 / lanes units; when it is too small
    after all the native builder says so (n_units = -1) and the units are built on their own."""
    lanes = d // 4 if d <= 32 else d // 8
    rows = n_user + n_item
    return 4 * (2 * (layout_capacity(n, n_user, n_item) // (8 * lanes) + rows) + 2 * (256 // lanes)) + 8


def build_layouts(triples, n_user, n_item, regions, threads=0, units_d=0):
    """ure_host_build_layouts: triples = [(uid int64, iid int64, rating float64)], regions = [int32 numpy views of
    layout_region_words() words] -> (n_slots [S], n_active [S]).  units_d: also the work units of that table width, behind each
    layout (ure_host_build_layouts_units; regions then hold layout_region_words() + units_capacity_words() words)
    -> (n_slots, n_active, n_units [S] (-1: did not fit))."""
    S = len(triples)
    keep = []
    def col(c, dt):
        arr = [np.ascontiguousarray(t[c], dtype=dt) for t in triples]
        keep.append(arr)
        return (_vp * S)(*[a.ctypes.data for a in arr])
    n = np.array([len(t[0]) for t in triples], dtype=np.int64)
    for s, (t, r) in enumerate(zip(triples, regions)):
        assert len(t[0]) == len(t[1]) == len(t[2]) and r.dtype == np.int32 and r.flags['C_CONTIGUOUS']
        assert len(r) >= layout_region_words(len(t[0]), n_user, n_item)
    reg = (_vp * S)(*[r.ctypes.data for r in regions])
    n_slots, n_active = np.zeros(S, dtype=np.int64), np.zeros(S, dtype=np.int32)
    if units_d:
        words = np.array([len(r) for r in regions], dtype=np.int64)
        n_units = np.zeros(S, dtype=np.int64)
        check(lib().ure_host_build_layouts_units(S, col(0, np.int64), col(1, np.int64), col(2, np.float64), n.ctypes.data, n_user, n_item, reg,
                                                 words.ctypes.data, n_slots.ctypes.data, n_active.ctypes.data, int(units_d), n_units.ctypes.data,
                                                 int(threads)), 'ure_host_build_layouts_units')
        return n_slots, n_active, n_units
    check(lib().ure_host_build_layouts(S, col(0, np.int64), col(1, np.int64), col(2, np.float64), n.ctypes.data, n_user, n_item, reg,
                                       n_slots.ctypes.data, n_active.ctypes.data, int(threads)), 'ure_host_build_layouts')
    return n_slots, n_active


class LayoutBuild:
    """ure_host_build_layouts_units under way on a thread of the library (build_layouts_start); result() joins it.  Holds every array the
    native thread reads and writes; an object that is dropped without result() joins the thread first."""

    def __init__(self, handle, keep, outs):
        self.handle, self.keep, self.outs = handle, keep, outs

    def result(self):
        if self.handle:
            h, self.handle = self.handle, 0
            check(lib().ure_host_build_layouts_units_wait(h), 'ure_host_build_layouts_units')
        return self.outs

    def __del__(self):
        try:
            if self.handle:
                h, self.handle = self.handle, 0
                lib().ure_host_build_layouts_units_wait(h)
        except Exception:
            pass


def build_layouts_start(triples, n_user, n_item, regions, threads=0, units_d=0, dev_regions=None, device=-1, stream=None):
    """build_layouts(units_d > 0) started on a native thread by THIS call (no Python worker has to be woken first) -> LayoutBuild.
    dev_regions: device addresses, one per shard -- every shard's region is copied there on `stream` as soon as it is built."""
    S = len(triples)
    keep = []
    def col(c, dt):
        arr = [np.ascontiguousarray(t[c], dtype=dt) for t in triples]
        keep.append(arr)
        ptrs = (_vp * S)(*[a.ctypes.data for a in arr])
        keep.append(ptrs)
        return ptrs
    n = np.array([len(t[0]) for t in triples], dtype=np.int64)
    for t, r in zip(triples, regions):
        assert len(t[0]) == len(t[1]) == len(t[2]) and r.dtype == np.int32 and r.flags['C_CONTIGUOUS']
        assert len(r) >= layout_region_words(len(t[0]), n_user, n_item)
    reg = (_vp * S)(*[r.ctypes.data for r in regions])
    words = np.array([len(r) for r in regions], dtype=np.int64)
    n_slots, n_active, n_units = np.zeros(S, dtype=np.int64), np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int64)
    dreg = (_vp * S)(*[int(a) for a in dev_regions]) if dev_regions is not None else None
    keep += [n, reg, words, list(regions), n_slots, n_active, n_units, dreg]
    handle = lib().ure_host_build_layouts_units_start(S, col(0, np.int64), col(1, np.int64), col(2, np.float64), n.ctypes.data, n_user, n_item, reg,
                                                      words.ctypes.data, n_slots.ctypes.data, n_active.ctypes.data, int(units_d), n_units.ctypes.data,
                                                      int(threads), dreg, int(device), stream_handle(stream) if dreg is not None else None)
    if not handle:
        raise NativeError('ure_host_build_layouts_units_start: ' + lib().ure_last_error().decode(errors='replace'))
    return LayoutBuild(handle, keep, (n_slots, n_active, n_units))


def layout_capacity(n, n_user, n_item):
    """Slots ure_host_build_layout may write for n interactions (every row's segment is padded to a multiple of 8)."""
    return 2 * n + 8 * (n_user + n_item) + 8


def partition64(uid, iid, rating, shard_of_user, n_shards, max_rating):
    """ure_host_partition64: one counting pass and one writing pass over the rows -> per shard the [3, N_s] float64 array
    readRating returns (uid, iid, rating / max_rating; file order kept), views of one block."""
    uid = np.ascontiguousarray(uid, dtype=np.int32)
    iid = np.ascontiguousarray(iid, dtype=np.int32)
    rating = np.ascontiguousarray(rating, dtype=np.float64)
    shard_of_user = np.ascontiguousarray(shard_of_user, dtype=np.int32)
    counts = np.zeros(n_shards, dtype=np.int64)
    args = (uid.ctypes.data, iid.ctypes.data, rating.ctypes.data, len(uid), shard_of_user.ctypes.data, len(shard_of_user),
            n_shards, float(max_rating), counts.ctypes.data)
    check(lib().ure_host_partition64(*args, None), 'ure_host_partition64')
    block = np.empty(3 * int(counts.sum()), dtype=np.float64)
    check(lib().ure_host_partition64(*args, block.ctypes.data), 'ure_host_partition64')
    off = np.concatenate([[0], np.cumsum(3 * counts)])
    return [block[off[s]:off[s + 1]].reshape(3, int(counts[s])) for s in range(n_shards)]


def build_layout(uid, iid, rating, n_user, n_item, want_pos=False, out=None):
    """ure_host_build_layout on numpy triples -> dict of numpy arrays + counts.  out = (ent_oid, ent_r, ent_src, sched):
    caller-owned destinations of layout_capacity() slots / [n_user + n_item, 4] (e.g. views of a pinned staging buffer)."""
    n = len(uid)
    cap = layout_capacity(n, n_user, n_item)
    if out is not None:
        ent_oid, ent_r, ent_src, sched = out
        assert ent_oid.dtype == np.int32 and ent_r.dtype == np.float32 and ent_src.dtype == np.int32 and sched.dtype == np.int32
        assert min(len(ent_oid), len(ent_r), len(ent_src)) >= cap and sched.shape == (n_user + n_item, 4)
        assert all(a.flags['C_CONTIGUOUS'] for a in out)
    else:
        ent_oid = np.empty(cap, dtype=np.int32)
        ent_r = np.empty(cap, dtype=np.float32)
        ent_src = np.empty(cap, dtype=np.int32)
        sched = np.empty((n_user + n_item, 4), dtype=np.int32)
    u_pos = np.empty(n, dtype=np.int32) if want_pos else None
    i_pos = np.empty(n, dtype=np.int32) if want_pos else None
    ns, na = _i64(), _i32()
    check(lib().ure_host_build_layout(uid.ctypes.data, iid.ctypes.data, rating.ctypes.data, n, n_user, n_item,
                                      ent_oid.ctypes.data, ent_r.ctypes.data, ent_src.ctypes.data, sched.ctypes.data,
                                      ctypes.byref(ns), ctypes.byref(na),
                                      u_pos.ctypes.data if want_pos else None, i_pos.ctypes.data if want_pos else None),
          'ure_host_build_layout')
    k = ns.value
    return {'ent_oid': ent_oid[:k], 'ent_r': ent_r[:k], 'ent_src': ent_src[:k], 'sched': sched, 'n_slots': k,
            'n_active': na.value, 'u_pos': u_pos, 'i_pos': i_pos}


def build_units(sched, n_active, d, unit_passes=1):
    """ure_host_build_units: the work units [n_units, 4] of a row schedule for table width d."""
    sched = np.ascontiguousarray(sched, dtype=np.int32)
    n = _i64()
    check(lib().ure_host_build_units(sched.ctypes.data, n_active, d, int(unit_passes), None, 0, ctypes.byref(n)), 'ure_host_build_units')
    units = np.empty((max(n.value, 1), 4), dtype=np.int32)
    check(lib().ure_host_build_units(sched.ctypes.data, n_active, d, int(unit_passes), units.ctypes.data, n.value, ctypes.byref(n)), 'ure_host_build_units')
    return units[:n.value]
