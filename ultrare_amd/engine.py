"""Host side of the MI355X engine: HBM layout of a shard, training jobs, evaluation.

This file holds no arithmetic of the hot path -- that lives in csrc/*.hip behind
include/ultrare_hip.h -- only layout building (numpy, once per shard), device memory
(torch tensors) and launches.

HBM layout of one shard:
  slot arrays  ent_oid[n_slots] i32 | ent_r[n_slots] f32 | ent_tag[2][n_slots] u16 -- one 8-aligned,
               padded segment per destination row (users and items), in schedule order
  ent_src[n_slots] i32     slot -> file-order index of its interaction (-1 in padding)
  file_tag[N] u16          scratch: batch of interaction j in the epoch being prepared
  sched[n_user+n_item][4] i32   {row id, first slot, end slot, nnz}, heaviest row first
  units[n_units][4] i32    work units of the step kernel for one table width (ShardData.units(d))
  U[2][n_user][d] V[2][n_item][d] f32 ping-pong weights ; mU, mV momentum
  perm[epochs][N] i32 ; lr[epochs] f32 ; sse[epochs][n_user] f32 (per-user squared error)
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _native as nv

SCORE_PARTIALS = 2048     # URE_SCORE_PARTIALS of the C ABI
SERIES_SCRATCH_BYTES = 256 << 20     # prediction scratch of one ure_eval_series call
LAZY_ROWS = os.environ.get('URE_LAZY_ROWS', '1') != '0'
TOUCH_MAX_STEPS = 32000              # kTouchMaxSteps of csrc/mf_touch.h (epochs longer than 64 steps run in windows of 64)
# touch mode, epochs of several windows: rows of up to this many scan passes are one work item.  Full MF at the 25 M shape, us per
# launch: 106.7 at 1 (a unit per pass), 95.5 at 2, 97.0 at 4, 118.8 at 8, 157.3 at 16, 386.7 at 64 -- a lane group walks its row's
# passes one after the other (a scan, a compaction and a gather each), units work theirs side by side
TOUCH_ROW_PASSES = 2
# ... and the work units of the longer rows take this many passes each (a unit skips the passes without a slot of the step by their masks;
# fewer, larger units: the step visits 1,700 workgroups of them instead of 13,600).  The same leg: 105.8 us at 1, 81.7 at 2, 68.0 at 4,
# 62.9 at 8, 61.6 at 16
TOUCH_UNIT_PASSES = 8
TOUCH_AHEAD_MAX_STEPS = 63           # kAheadMaxSteps of csrc/mf_touch.h
INDEX_MAX_STEPS = 1008               # kIdxMaxSteps of csrc/mf_index.h (touch_mode 3: the epoch's slots sorted by step)
INDEX_HEAVY_SLOTS = 16               # touch_mode 3: rows with at least this many slots per step on average get a workgroup per step ...
INDEX_SPLIT_SLOTS = 384              # ... and with this many, one per 256 slots of the step
INDEX_HEAVY_MAX = 256                # kIdxHeavyMax
INDEX_SHORT_EPOCH_MAX_D = 64         # auto rule: epochs of at most 63 steps take touch_mode 3 up to this table width (beyond it touch_mode 2 / 1)
TOUCH_MIN_TABLE_BYTES = 256 << 20    # auto rule: the job's live rows (w, m, second buffer) exceed the Infinity Cache


# Host timeline of a call (URE_HOST_TRACE=1): (label, seconds) marks that tools/profile_e2e.py prints; off by default.
HOST_TRACE = [] if os.environ.get('URE_HOST_TRACE', '0') == '1' else None


def mark(label):
    if HOST_TRACE is not None:
        import time
        HOST_TRACE.append((label, time.perf_counter()))


_SIDE_STREAMS = {}


def side_streams(dev, n):
    """n HIP streams beside the current one, kept per device (independent chains of short launches -- the test series of a
    request's shards -- run on them side by side)."""
    key = str(torch.device(dev))
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(dev))
    return have[:n]


def to_device_async(arr, dev):
    """A small host array -> device tensor without stalling the host: through a pooled pinned buffer and an asynchronous
    copy (from pageable memory `.to(device)` waits until the stream has reached and finished the copy)."""
    from . import rng
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if torch.device(dev).type != 'cuda' or t.numel() == 0:
        return t.to(dev)
    stage = rng.SMALL.take(tuple(t.shape), t.dtype)
    stage.numpy()[...] = t.numpy()                       # (a plain memcpy: torch's copy_ would go through its intra-op thread pool)
    out = stage.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    rng.SMALL.give(stage, ev)
    return out


def upload_many(arrays, dev):
    """Several small host arrays -> device tensors through ONE pinned staging buffer and ONE asynchronous copy (each array
    starts on a 64-byte boundary of one device allocation).  A request of 16 shards sends ~50 such descriptors (work units,
    row lists, schedules of scalars): one by one they cost 0.1-0.2 ms of host time each."""
    from . import rng
    arrays = [np.ascontiguousarray(a) for a in arrays]
    if torch.device(dev).type != 'cuda':
        return [torch.from_numpy(a.copy()).to(dev) for a in arrays]
    offs, at = [], 0
    for a in arrays:
        offs.append(at)
        at += (a.nbytes + 63) // 64 * 64
    if at == 0:
        return [torch.from_numpy(a.copy()).to(dev) for a in arrays]
    stage = rng.SMALL.take((at,), torch.uint8)
    host = stage.numpy()
    for a, o in zip(arrays, offs):
        host[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    blob = stage.to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    rng.SMALL.give(stage, ev)
    out = []
    for a, o in zip(arrays, offs):
        out.append(blob[o:o + a.nbytes].view(torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype).view(a.shape))
    return out


def pad_dim(d):
    """Table width used on the device: next power of two >= max(d, 4).  Padding
    columns are zero at init; their gradient and decay keep them exactly zero."""
    p = 4
    while p < d:
        p *= 2
    if p > 256:
        raise ValueError(f'embedding width {d} > 256 is not supported by the gfx950 kernels')
    return p


def _device():
    if not torch.cuda.is_available():
        raise nv.NativeError('no HIP device visible: the SISA hot path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def build_shards(triples, n_user, n_item, device=None, keep_positions=False, units_for=None):
    """The HBM layouts of the shards of one call: triples = [(uid, iid, rating)] -> [ShardData].
    ONE native call builds every layout, side by side on host threads, packed into one pinned staging buffer (pooled); all of them go up
    in ONE asynchronous copy into one device allocation, and the engine-side scratch of all shards (batch tags, inverse-permutation
    stages) comes from two fills.  units_for = a table width k: the work units of that width are built by the same native call and
    travel in the same copy (ShardData.units finds them).  From pageable numpy arrays, shard after shard, the same 22 MB of a 5-shard
    ml-1m call took 9-10 ms of a 20 ms Sisa.learn (profiles/r03/NOTES.md)."""
    return LayoutPlan(triples, n_user, n_item, device, units_for).build(keep_positions)


class LayoutPlan:
    """build_shards in pieces that a request overlaps: __init__ checks the triples and takes the host staging buffer; allocate() (on
    the request's own thread and stream) makes every device allocation and fill from UPPER BOUNDS of the slot counts -- nothing in it
    waits for the layouts --; build() (any thread: a worker of the request, started BEFORE allocate()) runs the native builder, waits
    for allocate(), queues the one copy and returns the ShardData, which hold ADDRESSES; the tensor views of a layout's arrays are
    made when somebody asks for them."""

    def __init__(self, triples, n_user, n_item, device=None, units_for=None):
        from . import rng
        self.n_user, self.n_item = int(n_user), int(n_item)
        self.dev = dev = device or _device()
        self.on_gpu = torch.device(dev).type == 'cuda'
        cols = []
        for uid, iid, rating in triples:
            uid, iid, rating = np.asarray(uid), np.asarray(iid), np.asarray(rating)
            n = len(uid)
            if n == 0:
                raise ValueError('a shard needs at least one interaction')
            if not (len(iid) == n and len(rating) == n):
                raise ValueError('uid / iid / rating lengths differ')
            # (ids outside [0, n_user) x [0, n_item) are refused by the native builder, which walks the arrays anyway)
            cols.append((uid, iid, rating))
        self.cols = cols
        al = lambda x: (x + 7) // 8 * 8                              # every array starts on a 32-byte boundary
        self.d_units = pad_dim(int(units_for)) if units_for else 0
        self.words = [al(nv.layout_region_words(len(c[0]), self.n_user, self.n_item) +
                         (nv.units_capacity_words(len(c[0]), self.n_user, self.n_item, self.d_units) if self.d_units else 0)) for c in cols]
        self.off = np.concatenate([[0], np.cumsum(self.words)]).astype(np.int64)
        self.stage = rng.STAGING.take((int(self.off[-1]),), torch.int32)
        # the native builder starts HERE, on a thread of the library's own (a Python worker took 0.3-0.4 ms to get going while the calling
        # thread held the interpreter); build() joins it
        # ... and sends every shard's layout to the device as soon as it is built (the blob is allocated here for that; as ONE copy behind the
        # last shard, the 1.1 GB of a 32-shard request at the 25 M shape stood between the build and the job for 28 ms)
        self._async = None
        self.blob = None
        self._uploaded = False
        self._stream = None
        if self.d_units:
            host = self.stage.numpy()
            try:
                dev_regions, dev_id, stream = None, -1, None
                if self.on_gpu:
                    with torch.cuda.device(dev):
                        self.blob = torch.empty(int(self.off[-1]), dtype=torch.int32, device=dev)
                        stream = torch.cuda.current_stream(dev)
                    dev_regions = [self.blob.data_ptr() + 4 * int(self.off[s]) for s in range(len(cols))]
                    dev_id = self.blob.device.index
                    self._uploaded, self._stream = True, stream
                mark('layouts native start')
                self._async = nv.build_layouts_start(cols, self.n_user, self.n_item, [host[self.off[s]:self.off[s + 1]] for s in range(len(cols))],
                                                     threads=min(len(cols), rng.host_cpus()), units_d=self.d_units, dev_regions=dev_regions, device=dev_id,
                                                     stream=stream)
            except BaseException:
                rng.STAGING.give(self.stage, None)
                raise
        self._owner = __import__('threading').current_thread()
        self._allocated = __import__('threading').Event()
        self._alloc_error = None

    def allocate(self):
        from . import rng
        cols, dev, al = self.cols, self.dev, (lambda x: (x + 7) // 8 * 8)
        try:
            if self.blob is None:
                self.blob = torch.empty(int(self.off[-1]), dtype=torch.int32, device=dev)
            # engine-side scratch: batch tags (0xFFFF matches no batch; three buffers: touch_mode 2 prepares two epochs ahead) and
            # the stages of the inverse permutation.  Sized for the most slots a shard of n interactions can have.
            self.t_words = [(al(3 * nv.layout_capacity(len(c[0]), self.n_user, self.n_item)), al(len(c[0]))) for c in cols]
            self.z_words = [(al(len(c[0])), al(max(((len(c[0]) + 2047) // 2048) * ((len(c[0]) + 2047) // 2048 + 1), 1) if (len(c[0]) + 2047) // 2048 <= 1024 else 1))
                            for c in cols]
            self.tags = torch.full((sum(a + b for a, b in self.t_words),), -1, dtype=torch.int16, device=dev)
            self.zeros = torch.zeros(sum(a + b for a, b in self.z_words), dtype=torch.int32, device=dev)
            self.allocated = None
            if self.on_gpu:
                self.allocated = torch.cuda.Event()
                self.allocated.record(torch.cuda.current_stream(dev))
        except BaseException as e:
            self._alloc_error = e
            raise
        finally:
            self._allocated.set()
        return self

    def build(self, keep_positions=False):
        from . import rng
        if not self._allocated.is_set() and __import__('threading').current_thread() is self._owner:
            self.allocate()                         # (one thread does everything: build_shards)
        S, cols, off, host = len(self.cols), self.cols, self.off, self.stage.numpy()
        n_user, n_item, dev, d_units = self.n_user, self.n_item, self.dev, self.d_units
        rows = n_user + n_item
        handed_back = False
        try:
            try:
                if self._async is not None:
                    built = self._async.result()
                    self._async = None
                else:
                    mark('w: layouts native start')
                    built = nv.build_layouts(cols, n_user, n_item, [host[off[s]:off[s + 1]] for s in range(S)], threads=min(S, rng.host_cpus()), units_d=d_units)
            except nv.NativeError as e:
                if getattr(e, 'code', None) == -2:             # (-3: a shard beyond 2^31 slots -- reported as it is)
                    raise ValueError(f'user or item id outside [0, n_user) x [0, n_item): {e}') from None
                raise
            mark('w: layouts built (native)')
            if not self._allocated.wait(60.0):
                raise RuntimeError('LayoutPlan.build() on a worker: nobody called allocate()')
            if self._alloc_error is not None:
                raise RuntimeError('the device allocations of the layouts failed') from self._alloc_error
            n_slots, n_active = built[0], built[1]
            n_units = built[2] if d_units else [-1] * S
            al = lambda x: (x + 7) // 8 * 8
            units_at = [al(3 * int(k) + 5 * rows) for k in n_slots]                       # where a shard's units start inside its region (when built)
            used = [a + (al(4 * int(u)) if u > 0 else 0) for a, u in zip(units_at, n_units)]
            end = int(off[S - 1]) + used[S - 1]
            pinned = self.stage.is_pinned() and self.on_gpu
            ready = None
            with (torch.cuda.device(dev) if self.on_gpu else contextlib.nullcontext()):
                if self.on_gpu:
                    # (the allocations and fills were queued on the planner's stream, which need not be this thread's)
                    torch.cuda.current_stream(dev).wait_event(self.allocated)
                if not self._uploaded:
                    self.blob[:end].copy_(self.stage[:end], non_blocking=pinned)      # ONE copy (the slack between the regions travels along)
                if self.on_gpu:
                    # whoever trains on a layout from another stream waits for this event first: TrainJob does
                    ready = torch.cuda.Event()
                    ready.record(self._stream if self._uploaded else torch.cuda.current_stream(dev))     # (the stream the copies went to)
            mark('w: layouts copy queued')
            out, t_at, z_at = [], 0, 0
            for s, (uid, iid, rating) in enumerate(cols):
                sh = object.__new__(ShardData)
                n, k = len(uid), int(n_slots[s])
                sh.N, sh.n_user, sh.n_item, sh.device = n, n_user, n_item, dev
                sh.n_slots, sh.n_active = k, int(n_active[s])
                o = int(off[s])
                ta, tb = self.t_words[s]
                za, zb = self.z_words[s]
                f32, i32, i16 = torch.float32, torch.int32, torch.int16
                # name -> (tensor it lives in, first element, elements, dtype, shape)
                sh._where = {'ent_oid': (self.blob, o, k, i32, None), 'ent_r': (self.blob, o + k, k, f32, None), 'ent_src': (self.blob, o + 2 * k, k, i32, None),
                             'sched': (self.blob, o + 3 * k, 4 * rows, i32, (rows, 4)), '_row_slot': (self.blob, o + 3 * k + 4 * rows, rows, i32, None),
                             'ent_tag': (self.tags, t_at, 3 * k, i16, (3, k)), 'file_tag': (self.tags, t_at + ta, n, i16, None),
                             'inv_stage': (self.zeros, z_at, n, i32, None), 'inv_off': (self.zeros, z_at + za, zb, i32, None)}
                t_at += ta + tb
                z_at += za + zb
                sh._blob = self.blob
                # (the schedule's head stays on the host -- what a job reads of it --; the whole of it, 3.5 MB per shard at the 25 M shape and 6 ms of
                # copying for a 32-shard request, comes back from the device when somebody asks: units of another table width)
                sh._sched_head = host[o + 3 * k:o + 3 * k + 4 * min(rows, INDEX_HEAVY_MAX)].reshape(-1, 4).copy()
                sh.max_row = int(sh._sched_head[0, 3])
                sh.u_pos = sh.i_pos = None
                if keep_positions:                                        # host copies of every interaction's two slots (tests, tools)
                    lay = nv.build_layout(np.ascontiguousarray(uid, dtype=np.int32), np.ascontiguousarray(iid, dtype=np.int32),
                                          np.ascontiguousarray(rating, dtype=np.float32), n_user, n_item, want_pos=True)
                    sh.u_pos, sh.i_pos = lay['u_pos'], lay['i_pos']
                sh._units = {}
                if n_units[s] > 0:
                    ua = o + units_at[s]
                    sh._units[(d_units, False)] = (self.blob[ua:ua + 4 * int(n_units[s])].view(int(n_units[s]), 4), int(n_units[s]), sh.n_active)
                sh.ready = ready
                out.append(sh)
            rng.STAGING.give(self.stage, ready if pinned else None)
            handed_back = True
        finally:
            if not handed_back:
                rng.STAGING.give(self.stage, None)                  # whatever went wrong, the pooled staging buffer goes back
        with ShardData._count_lock:
            ShardData.built += S
        return out


class ShardData:
    """One shard's interactions laid out for the step kernel: every destination row
    (users, then items) owns an 8-aligned, padded segment of one slot array; segments
    follow the row schedule (heaviest first).  Built by build_shards()."""

    def __init__(self, uid, iid, rating, n_user, n_item, device=None, keep_positions=False):
        self.__dict__.update(build_shards([(uid, iid, rating)], n_user, n_item, device, keep_positions)[0].__dict__)

    _count_lock = __import__('threading').Lock()
    built = 0            # layouts built (and uploaded) by this process: lets a measurement show that its timed call paid for them

    def ptr(self, name):
        """Device address of one of the layout's arrays (no tensor is made for it)."""
        base, first, _, dtype, _ = self._where[name]
        return base.data_ptr() + first * base.element_size()

    def __getattr__(self, name):
        # the layout's arrays as tensors (ent_oid, ent_r, ent_src, sched, ent_tag, file_tag, inv_stage, inv_off, _row_slot): views made on first use
        where = self.__dict__.get('_where')
        if name == '_sched_host' and where is not None:
            got = self.sched.cpu().numpy()                   # (synchronises; rare)
            self.__dict__[name] = got
            return got
        if where is None or name not in where:
            raise AttributeError(name)
        base, first, count, dtype, shape = where[name]
        t = base[first:first + count]
        if dtype != base.dtype:
            t = t.view(dtype)
        if shape is not None:
            t = t.view(*shape)
        self.__dict__[name] = t
        return t

    def row_slot(self):
        """Device int32 [n_user + n_item]: a row's index in the schedule when it is one of the n_active rows with
        interactions in this shard (the rows a compact snapshot stores), -1 otherwise."""
        return self._row_slot

    def units(self, d, touch=False, min_passes=1, unit_passes=1):
        """The work units of the step kernel for table width d (device int32 [n_units, 4]).  touch: only the rows
        longer than one scan pass get units (-> (units, n_multi)); the others are worked off per step from a
        compaction of the rows that are trained in it (csrc/mf_touch.h)."""
        key = (d, bool(touch)) if min_passes == 1 and unit_passes == 1 else (d, bool(touch), int(min_passes), int(unit_passes))
        if key not in self._units:
            u, n_units, n_rows = self.units_host(d, touch, min_passes, unit_passes)
            self._units[key] = (to_device_async(u, self.device), n_units, n_rows)
        dev_u, n_units, n_rows = self._units[key]
        return (dev_u, n_units, n_rows) if touch else dev_u

    def units_host(self, d, touch=False, min_passes=1, unit_passes=1):
        """The host half of units(): -> (int32 [max(n_units, 1), 4] array to upload, n_units, rows covered).  touch: only the
        rows of more than min_passes scan passes get units (the others are worked off per row); unit_passes = scan passes per unit."""
        n_rows = self.n_active
        if touch:
            lanes = d // 4 if d <= 32 else d // 8
            seg = self._sched_host[:self.n_active, 2] - self._sched_host[:self.n_active, 1]
            n_rows = int(np.count_nonzero(seg > 8 * lanes * min_passes))
            assert n_rows == 0 or (seg[:n_rows] > 8 * lanes * min_passes).all()          # the schedule is heaviest first
        u = nv.build_units(self._sched_host, n_rows, d, unit_passes)
        return (u if len(u) else np.full((1, 4), -1, np.int32)), len(u), n_rows

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in
                   (self.ent_oid, self.ent_r, self.ent_tag, self.ent_src, self.file_tag, self.sched))


_CLOSED_FORM = {}


def closed_form_scalars(lr_host, steps, lam, mu):
    """a_e with w = a_e * w0 after epoch e for a row that only decays: the optimizer's recurrence
    g = lam*w; m = mu*m + g (m = g on the first step); w -= lr*m, in float64 (as ure_job_materialize)."""
    key = (np.asarray(lr_host, dtype=np.float32).tobytes(), int(steps), float(lam), float(mu))
    if key not in _CLOSED_FORM:
        if len(_CLOSED_FORM) > 256:
            _CLOSED_FORM.clear()
        _CLOSED_FORM[key] = _closed_form_scalars(lr_host, steps, lam, mu)
    return _CLOSED_FORM[key].copy()


def _closed_form_scalars(lr_host, steps, lam, mu):
    a, b, out, t = 1.0, 0.0, [], 0
    for e in range(len(lr_host)):
        lr = float(lr_host[e])
        for _ in range(steps):
            b = lam * a if t == 0 else mu * b + lam * a
            a -= lr * b
            t += 1
        out.append(a)
    return np.asarray(out, dtype=np.float32)


class _States:
    """TrainJob.state: per shard the views of the job's device memory ({'U', 'V', 'mU', 'mV', 'perm', 'sse', 'snap' ...}), made when first asked for."""

    def __init__(self, n, make):
        import weakref
        # (a weak reference to the job's method: job -> state -> job would be a cycle, and a job's 0.7 GB of device memory -- 660 MB per
        # request measured -- would wait for the cyclic collector instead of going back when the last reference to the job does)
        self._n, self._make, self._got = n, weakref.WeakMethod(make), {}

    def __len__(self):
        return self._n

    def __getitem__(self, s):
        s = s + self._n if s < 0 else s
        if not 0 <= s < self._n:
            raise IndexError(s)
        if s not in self._got:
            make = self._make()
            if make is None:
                raise ReferenceError('the job of these views is gone')
            self._got[s] = make(s)
        return self._got[s]

    def __iter__(self):
        return (self[s] for s in range(self._n))


class TrainJob:
    """A set of shards trained side by side, one optimizer step of each per launch.

    shards : list[ShardData]
    inits  : list of (U0 [n_user,k], V0 [n_item,k]) float32 numpy / CPU tensors
    perms  : list of int32 [epochs, N_s] arrays (numpy or torch; CPU or device), or int16 batch tags (rng.epoch_tags)
    """

    def __init__(self, shards, inits, perms, k, batch, epochs, lr, lam, momentum, lr_decay=1.0, lr_step=50, lazy_rows=None, snapshots=False,
                 touch=None, final_only=False, epoch_reads=False):
        """touch: None = the auto rule (touch mode when the job's live rows exceed the Infinity Cache AND the caller reads the tables
        at epoch ends only (epoch_reads) or after the last epoch only (final_only): a job in touch mode cannot be read inside an
        epoch), True / False, or 'index' (touch_mode 3 whatever the epoch length)."""
        assert len(shards) == len(inits) == len(perms) and len(shards) > 0
        self.shards, self.k, self.d = shards, int(k), pad_dim(int(k))
        self.batch, self.epochs = int(batch), int(epochs)
        dev = shards[0].device
        self.device = dev
        # StepLR(step_size=50, gamma): lr of epoch t (scratch.py:69,79-80)
        lr_host = np.array([lr * (lr_decay ** (t // lr_step)) for t in range(self.epochs)], dtype=np.float32)
        self._lr_host = lr_host
        # rows a shard never touches only decay: advance them in closed form when the tables are read
        # (URE_LAZY_ROWS=0 streams them every step, exactly as the reference's dense optimizer does)
        self.lazy_rows = LAZY_ROWS if lazy_rows is None else bool(lazy_rows)
        self._fresh = 0          # ticks for which the lazily advanced rows are up to date
        # touch mode (csrc/mf_touch.h): a step visits only the rows it trains, the others are advanced in closed form when
        # they are next trained.  Pays when the tables do not fit the caches (configs[3]); URE_TOUCH=0/1 overrides the rule.
        steps_all = [(sh.N + self.batch - 1) // self.batch for sh in shards]
        by_rule = False
        if touch is None:
            env = os.environ.get('URE_TOUCH', 'auto')
            live = sum(sh.n_active for sh in shards) * self.d * 12
            by_rule = env == 'auto' and live > TOUCH_MIN_TABLE_BYTES and (final_only or epoch_reads)
            touch = (env == '1') or by_rule
        self.touch = bool(touch) and self.lazy_rows and max(steps_all) <= TOUCH_MAX_STEPS
        # touch_mode 2 (csrc/mf_touch.h, "masks one epoch ahead"): no dense pass at the epoch starts; for callers that read the tables
        # only after the last epoch (final_only) and epochs of at most 63 steps.  URE_TOUCH_AHEAD=0 keeps mode 1.
        # Short epochs (<= 63 steps) of NARROW rows: sorting the epoch's slots by step (touch_mode 3) costs ~3 ms per epoch whatever the row width, and saves the
        # scan of every trained row's slots in every step -- most of the traffic at 64-byte rows.  BASELINE.json configs[3]'s shape (32 shards, 27 steps per
        # epoch), epoch time in touch_mode 2 -> 3: d = 16 6.32 -> 4.95 ms, d = 32 6.25 -> 5.87, d = 64 8.36 -> 8.02, d = 128 12.18 -> 13.17
        # (profiles/r04/exp_short_epochs_index.txt).  Under the auto rule touch_mode 3 is taken up to d = 64.
        force_index = touch == 'index' or os.environ.get('URE_TOUCH_INDEX', '1') == '2' or (by_rule and self.d <= INDEX_SHORT_EPOCH_MAX_D and
                                                                                             os.environ.get('URE_TOUCH_INDEX', '1') != '0')
        self.ahead = (self.touch and bool(final_only) and max(steps_all) <= TOUCH_AHEAD_MAX_STEPS and snapshots in (False, None, 'compact')
                      and os.environ.get('URE_TOUCH_AHEAD', '1') != '0' and not force_index)
        # touch_mode 3 (csrc/mf_index.h): epochs of more than 63 steps -- the epoch's slots are sorted by step at its start and a step
        # launches over exactly the rows it trains (64-step windows look at every work unit in every step).  URE_TOUCH_INDEX=0 keeps windows.
        self.index = (self.touch and not self.ahead and max(steps_all) <= INDEX_MAX_STEPS and self.batch <= 200000 and
                      (force_index or (max(steps_all) > TOUCH_AHEAD_MAX_STEPS and os.environ.get('URE_TOUCH_INDEX', '1') != '0')))
        # end-of-epoch snapshots: 'compact' keeps the n_active rows with interactions only (every other row is a_e * w0 and is
        # rebuilt where it is read: ure_eval_series_compact; needs lazy_rows), True / 'full' keeps complete tables
        self.snapshots = ('compact' if self.lazy_rows else 'full') if snapshots == 'compact' else ('full' if snapshots else False)
        steps_of = [(sh.N + self.batch - 1) // self.batch for sh in shards]
        small = [lr_host] + ([closed_form_scalars(lr_host, st_, float(np.float32(lam)), float(np.float32(momentum))) for st_ in steps_of] if self.snapshots else [])
        small = upload_many(small, dev)                  # the learning rates and every shard's closed-form scalars: one copy
        self.lr = small[0]
        self._chunks = []        # per shard whose permutations are uploaded in chunks: [(first epoch after the chunk, event), ...]
        descs = (nv.UreShard * len(shards))()
        mark('job: start')
        cur = torch.cuda.current_stream(dev) if torch.device(dev).type == 'cuda' else None
        seen = set()
        for sh in shards:
            if sh.ready is not None and id(sh.ready) not in seen:
                seen.add(id(sh.ready))
                cur.wait_event(sh.ready)
        # every float table of every shard from ONE zero-filled allocation (a request of 16 shards made ~130 small allocations
        # and fills here: 8-12 ms of host time beside 16 busy worker threads), the snapshots from another.  The descriptors are
        # filled from ADDRESSES (base + offset); the views of the tables are made when somebody asks for them (self.state), which a
        # request does after its launches are queued.
        al = lambda x: (x + 63) // 64 * 64
        d = self.d
        sizes = [(2 * sh.n_user * d, 2 * sh.n_item * d, sh.n_user * d, sh.n_item * d, self.epochs * sh.n_user,
                  sh.n_user * d if self.lazy_rows else 0, sh.n_item * d if self.lazy_rows else 0) for sh in shards]
        pool = torch.zeros(sum(al(x) for sz in sizes for x in sz), dtype=torch.float32, device=dev)
        snap_rows = [(sh.n_active if self.snapshots == 'compact' else sh.n_user + sh.n_item) if self.snapshots else 0 for sh in shards]
        snap_pool = torch.empty(sum(al(self.epochs * r * d) for r in snap_rows), dtype=torch.float32, device=dev) if self.snapshots else None
        self._pool, self._snap_pool, self._small = pool, snap_pool, small
        base, snap_base = pool.data_ptr(), (snap_pool.data_ptr() if snap_pool is not None else 0)
        at = snap_at = 0
        mark('job: pools')
        self._offs, self._snap_offs, self._perms, self._init_src = [], [], [], []
        copies = []                                   # (src, rows, dst, dst2) of the start tables: one launch below
        for s, (sh, (U0, V0), perm) in enumerate(zip(shards, inits, perms)):
            srcs = []
            for t, n_rows in ((U0, sh.n_user), (V0, sh.n_item)):
                ev = getattr(t, '_ure_event', None)
                if ev is not None and id(ev) not in seen:               # uploaded on a side stream (rng.shard_draws_async)
                    seen.add(id(ev))
                    cur.wait_event(ev)
                if ev is not None:
                    t.record_stream(cur)
                t = torch.as_tensor(t, dtype=torch.float32)
                assert t.shape == (n_rows, self.k)
                if t.device != torch.device(dev) or not t.is_contiguous():
                    t = t.to(dev, non_blocking=True).contiguous()
                srcs.append(t)
            self._init_src.append(srcs)                                 # (alive until the copy below has run: released by close())
            off = []
            for x in sizes[s]:
                off.append(at)
                at += al(x)
            self._offs.append(off)
            pU, pV, pmU, pmV, psse, pU0, pV0 = (base + 4 * o for o in off)
            perm = torch.as_tensor(perm)
            if getattr(perm, '_ure_chunks', None) is not None:      # still arriving in chunks of epochs (rng.shard_draws_async)
                self._chunks.append(list(perm._ure_chunks))
            elif getattr(perm, '_ure_event', None) is not None:     # uploaded on a side stream (rng.epoch_perms_async)
                cur.wait_event(perm._ure_event)
            assert perm.shape == (self.epochs, sh.N), f'perm of shard {s} must be [epochs, N]'
            # int16: not permutations but the batch tags the host made of them (rng.epoch_tags; struct ure_shard: file_tags)
            as_tags = perm.dtype == torch.int16
            perm = perm.to(device=dev, dtype=torch.int16 if as_tags else torch.int32).contiguous()
            self._perms.append(perm)
            D = descs[s]
            for name in ('ent_oid', 'ent_r', 'ent_tag', 'ent_src', 'file_tag', 'inv_stage', 'inv_off', 'sched'):
                setattr(D, name, sh.ptr(name))
            if self.index:
                # no work units: the step's items come from the epoch's index.  Rows by weight class (slots per step on average)
                nnz, st_s = sh._sched_head[:min(sh.n_active, INDEX_HEAVY_MAX), 3], steps_all[s]
                D.n_multi = int(np.count_nonzero(nnz >= INDEX_HEAVY_SLOTS * st_s))
                D.n_split = min(D.n_multi, int(np.count_nonzero(nnz >= INDEX_SPLIT_SLOTS * st_s)))
                units, n_units = sh.ptr('sched'), 0
            elif self.touch:
                # epochs of several windows (more than 64 steps): a row of up to TOUCH_ROW_PASSES scan passes stays ONE work item -- its
                # lane group skips the passes without a slot of the step (csrc/mf_touch.h: pass masks) -- instead of one unit per pass
                long_epochs = max(steps_all) > 64
                units, n_units, n_multi = sh.units(self.d, touch=True, min_passes=TOUCH_ROW_PASSES if long_epochs else 1,
                                                   unit_passes=TOUCH_UNIT_PASSES if long_epochs else 1)
                D.n_multi = n_multi
                units = nv.ptr(units)
            else:
                units = sh.units(self.d)
                n_units = units.shape[0]
                units = nv.ptr(units)
            D.units, D.n_units, D.n_active, D.n_slots = units, n_units, sh.n_active, sh.n_slots
            D.U[0], D.U[1] = pU, pU + 4 * sh.n_user * d
            D.V[0], D.V[1] = pV, pV + 4 * sh.n_item * d
            D.mU, D.mV = pmU, pmV
            D.perm, D.lr, D.sse = (None if as_tags else nv.ptr(perm)), nv.ptr(self.lr), psse
            D.file_tags = nv.ptr(perm) if as_tags else None
            D.N, D.n_user, D.n_item, D.d = sh.N, sh.n_user, sh.n_item, self.d
            D.batch, D.epochs = self.batch, self.epochs
            D.lam, D.mu = float(lam), float(momentum)
            D.touch_mode = (3 if self.index else 2 if self.ahead else 1) if self.touch else 0
            self._snap_offs.append(snap_at)
            if self.snapshots:
                D.snap_a = small[1 + s].data_ptr()
                if self.snapshots == 'compact':
                    D.snap, D.row_slot = snap_base + 4 * snap_at, sh.ptr('_row_slot')
                else:
                    D.snapU, D.snapV = snap_base + 4 * snap_at, snap_base + 4 * (snap_at + self.epochs * sh.n_user * d)
                snap_at += al(self.epochs * snap_rows[s] * d)
            if self.lazy_rows:
                D.U0, D.V0, D.lr_host, D.lazy_rows = pU0, pV0, lr_host.ctypes.data, 1
            copies += [(srcs[0].data_ptr(), sh.n_user, pU, pU0 if self.lazy_rows else 0), (srcs[1].data_ptr(), sh.n_item, pV, pV0 if self.lazy_rows else 0)]
        # the start tables into buffer 0 (and the closed form's copy), every shard's in one launch
        n_c = len(copies)
        src_a, dst_a, dst2_a = (ctypes.c_void_p * n_c)(*[c[0] for c in copies]), (ctypes.c_void_p * n_c)(*[c[2] for c in copies]), \
            (ctypes.c_void_p * n_c)(*[c[3] or None for c in copies])
        rows_a = (ctypes.c_int64 * n_c)(*[c[1] for c in copies])
        nv.check(nv.lib().ure_copy_rows_batch(n_c, src_a, dst_a, dst2_a if self.lazy_rows else None, rows_a, self.k, d, nv.stream_handle()), 'ure_copy_rows_batch')
        self._init_src = None             # (allocator: their memory is reused only after the streams they were recorded on have passed this point)
        self.state = _States(len(shards), self._state_of)
        self._descs = descs
        self._job = ctypes.c_void_p()
        mark('job: tables allocated, descriptors filled')
        nv.check(nv.lib().ure_job_create(descs, len(shards), ctypes.byref(self._job)), 'ure_job_create')
        mark('job: ure_job_create')
        self.ticks = int(nv.lib().ure_job_ticks(self._job))
        self.shard_steps = [int(nv.lib().ure_job_shard_steps(self._job, s)) for s in range(len(shards))]
        self.done = 0

    def _state_of(self, s):
        sh, d, pool = self.shards[s], self.d, self._pool
        o = self._offs[s]
        st = {'U': pool[o[0]:o[0] + 2 * sh.n_user * d].view(2, sh.n_user, d), 'V': pool[o[1]:o[1] + 2 * sh.n_item * d].view(2, sh.n_item, d),
              'mU': pool[o[2]:o[2] + sh.n_user * d].view(sh.n_user, d), 'mV': pool[o[3]:o[3] + sh.n_item * d].view(sh.n_item, d),
              'perm': self._perms[s], 'sse': pool[o[4]:o[4] + self.epochs * sh.n_user].view(self.epochs, sh.n_user)}
        if self.lazy_rows:
            st.update(U0=pool[o[5]:o[5] + sh.n_user * d].view(sh.n_user, d), V0=pool[o[6]:o[6] + sh.n_item * d].view(sh.n_item, d))
        if self.snapshots:
            at, sp = self._snap_offs[s], self._snap_pool
            st.update(snap_a=self._small[1 + s])
            if self.snapshots == 'compact':
                st.update(snap=sp[at:at + self.epochs * sh.n_active * d].view(self.epochs, sh.n_active, d))
            else:
                nU = self.epochs * sh.n_user * d
                st.update(snapU=sp[at:at + nU].view(self.epochs, sh.n_user, d),
                          snapV=sp[at + nU:at + nU + self.epochs * sh.n_item * d].view(self.epochs, sh.n_item, d))
        return st

    def steps_per_epoch(self, s):
        return self.shard_steps[s] // self.epochs

    def run(self, n_ticks=None, stream=None):
        """Enqueue the next n_ticks optimizer steps of every shard (default: all)."""
        t1 = self.ticks if n_ticks is None else min(self.ticks, self.done + int(n_ticks))
        while t1 > self.done:
            t_next = t1
            if self._chunks:
                # the launches of tick t read the permutation of the epoch AFTER the one a shard is in (the batch tags are
                # prepared one epoch ahead): wait for the chunk that holds it, and launch only up to where the next one is needed
                min_steps = min(self.steps_per_epoch(s) for s in range(len(self.shards)))
                need = self.done // min_steps + (2 if self.ahead else 1)  # newest epoch any shard can read at tick self.done
                st = stream if stream is not None else torch.cuda.current_stream(self.device)
                horizon = self.epochs
                def wait(chunk):                                        # host: until the worker queued the upload; device: until it is done
                    if HOST_TRACE is not None and not chunk[1].is_set():
                        import time
                        t0 = time.perf_counter()
                        chunk[1].wait()
                        self.chunk_wait_s = getattr(self, 'chunk_wait_s', 0.0) + time.perf_counter() - t0
                    chunk[1].wait()
                    if chunk[2][0] is None:
                        raise nv.NativeError('the permutation worker failed before this chunk was uploaded')
                    marks = getattr(self, 'wait_marks', None)
                    if marks is None:
                        st.wait_event(chunk[2][0])
                    else:           # (measurement: how long the stream stood still for the chunk -- bench.py takes it out of the launches' time)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        st.wait_event(chunk[2][0])
                        e1.record(st)
                        marks.append((e0, e1))
                for ch in self._chunks:
                    while ch and ch[0][0] <= need and len(ch) > 1:      # chunks that end at or before `need`, then the one holding it
                        wait(ch.pop(0))
                    if ch:
                        wait(ch[0])
                        horizon = min(horizon, ch[0][0])                # epochs < horizon have arrived (after these waits)
                        if len(ch) == 1 and ch[0][0] >= self.epochs:
                            ch.pop(0)
                self._chunks = [ch for ch in self._chunks if ch]
                if horizon < self.epochs:
                    t_next = min(t_next, max(self.done + 1, (horizon - (2 if self.ahead else 1)) * min_steps))
            nv.check(nv.lib().ure_job_train(self._job, self.done, t_next, nv.stream_handle(stream)), 'ure_job_train')
            self.done = t_next
        return self.done

    def run_profiled(self, n_ticks, stream=None):
        """Like run(), with every launch bracketed by HIP events (synchronises).
        -> (step kernel ms, launches, tag kernel ms, launches)."""
        t1 = min(self.ticks, self.done + int(n_ticks))
        sm, am = ctypes.c_double(), ctypes.c_double()
        ns, na = ctypes.c_int64(), ctypes.c_int64()
        nv.check(nv.lib().ure_job_train_profiled(self._job, self.done, t1, nv.stream_handle(stream), ctypes.byref(sm),
                                                 ctypes.byref(ns), ctypes.byref(am), ctypes.byref(na)),
                 'ure_job_train_profiled')
        self.done = t1
        return sm.value, ns.value, am.value, na.value

    def run_epochs(self, n_epochs, stream=None):
        """Single-shard convenience: advance by whole epochs."""
        assert len(self.shards) == 1
        return self.run(n_epochs * self.steps_per_epoch(0), stream)

    @staticmethod
    def snapshot_bytes(shards, epochs, d, mode):
        """Device bytes the end-of-epoch snapshots of `shards` need ('compact': active rows only)."""
        rows = sum((sh.n_active if mode == 'compact' else sh.n_user + sh.n_item) for sh in shards)
        return int(epochs) * rows * pad_dim(d) * 4

    def snapshot(self, s, epoch):
        """(U, V) of shard s as they were at the end of `epoch` (padded width; needs full snapshots)."""
        st = self.state[s]
        return st['snapU'][epoch], st['snapV'][epoch]

    def snapshots_of(self, s):
        """All end-of-epoch tables of shard s: (U [epochs, n_user, d], V [epochs, n_item, d]) (full snapshots)."""
        st = self.state[s]
        return st['snapU'], st['snapV']

    def own_scores(self, s, eval_set, stream=None):
        """own [epochs, n] float32: the score of every pair of `eval_set` under shard s's own model after every epoch, from the compact
        snapshots (ure_score_own_compact) -- the half of a series that does not depend on other shards; EvalSet.evaluate_series_own adds
        the fixed models and ranks.  After run()."""
        assert self.snapshots == 'compact' and self.done == self.ticks
        own = torch.empty(self.epochs, max(eval_set.n, 1), dtype=torch.float32, device=self.device)
        state, sh = self.state[s], self.shards[s]
        snap = state['snap']
        if eval_set.n:
            nv.check(nv.lib().ure_score_own_compact(nv.ptr(snap), snap.stride(0), nv.ptr(sh.row_slot()), nv.ptr(state['U0']), nv.ptr(state['V0']),
                                                    nv.ptr(state['snap_a']), sh.n_user, self.epochs, nv.ptr(eval_set.uid), nv.ptr(eval_set.iid), eval_set.n,
                                                    self.d, nv.ptr(own), nv.stream_handle(stream)), 'ure_score_own_compact')
        return own

    def evaluate_series(self, s, eval_set, fixed, out, stream=None, subset=None, lane=0):
        """scratch.py:83-97 for every epoch of shard s on `eval_set`: member e = the ensemble `fixed` + the shard's model
        after epoch e, from whichever kind of snapshots the job keeps.  out: device float64 [epochs, 3].
        subset = (EvalSet.subset_of plan, out_sub [epochs, 3]): the same numbers for a subset of the set's users as well."""
        st = self.state[s]
        if self.snapshots == 'compact':
            sh = self.shards[s]
            return eval_set.evaluate_series_compact(fixed, st['snap'], sh.row_slot(), st['U0'], st['V0'], st['snap_a'], sh.n_user, self.d, out, stream,
                                                    subset=subset, lane=lane)
        return eval_set.evaluate_series(fixed, st['snapU'], st['snapV'], self.d, out, stream, subset=subset, lane=lane)

    def evaluate_series_pair(self, s, test_ev, total_ev, fixed, out_test, out_total, stream=None, lane=0):
        """The two per-epoch series of scratch.py:83-97 -- the shard's own test set and the total test set.  Where the first is a subset of
        the second (the reference builds the total set from the shards' sets: config.py:144-148) ONE series on the total set yields both."""
        plan = test_ev.subset_of(total_ev)
        if plan is None:
            self.evaluate_series(s, test_ev, fixed, out_test, stream, lane=lane)
            return self.evaluate_series(s, total_ev, fixed, out_total, stream, lane=lane)
        return self.evaluate_series(s, total_ev, fixed, out_total, stream, subset=(plan, out_test), lane=lane)

    def materialize(self, stream=None):
        """Bring the lazily advanced rows (lazy_rows) up to date in the current tables."""
        if self.lazy_rows and self._fresh != self.done:
            nv.check(nv.lib().ure_job_materialize(self._job, self.done, nv.stream_handle(stream)), 'ure_job_materialize')
            self._fresh = self.done

    def tables(self, s):
        """Current (U, V) of shard s as device views [rows, k]."""
        self.materialize()
        cur = min(self.done, self.shard_steps[s]) & 1
        st = self.state[s]
        return st['U'][cur, :, :self.k], st['V'][cur, :, :self.k]

    def padded_tables(self, s):
        self.materialize()
        cur = min(self.done, self.shard_steps[s]) & 1
        st = self.state[s]
        return st['U'][cur], st['V'][cur]

    def touch_rows_per_step(self):
        """Touch mode: rows the step kernel reads and rewrites per optimizer step, per shard (average over the shard's
        current window of up to 64 steps, from the window's row masks; synchronises).  None otherwise."""
        if not self.touch:
            return None
        out, win = np.zeros(len(self.shards), dtype=np.int64), np.zeros(len(self.shards), dtype=np.int64)
        nv.check(nv.lib().ure_job_touch_rows(self._job, out.ctypes.data, win.ctypes.data), 'ure_job_touch_rows')
        return [float(n) / max(int(w), 1) for n, w in zip(out, win)]

    _INDEX_ARRAYS = {'step_begin': (0, np.uint32, 1), 'step_item': (1, np.uint32, 1), 'items': (2, np.int32, 4), 'sslot': (3, np.uint32, 4),
                     'W': (4, np.uint64, 1), 'heavy_cnt': (5, np.uint32, 1), 'heavy_cum': (6, np.uint32, 257)}

    def index_array(self, s, name):
        """Test aid (touch_mode 3): one array of shard s's slot index of its current epoch as a host array (ure_job_index_read; synchronises)."""
        which, dtype, cols = self._INDEX_ARRAYS[name]
        n = ctypes.c_int64()
        nv.check(nv.lib().ure_job_index_read(self._job, s, which, None, 0, ctypes.byref(n)), 'ure_job_index_read')
        out = np.empty(n.value // np.dtype(dtype).itemsize, dtype=dtype)
        nv.check(nv.lib().ure_job_index_read(self._job, s, which, out.ctypes.data, out.nbytes, ctypes.byref(n)), 'ure_job_index_read')
        return out.reshape(-1, cols) if cols > 1 else out

    def epoch_sse(self, s):
        """Per-epoch sum of squared training errors (host float64 array; synchronises)."""
        return self.epoch_sse_queue([s]).cpu().numpy()[0]

    def epoch_sse_queue(self, which=None, out=None, stream=None):
        """epoch_sse of the shards `which` (default: all) as a DEVICE float64 tensor [n, epochs] (`out`, if given): one launch
        (ure_epoch_sse_batch: a fixed summation order, the same whoever asks), nothing synchronises."""
        which = list(range(len(self.shards))) if which is None else list(which)
        n = len(which)
        out = torch.empty(n, self.epochs, dtype=torch.float64, device=self.device) if out is None else out
        assert out.shape == (n, self.epochs) and out.dtype == torch.float64 and out.is_contiguous()
        base = self._pool.data_ptr()
        ptrs = (ctypes.c_void_p * n)(*[base + 4 * self._offs[s][4] for s in which])
        rows = (ctypes.c_int64 * n)(*[self.shards[s].n_user for s in which])
        nv.check(nv.lib().ure_epoch_sse_batch(n, ptrs, rows, self.epochs, out.data_ptr(), nv.stream_handle(stream)), 'ure_epoch_sse_batch')
        return out

    def epoch_sse_all(self):
        """epoch_sse of every shard, [n_shards, epochs] (host; synchronises once)."""
        return self.epoch_sse_queue().cpu().numpy()

    def check_tags(self):
        """Device-made batch tags (rng.device_tags): did a workgroup of perm_tags_kernel give up on a shuffle?  It cannot -- and its tags
        then match no batch, so nothing trained on a wrong permutation -- but a job that trained on fewer rows must not pass silently.
        Reads one small flag array (synchronises): called by close() and by callers right after they read a job's results."""
        from . import rng
        rng.device_tags_check(self._perms or [])

    def close(self):
        if self._job:
            nv.lib().ure_job_destroy(self._job)
            self._job = ctypes.c_void_p()
            try:
                self.check_tags()                 # (ure_job_destroy has waited for the device)
            finally:
                # the device memory goes back now: tables, snapshots, batch tags and what made them
                self._pool = self._snap_pool = self._small = None
                self._perms = []
                if isinstance(getattr(self, 'state', None), _States):
                    self.state._got.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_LOG2_TAB = np.log2(np.arange(2, 11)).astype(np.float64)      # utils.py:210
_IDCG = 1.0 + np.sum(np.ones(9) / _LOG2_TAB)                   # computeDCG(np.ones(10)), utils.py:207
_LOG2_TAB = np.concatenate([_LOG2_TAB, [_IDCG]])


class FixedBase:
    """The fixed models of a series given as their running sum on the set's pairs (ScoreCache.base): n models, base [n_pairs] float32."""

    def __init__(self, n, base):
        self.n, self.base = int(n), base

    def __len__(self):
        return self.n


class ScoreCache:
    """Score vectors of a request's models on ONE test set, each model scored once (ure_score with one model: what it adds to an ensemble's
    running sum), and an ensemble's base as their sum in list order (ure_sum_vectors) -- the additions ure_score makes over a list of
    tables, so a series evaluated from such a base has the same bits.  Everything is queued on the current stream."""
    LIMIT_BYTES = 4 << 30

    def __init__(self, eval_set, d):
        self.ev, self.d, self.vec = eval_set, int(d), {}

    def fits(self, n_vectors):
        return int(n_vectors) * 4 * self.ev.n <= self.LIMIT_BYTES

    def base(self, before):
        """before: [(key, get_tables, arg)] in the ensemble's order; get_tables(arg) -> padded (U, V) of the model `key` names.  -> FixedBase"""
        if not before:
            return FixedBase(0, None)
        for key, get, arg in before:
            if key not in self.vec:
                U, V = get(arg)
                self.vec[key] = self.ev.score_vector(U, V, self.d)
        out = torch.empty(max(self.ev.n, 1), dtype=torch.float32, device=self.ev.device)
        n = len(before)
        ptrs = (ctypes.c_void_p * n)(*[self.vec[key].data_ptr() for key, _, _ in before])
        nv.check(nv.lib().ure_sum_vectors(ptrs, n, self.ev.n, nv.ptr(out), nv.stream_handle()), 'ure_sum_vectors')
        return FixedBase(n, out)


def _fixed_args(fixed, d, own_base):
    """-> (U pointers, V pointers, n_fixed, base pointer) of a series call for `fixed` = a list of padded (U, V) or a FixedBase."""
    if isinstance(fixed, FixedBase):
        return None, None, fixed.n, (nv.ptr(fixed.base) if fixed.n else nv.ptr(own_base))
    for U, V in fixed:
        assert U.is_contiguous() and V.is_contiguous() and U.shape[1] == d and V.shape[1] == d
    Up = (ctypes.c_void_p * max(len(fixed), 1))(*[U.data_ptr() for U, _ in fixed])
    Vp = (ctypes.c_void_p * max(len(fixed), 1))(*[V.data_ptr() for _, V in fixed])
    return Up, Vp, len(fixed), nv.ptr(own_base)


class EvalSet:
    """Test interactions grouped by user (first-appearance order, utils.py:156-163)
    and resident on the device, plus the output buffers of the two eval kernels."""

    def __init__(self, uid, iid, rating, device=None):
        uid = np.ascontiguousarray(uid, dtype=np.int32)
        iid = np.ascontiguousarray(iid, dtype=np.int32)
        rating = np.ascontiguousarray(rating, dtype=np.float32)
        self.n = len(uid)
        self.device = device or _device()
        self.n_wide = self.n_half = 0
        if self.n:
            _, first = np.unique(uid, return_index=True)
            users = uid[np.sort(first)]                       # first-appearance order (utils.py:156-163) ...
            rank = np.empty(int(uid.max()) + 1, dtype=np.int64)
            rank[users] = np.arange(len(users))
            counts = np.bincount(rank[uid], minlength=len(users))
            # ... inside three classes: users with more than 32 test items first (a wavefront each in the ranking kernel), then those
            # with 17 .. 32 (two per wavefront), then the others (four per wavefront).  The metrics are means over users: their order
            # does not enter.
            klass = np.where(counts > 32, 0, np.where(counts > 16, 1, 2))
            self.n_wide, self.n_half = int((klass == 0).sum()), int((klass == 1).sum())
            cls = np.argsort(klass, kind='stable')
            users, counts = users[cls], counts[cls]
            rank[users] = np.arange(len(users))
            order = np.argsort(rank[uid], kind='stable')
        else:
            users, order, counts = np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64)
        off = np.zeros(len(users) + 1, dtype=np.int32)
        np.cumsum(counts, out=off[1:])
        self.users, self.n_users = users, len(users)
        dev = self.device
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.uid, self.iid, self.rating, self.off = to(uid[order]), to(iid[order]), to(rating[order]), to(off)
        self.pred = torch.zeros(max(self.n, 1), dtype=torch.float32, device=dev)
        self.hits = torch.zeros(max(self.n_users, 1), dtype=torch.int32, device=dev)
        self.ndcg = torch.zeros(max(self.n_users, 1), dtype=torch.float64, device=dev)
        self.sse = torch.zeros(SCORE_PARTIALS, dtype=torch.float64, device=dev)
        self.log2 = to(_LOG2_TAB)
        self.order = order
        # host copies in the set's own order: what subset_of() compares
        self._h_iid, self._h_rating, self._h_off = iid[order], rating[order], off
        self._subsets = {}
        # the ranking of the ratings is a property of the test set: once, here
        self.top_rating = torch.empty(max(self.n_users, 1) * 10, dtype=torch.int32, device=dev)
        if self.n_users:
            nv.check(nv.lib().ure_eval_rank_ratings(nv.ptr(self.off), self.n_users, nv.ptr(self.rating), nv.ptr(self.top_rating),
                                                    nv.stream_handle()), 'ure_eval_rank_ratings')

    def subset_of(self, total):
        """This set as a subset of `total`: -> {'users': device int32 [n_users] (this set's users as indices into total's user order), 'n': n_users,
        'pairs': device int32 [n] (its pairs as indices into total's pair order), 'n_pairs': n}
        when every user of this set is in `total` with exactly the same (item, rating) rows in the same order -- the reference's total test
        set is the shards' test sets side by side (config.py:144-148) --, else None.  Checked on the host once per pair of sets and kept
        (the sets live with their loaders).  With a plan, a series on `total` also yields this set's three numbers (ure_eval_subset)
        instead of a second series on the same models and pairs."""
        if total is self or self.n == 0 or total.n == 0:
            return None
        key = id(total)
        hit = self._subsets.get(key)
        if hit is not None and hit[0]() is total:
            return hit[1]
        import weakref
        plan = None
        pos = getattr(total, '_user_pos', None)
        if pos is None:
            pos = np.full(int(total.users.max()) + 1, -1, dtype=np.int64)
            pos[total.users] = np.arange(total.n_users)
            total._user_pos = pos
        mine = self.users.astype(np.int64)
        if mine.max() < len(pos):
            at = pos[mine]
            cnt = np.diff(self._h_off).astype(np.int64)
            if (at >= 0).all() and np.array_equal(cnt, np.diff(total._h_off).astype(np.int64)[at]):
                # the rows of every user, side by side in this set's order: equal items and ratings, in the same order
                src = np.repeat(total._h_off[at].astype(np.int64) - self._h_off[:-1].astype(np.int64), cnt) + np.arange(self.n, dtype=np.int64)
                if np.array_equal(total._h_iid[src], self._h_iid) and np.array_equal(total._h_rating[src], self._h_rating):
                    up = upload_many([at.astype(np.int32), src.astype(np.int32)], self.device)
                    plan = {'users': up[0], 'n': int(self.n_users), 'pairs': up[1], 'n_pairs': int(self.n)}
        self._subsets[key] = (weakref.ref(total), plan)
        return plan

    def _subset_after(self, subset, m, e0, st, b):
        """ure_eval_subset on the m members a series call just left in the scratch buffers b (subset = (plan of subset_of, out [E, 3]))."""
        plan, out_sub = subset
        nv.check(nv.lib().ure_eval_subset(nv.ptr(plan['users']), plan['n'], nv.ptr(plan['pairs']), plan['n_pairs'], nv.ptr(b['pred']), nv.ptr(self.rating),
                                          nv.ptr(b['hits']), nv.ptr(b['ndcg']), self.n, self.n_users, m, nv.ptr(out_sub[e0]), st), 'ure_eval_subset')

    def evaluate(self, models, d, stream=None, top_k=10, out=None):
        """baseTest (utils.py:115-187) for an ensemble: `models` = list of (U, V) device
        tensors with row stride d (the padded width).  Returns (rmse, ndcg, hr)."""
        assert top_k == 10, 'the kernel implements the reference default top_k=10'
        if self.n == 0:
            if out is not None:
                return out.fill_(float('nan'))
            return float('nan'), float('nan'), float('nan')
        L, st = nv.lib(), nv.stream_handle(stream)
        S = len(models)
        for c0 in range(0, S, nv.MAX_MODELS_PER_CALL):
            chunk = models[c0:c0 + nv.MAX_MODELS_PER_CALL]
            for U, V in chunk:
                assert U.is_contiguous() and V.is_contiguous() and U.shape[1] == d and V.shape[1] == d
            Up = (ctypes.c_void_p * len(chunk))(*[U.data_ptr() for U, _ in chunk])
            Vp = (ctypes.c_void_p * len(chunk))(*[V.data_ptr() for _, V in chunk])
            nv.check(L.ure_score(Up, Vp, len(chunk), S, int(c0 == 0), int(c0 + len(chunk) >= S),
                                 nv.ptr(self.uid), nv.ptr(self.iid), nv.ptr(self.rating), self.n, d,
                                 nv.ptr(self.pred), nv.ptr(self.sse), st), 'ure_score')
        nv.check(L.ure_eval_users(nv.ptr(self.off), self.n_users, nv.ptr(self.pred), nv.ptr(self.rating),
                                  nv.ptr(self.log2), nv.ptr(self.hits), nv.ptr(self.ndcg), nv.ptr(self.top_rating), self.n_wide, self.n_half, st),
                 'ure_eval_users')
        if out is not None:
            # queued evaluation: (rmse, ndcg, hr) land in `out` (device, 3 float64); nothing synchronises
            nv.check(L.ure_eval_reduce(nv.ptr(self.hits), nv.ptr(self.ndcg), self.n_users, nv.ptr(self.sse), self.n,
                                       nv.ptr(out), st), 'ure_eval_reduce')
            return out
        sse = float(self.sse.cpu().numpy().sum())
        hits = self.hits[:self.n_users].cpu().numpy()
        ndcg = self.ndcg[:self.n_users].cpu().numpy()
        rmse = float(np.sqrt(sse / self.n))
        return rmse, float(np.mean(ndcg)), float(np.mean(hits / top_k))

    def _series_buffers(self, E, lane=0):
        """Scratch of one series call (kept on the set): -> (buffers, members per call).  lane: series that run side by side on
        different streams (Sisa: the shards of a request) take a scratch of their own each; a lane belongs to ONE stream."""
        per_call = max(1, min(E, SERIES_SCRATCH_BYTES // (4 * self.n)))
        lanes = self.__dict__.setdefault('_series_lanes', {})
        if lanes.get(lane, (0, None))[0] < per_call:
            dev = self.device
            lanes[lane] = (per_call, {'base': torch.empty(self.n, dtype=torch.float32, device=dev),
                                      'pred': torch.empty(per_call, self.n, dtype=torch.float32, device=dev),
                                      'sse': torch.empty(per_call, SCORE_PARTIALS, dtype=torch.float64, device=dev),
                                      'hits': torch.empty(per_call, max(self.n_users, 1), dtype=torch.int32, device=dev),
                                      'ndcg': torch.empty(per_call, max(self.n_users, 1), dtype=torch.float64, device=dev)})
        return lanes[lane][1], per_call

    def score_vector(self, U, V, d, stream=None):
        """What the model (U, V) (padded, device) adds to an ensemble's running sum on this set's pairs: 0 + <u, v> per pair, float32
        [n] in the set's own order (ure_score with one model, first = 1, last = 0)."""
        assert U.is_contiguous() and V.is_contiguous() and U.shape[1] == d and V.shape[1] == d
        out = torch.empty(max(self.n, 1), dtype=torch.float32, device=self.device)
        if self.n:
            Up, Vp = (ctypes.c_void_p * 1)(U.data_ptr()), (ctypes.c_void_p * 1)(V.data_ptr())
            nv.check(nv.lib().ure_score(Up, Vp, 1, 1, 1, 0, nv.ptr(self.uid), nv.ptr(self.iid), nv.ptr(self.rating), self.n, d, nv.ptr(out), None,
                                        nv.stream_handle(stream)), 'ure_score')
        return out

    def evaluate_series(self, fixed, U_series, V_series, d, out, stream=None, subset=None, lane=0):
        """scratch.py:83-97 for every epoch of a shard in four launches (ure_eval_series): member e of
        the series is the ensemble `fixed` + [(U_series[e], V_series[e])]; out[e] (device float64
        [E, 3]) receives its (rmse, ndcg, hr).  Nothing synchronises."""
        E = int(U_series.shape[0])
        assert out.shape == (E, 3) and out.dtype == torch.float64 and out.is_contiguous()
        assert U_series.is_contiguous() and V_series.is_contiguous() and U_series.shape[2] == d and V_series.shape[2] == d
        if self.n == 0:
            return out.fill_(float('nan'))
        L, st = nv.lib(), nv.stream_handle(stream)
        b, per_call = self._series_buffers(E, lane)
        Up, Vp, n_fixed, base_ptr = _fixed_args(fixed, d, b['base'])
        for e0 in range(0, E, per_call):
            m = min(per_call, E - e0)
            nv.check(L.ure_eval_series(Up, Vp, n_fixed, nv.ptr(U_series[e0]), nv.ptr(V_series[e0]), U_series.stride(0),
                                       V_series.stride(0), m, nv.ptr(self.uid), nv.ptr(self.iid), nv.ptr(self.rating), self.n, d,
                                       nv.ptr(self.off), self.n_users, nv.ptr(self.log2), base_ptr, nv.ptr(b['pred']),
                                       nv.ptr(b['sse']), nv.ptr(b['hits']), nv.ptr(b['ndcg']), nv.ptr(out[e0]), nv.ptr(self.top_rating),
                                       self.n_wide, self.n_half, st),
                     'ure_eval_series')
            if subset is not None:
                self._subset_after(subset, m, e0, st, b)
        return out

    def evaluate_series_compact(self, fixed, snap, row_slot, U0, V0, snap_a, n_user_rows, d, out, stream=None, subset=None, lane=0):
        """evaluate_series on COMPACT snapshots (ure_eval_series_compact): snap [E, n_active, d] holds the rows with
        interactions in the shard, row_slot maps a row id to its place in it (-1: the row is snap_a[e] * (U0 | V0)[row])."""
        E = int(snap.shape[0])
        assert out.shape == (E, 3) and out.dtype == torch.float64 and out.is_contiguous()
        assert snap.is_contiguous() and snap.shape[2] == d and U0.is_contiguous() and V0.is_contiguous() and U0.shape[1] == d and V0.shape[1] == d
        assert row_slot.dtype == torch.int32 and row_slot.numel() == U0.shape[0] + V0.shape[0] and snap_a.numel() == E
        if self.n == 0:
            return out.fill_(float('nan'))
        L, st = nv.lib(), nv.stream_handle(stream)
        b, per_call = self._series_buffers(E, lane)
        Up, Vp, n_fixed, base_ptr = _fixed_args(fixed, d, b['base'])
        for e0 in range(0, E, per_call):
            m = min(per_call, E - e0)
            nv.check(L.ure_eval_series_compact(Up, Vp, n_fixed, nv.ptr(snap[e0]), snap.stride(0), nv.ptr(row_slot), nv.ptr(U0), nv.ptr(V0),
                                               nv.ptr(snap_a[e0:]), int(n_user_rows), m, nv.ptr(self.uid), nv.ptr(self.iid), nv.ptr(self.rating),
                                               self.n, d, nv.ptr(self.off), self.n_users, nv.ptr(self.log2), base_ptr, nv.ptr(b['pred']),
                                               nv.ptr(b['sse']), nv.ptr(b['hits']), nv.ptr(b['ndcg']), nv.ptr(out[e0]), nv.ptr(self.top_rating),
                                               self.n_wide, self.n_half, st), 'ure_eval_series_compact')
            if subset is not None:
                self._subset_after(subset, m, e0, st, b)
        return out

    def evaluate_series_own(self, fixed, own, d, out, stream=None):
        """The second half of a series whose own scores own [E, n] exist (ure_eval_series_own)."""
        E = int(own.shape[0])
        assert out.shape == (E, 3) and out.dtype == torch.float64 and out.is_contiguous() and own.is_contiguous() and own.shape[1] == self.n
        L, st = nv.lib(), nv.stream_handle(stream)
        b, per_call = self._series_buffers(E)
        Up, Vp, n_fixed, base_ptr = _fixed_args(fixed, d, b['base'])
        for e0 in range(0, E, per_call):
            m = min(per_call, E - e0)
            nv.check(L.ure_eval_series_own(Up, Vp, n_fixed, nv.ptr(own[e0]), m, nv.ptr(self.uid), nv.ptr(self.iid), nv.ptr(self.rating), self.n, d,
                                           nv.ptr(self.off), self.n_users, nv.ptr(self.log2), base_ptr, nv.ptr(b['pred']), nv.ptr(b['sse']),
                                           nv.ptr(b['hits']), nv.ptr(b['ndcg']), nv.ptr(out[e0]), nv.ptr(self.top_rating), self.n_wide, self.n_half, st),
                     'ure_eval_series_own')
        return out

    def predictions(self):
        """Ensemble predictions of the last evaluate() in the caller's original row order."""
        out = np.empty(self.n, dtype=np.float32)
        out[self.order] = self.pred[:self.n].cpu().numpy()
        return out


def merge_rows(dst, src, rows, stream=None):
    """dst[rows] = src[rows] on the device (sisa.py:55-56)."""
    if not torch.is_tensor(rows):
        rows = to_device_async(np.asarray(rows, dtype=np.int64), dst.device)
    assert rows.dtype == torch.int64 and rows.device == dst.device
    assert dst.is_contiguous() and src.is_contiguous() and dst.shape == src.shape
    nv.check(nv.lib().ure_merge_rows(nv.ptr(dst), nv.ptr(src), nv.ptr(rows), rows.numel(), dst.shape[1],
                                     nv.stream_handle(stream)), 'ure_merge_rows')
    return dst
