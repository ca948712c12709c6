"""Sisa.learn / unlearn / test with the reference's signatures (method/sisa.py:7-118).

Two execution modes, same final models and log0 (training of a shard is deterministic
and isolated, and the RNG stream is consumed in the reference's order in both):

  sequential (default)  shard after shard, with the reference's per-epoch group / total
                        tests that average the models trained so far (scratch.py:83-97).
  parallel              `param.parallel = True` (CLI --parallel 1): all shards of the
                        call share every launch (engine.TrainJob) and, when
                        torch.distributed is initialised, are spread over the ranks
                        (one process per GPU, no collective while training; trained
                        tables are exchanged once over RCCL).  The reference's per-epoch
                        test logs depend on the sequential order (they average the models
                        trained before the shard); they are rebuilt afterwards from
                        end-of-epoch snapshots (`param.epoch_logs`, default on while the
                        snapshots fit URE_SNAPSHOT_LIMIT_GB).
"""
import os

import numpy as np
import torch
from torch import nn

from .. import engine
from ..read import as_loader
from .scratch import PERM_THREADS, Scratch, prepare_shard, snapshot_limit
from .utils import MF, baseTest, padded_tables, seed_all




def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def assign_shards(sizes, world):
    """Longest-processing-time-first placement of shards on ranks."""
    load = [0] * world
    owner = [0] * len(sizes)
    for s in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[s] = r
        load[r] += sizes[s]
    return owner


def prepare_owned(ids, owner, rank, train_dlist, n_user, n_item, k, epochs, on_device=True):
    """Host part of a parallel call.  Every rank walks the WHOLE RNG stream in shard
    order (the draws are data independent: 4 fills + 4 seeds per epoch per shard,
    SURVEY.md 3.4) but DRAWS only for its own shards: the generator is moved past everything
    a shard of another rank would draw (ure_host_mt_advance), so shard i starts from exactly
    the state it would have in a sequential single-process run and no rank makes a normal it
    does not train on.  (Round 3 drew the full U0 of every foreign shard on every rank -- the
    reference's per-epoch logs need the earlier shards' full user tables -- ; those now come
    from their owners with the path's one exchange: exchange_tables(full_u=True).)"""
    from .. import rng
    prepared = {}
    # the HBM layouts of the owned shards -- which training needs first -- are built at once on a worker: ONE native call for
    # all of them, side by side on host threads into pinned staging, uploaded asynchronously, plus the work units of this
    # table width (a layout a loader already holds is returned as it is)
    layouts = None
    own_ids = [i for pos, i in enumerate(ids) if owner[pos] == rank]

    def start_layouts(allocate=True):
        from ..read import plan_shard_layouts
        finish = plan_shard_layouts([as_loader(train_dlist[i]) for i in own_ids], n_user, n_item, engine._device(), k)
        fut = rng.worker_pool().submit(finish)               # (the native builder starts now ...
        if allocate:
            finish.allocate()                                #  ... and the device allocations are made beside it)
        engine.mark('layouts planned')
        return fut, finish.allocate
    # the layouts first: their native builder runs on a thread of the library's own from this call on (engine.LayoutPlan), the worker below
    # only joins it and queues the copy
    allocate_layouts = None
    if on_device:
        layouts, allocate_layouts = start_layouts(allocate=False)
    streams = rng.shard_streams(len(ids), n_user, n_item, k, epochs, True, want_seeds=True) if on_device else None
    engine.mark('streams')
    if on_device and streams is None:
        allocate_layouts()
    if streams is not None:
        # every shard's start state by skip-ahead; then, in the order of what the job waits for longest: the owned shards' model inits
        # (a worker each, started at once: rng.start_inits), the layouts (one native call on a worker, the device allocations beside
        # it), and last the permutations' buffers and chunk workers (rng.draws_batch_async: seeds by skip-ahead, chunks round robin)
        starts, end, seeds = streams
        torch.set_rng_state(end)
        futures = {}
        specs, order = [], []
        # permutation chunks: 8 epochs each for a few shards (a 5-shard call: 8 -> 13.8 / 11.7 ms learn / unlearn, 17 -> 19.7 /
        # 14.2), larger ones when many shards make many chunk uploads (16 shards: 8 -> 21.6 / 23.2, 13 -> 20.9 / 20.1;
        # tools/ab_host.py medians): about 64 chunks per call.
        n_owned = sum(1 for pos in range(len(ids)) if owner[pos] == rank)
        chunk_epochs = max(8, -(-epochs * n_owned // 64))
        for pos, i in enumerate(ids):
            loader = as_loader(train_dlist[i])
            base = dict(start_state=starts[pos], n_user=n_user, n_item=n_item, k=k, epochs=epochs, with_total_test=True, seeds=seeds[pos])
            if owner[pos] == rank:
                specs.append(dict(base, n_rows=len(loader.dataset), shuffle=loader.shuffle, device=engine._device(),
                                  tags_batch=loader.batch_size if os.environ.get('URE_HOST_TAGS', '1') != '0' else 0,
                                  chunk_epochs=chunk_epochs))
                order.append(i)
        tasks = rng.start_inits(specs)
        engine.mark('inits started')
        # (the layouts' builder is running); the epochs' batch tags are put on their way (rng.draws_batch_async: made on the device from the
        # seeds, or by host workers); the layouts' device allocations -- which the builder's worker needs only when it is done -- come last
        # few workers, several shards each (rng.draws_batch_async): the expansion threads of a worker's native calls share
        # the rank's CPUs
        W = max(1, min(len(specs), max(2, rng.host_cpus() // 2)))
        for sp in specs:
            sp['threads'] = max(2, PERM_THREADS // W)
        allocate_layouts()                               # (the builder's worker queues the layouts' copy as soon as both are done)
        engine.mark('layouts allocated')
        futures = dict(zip(order, rng.draws_batch_async(specs, W, tasks=tasks)))
        engine.mark('draws submitted')
        shards = dict(zip(own_ids, layouts.result()))
        engine.mark('layouts')
        for pos, i in enumerate(ids):
            if i not in futures:
                continue
            prepared[i] = (shards[i], futures[i].init(), futures[i].perms())         # the permutations may still be arriving (chunks)
        engine.mark('inits')
        return prepared
    if layouts is not None:
        layouts.result()                # (tables too small to skip ahead in the stream: the draws below run one after the other)
    for pos, i in enumerate(ids):
        if owner[pos] == rank:
            if on_device:
                prepared[i] = prepare_shard(train_dlist[i], n_user, n_item, k, epochs, True, defer=True)
            else:   # host-only (tests): same draws, no HBM layout
                U0, V0 = rng.mf_init(n_user, n_item, k)
                seeds = rng.epoch_seeds(epochs, True)
                n = len(as_loader(train_dlist[i]).dataset)
                prepared[i] = (None, (U0, V0), rng.epoch_perms(seeds, n))
        else:
            rng.skip_model(n_user, n_item, k, epochs, True)
    for i, (shard, init, perms) in list(prepared.items()):      # permutations expanded in the background: collect
        if hasattr(perms, 'result'):
            prepared[i] = (shard, init, perms.result())
    return prepared


def exchange_plan(ids, owner, group_sizes, n_item, k, world):
    """Layout of the path's one collective.  A rank's segment holds, for each shard it owns (in `ids`
    order), that shard's OWN user rows [len(group_index[i]), k] followed by its item table
    [n_item, k]; segments are padded to the longest (all-gather-v as one all_gather_into_tensor).
    With full user tables (exchange_tables(full_u=True)) pass n_user for every group size.
    -> ({shard: (rank, offset in floats)}, floats per segment)."""
    fill = [0] * world
    where = {}
    for pos, i in enumerate(ids):
        r = owner[pos]
        where[i] = (r, fill[r])
        fill[r] += (group_sizes[pos] + n_item) * k
    return where, max(max(fill), 1)


def exchange_tables(models, ids, owner, rank, rows, n_item, k, device, dist, full_u=0):
    """After isolated training: ONE all-gather over RCCL/xGMI.  Every rank contributes, per shard it
    owns, the rows of U_i that belong to the shard's own users (sisa.py:55-56 reads nothing else of
    U_i) and V_i (utils.py:140-145 needs every model's item table); payloads are padded to the
    longest segment.  `rows[i]` = device int64 tensor of group_index[i].
    full_u = n_user: the WHOLE pre-merge user table of every shard travels instead of its own rows --
    what the reference's per-epoch logs (scratch.py:83-86 averages the earlier models as they are) and
    user_mat{id}.npy need of a foreign shard.  S x n_user x k floats in all: 3.9 MB at ml-1m / 5 shards,
    2.6 GB at BASELINE.json configs[3] (~30 ms on a ring of 8 over xGMI), against 28 x 20.7 M normals
    drawn per rank to rebuild them from the init stream (round 3).
    -> {shard: (U rows of its own users [len(rows[i]), k] | U [n_user, k], V [n_item, k])}, device tensors."""
    world = dist.get_world_size()
    sizes = [int(full_u) if full_u else int(rows[i].numel()) for i in ids]
    where, seg = exchange_plan(ids, owner, sizes, n_item, k, world)
    send = torch.zeros(seg, dtype=torch.float32, device=device)
    for pos, i in enumerate(ids):
        if owner[pos] == rank:
            U, V = models[i]
            o = where[i][1]
            nu = sizes[pos] * k
            send[o:o + nu] = (U if full_u else U.index_select(0, rows[i])).reshape(-1)
            send[o + nu:o + nu + n_item * k] = V.reshape(-1)
    recv = torch.empty(world * seg, dtype=torch.float32, device=device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(recv, send)
    else:       # rehearsal transports (gloo): staged through the host
        host = torch.empty(world * seg, dtype=torch.float32)
        dist.all_gather_into_tensor(host, send.cpu())
        recv.copy_(host)
    out = {}
    for pos, i in enumerate(ids):
        r, o = where[i]
        base = r * seg + o
        nu = sizes[pos] * k
        out[i] = (recv[base:base + nu].view(sizes[pos], k), recv[base + nu:base + nu + n_item * k].view(n_item, k))
    return out


def untouched_scale(lr, lr_decay, epochs, steps, lam, momentum, lr_step=50):
    """a_T of a row that only decays (no interaction in its shard) after the whole training:
    w_T = float32(a_T) * w_0 -- the same closed form ure_job_materialize applies on the owner."""
    lr_host = np.array([lr * (lr_decay ** (t // lr_step)) for t in range(epochs)], dtype=np.float32)
    a = engine.closed_form_scalars(lr_host, steps, float(np.float32(lam)), float(np.float32(momentum)))
    return np.float32(a[-1]) if len(a) else np.float32(1.0)


class Sisa(Scratch):
    def __init__(self, param={}, model_type='mf', n_group=5, group_index=[]):
        super(Sisa, self).__init__(param, model_type)
        self.n_group = n_group
        self.group_index = group_index
        self.parallel = bool(getattr(param, 'parallel', False))
        self.epoch_logs = bool(getattr(param, 'epoch_logs', True))   # parallel mode: rebuild the per-epoch test logs
        self.model_list = []

    def test(self, test_data, verbose, save_dir):
        dist = _dist()
        if dist is not None and self.parallel:
            # several ranks hold the same merged models: rank 0 tests, the three numbers travel (every rank ran this test in round 3)
            box = [baseTest(test_data, self.model_list, nn.MSELoss(reduction='sum'), self.device, verbose) if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            rmse, ndcg, hr = box[0]
        else:
            rmse, ndcg, hr = baseTest(test_data, self.model_list, nn.MSELoss(reduction='sum'), self.device, verbose)
        log = {'total_rmse': rmse,
               'total_ndcg': ndcg,
               'total_hr': hr}
        if len(save_dir) > 0:
            np.save(save_dir + '/log0', log)
        self.log0 = log

    # ------------------------------------------------------------------ helpers
    def _rows(self, i):
        return torch.as_tensor(np.asarray(self.group_index[i], dtype=np.int64))

    def _rows_dev(self, i):
        """group_index[i] as a device int64 tensor; all groups go up together, once per (group_index object, its group sizes, device):
        the reference reads self.group_index at every merge (sisa.py:55-56), so a reassigned list or another current device must not
        find the rows of the old one.  (Groups edited in place at equal length are not detected: call forget_rows().)"""
        dev = engine._device()
        key = (id(self.group_index), tuple(len(g) for g in self.group_index), str(dev))
        if getattr(self, '_group_rows_key', None) != key:
            self._group_rows = engine.upload_many([np.asarray(g, dtype=np.int64) for g in self.group_index], dev)
            self._group_rows_key = key
        return self._group_rows[i]

    def forget_rows(self):
        self._group_rows_key = None

    def _check_closed(self):
        """The job of the last request is destroyed on a worker (ure_job_destroy waits for the device); a failure there surfaces at the
        start of the next request or here."""
        fut, self._closing = getattr(self, '_closing', None), None
        if fut is not None:
            fut.result()

    def _merge(self, merged, ids):
        """sisa.py:52-58 / 107-113: merged[group_index[i]] = U_i[group_index[i]]."""
        for i in ids:
            src = self.model_list[i].user_mat.weight.detach().contiguous()
            engine.merge_rows(merged, src, self._rows_dev(i))
        for m in self.model_list:
            m.user_mat.weight = nn.Parameter(merged, requires_grad=False)

    def _train_parallel(self, ids, train_dlist, test_dlist, test_data, verbose, save_dir, unlearning=False):
        """Train the shards `ids` side by side (and across ranks).  With epoch_logs the tables of
        every epoch end are kept on the device and the reference's per-epoch group / total tests
        (scratch.py:83-97: mean of the models trained BEFORE this shard + this shard's model of
        that epoch) are computed afterwards, so log{id}.npy carries the same series as a
        sequential run."""
        seed_all(self.seed)
        self._check_closed()
        engine.mark('start')
        # The shards' worker threads (draws, permutation chunks, uploads) run short pieces of Python between native calls; with
        # CPython's default switch interval (5 ms) this thread can wait that long for the GIL every time it comes back from a
        # native call of its own -- 10 ms between two marks of a 16-shard call.  A short interval for the duration of the call.
        import sys
        switch = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
        from .. import rng
        try:
            with rng.torch_threads():
                return self._train_parallel_body(ids, train_dlist, test_dlist, test_data, verbose, save_dir, unlearning)
        finally:
            sys.setswitchinterval(switch)

    def _train_parallel_body(self, ids, train_dlist, test_dlist, test_data, verbose, save_dir, unlearning):
        dist = _dist()
        world = dist.get_world_size() if dist else 1
        rank = dist.get_rank() if dist else 0
        sizes = [len(as_loader(train_dlist[i]).dataset) for i in ids]
        owner = assign_shards(sizes, world)
        # full pre-merge user tables of other ranks' shards are needed for the reference's per-epoch logs
        # (scratch.py:83-86 averages the earlier models as they are) and for user_mat{id}.npy on rank 0
        want_full = bool(dist) and (self.epoch_logs or len(save_dir) > 0)          # (the same on every rank: it sizes the collective)
        prepared = prepare_owned(ids, owner, rank, train_dlist, self.n_user, self.n_item, self.k, self.epochs)
        mine = [i for pos, i in enumerate(ids) if owner[pos] == rank]
        snap_mode = 'compact' if engine.LAZY_ROWS else 'full'
        snap_bytes = engine.TrainJob.snapshot_bytes([prepared[i][0] for i in mine], self.epochs, self.k, snap_mode)
        keep_logs = self.epoch_logs and snap_bytes <= snapshot_limit()
        if self.epoch_logs and not keep_logs:
            import warnings
            warnings.warn(f'per-epoch test logs of this call are NaN: {snap_bytes / 2**30:.1f} GiB of end-of-epoch snapshots '
                          f'exceed URE_SNAPSHOT_LIMIT_GB ({snapshot_limit() / 2**30:.3g}); train_loss and log0 are complete')
        models, job = {}, None
        if mine:
            batch = as_loader(train_dlist[mine[0]]).batch_size
            job = engine.TrainJob([prepared[i][0] for i in mine], [prepared[i][1] for i in mine],
                                  [prepared[i][2] for i in mine], self.k, batch, self.epochs, self.lr, self.lam,
                                  self.momentum, self.lr_decay, snapshots=snap_mode if keep_logs else False,
                                  final_only=True)       # (the tables are read once, after the last epoch)
            from .. import rng
            engine.mark('job_created')
            job.run()
            self._rows_dev(mine[0])             # (the merge's row lists go up while the device works through the launches)
            engine.mark(f'launched (waited {getattr(job, "chunk_wait_s", 0.0) * 1e3:.2f} ms for permutation chunks)')
            for i in mine:
                rng.release(prepared[i][2])                 # uploaded: host buffers go back to the pool

        # ---- the per-epoch test series, shard after shard in the reference's order (one dict for all shards)
        out, queued = {}, {}
        # the results of all series of the call in one device tensor: read with ONE copy below
        # (and the per-epoch training losses behind them: ONE device tensor, ONE copy to the host for the whole call)
        n_res = len(mine) * 6 * self.epochs if keep_logs else 0
        flat = torch.zeros(n_res + len(mine) * self.epochs, dtype=torch.float64, device=engine._device()) if mine else None
        res_all = flat[:n_res].view(len(mine), 2, self.epochs, 3) if (mine and keep_logs) else None
        if mine:
            job.epoch_sse_queue(out=flat[n_res:].view(len(mine), self.epochs))         # (one launch, right behind the last step)

        def queue_series(fixed_of):
            """Every shard's two test series (scratch.py:83-97) go into the queue, all on the current stream (the shards' series on four
            streams were measured: the events cost more than such short chains overlap -- NOTES r4 6); fixed_of(j) = the padded (U, V) of
            shard j's final model.
            Member e of shard i's series is the mean over an ensemble of FIXED models -- learn: the shards trained before i
            (scratch.py:83-86); unlearn: every shard's model, retrained if it came before i in this call, else the old one (sisa.py:89) --
            and i's own model after epoch e.  The fixed models' share is the same sum for every epoch and nearly the same for every
            shard: each distinct model is scored ONCE on the total test set (engine.ScoreCache) and a shard's base is the sum of
            those vectors in the ensemble's order -- the additions ure_score makes, so every number of every series is unchanged --
            instead of scoring i models again for shard i: S instead of S^2 / 2 table passes (S^2 when unlearning), which at
            BASELINE.json configs[3] (32 shards, 2.5 M test pairs, d = 128) was 1.3 TB of gathers per learn."""
            if not keep_logs:
                return
            total_ev = as_loader(test_data).eval_set()
            cache = engine.ScoreCache(total_ev, job.d)
            old = {}

            def old_tables(j):
                if j not in old:
                    old[j] = padded_tables(self.model_list[j])[:2]
                return old[j]
            for i in mine:
                test_ev = as_loader(test_dlist[i]).eval_set()
                if unlearning:                              # sisa.py:89: the shards before i in the retraining order are already replaced
                    replaced = set(ids[:ids.index(i)])
                    before = [(('new', j), fixed_of, j) if j in replaced else (('old', j), old_tables, j) for j in range(len(self.model_list))]
                else:
                    before = [(('new', j), fixed_of, j) for j in ids[:ids.index(i)]]
                pos = mine.index(i)
                res = res_all[pos]
                if test_ev.subset_of(total_ev) is not None and cache.fits(len({key for key, _, _ in before}) + len(cache.vec)):
                    fixed = cache.base(before)
                else:                                       # (the shard's own set is not the total set's rows of its users: two series, models scored per series)
                    fixed = [get(j) for _, get, j in before]
                job.evaluate_series_pair(pos, test_ev, total_ev, fixed, res[0], res[1])
                queued[i] = pos

        if mine and not dist:
            # one process: every final model is in the job's own tables -- the series are queued from THOSE, right behind the last
            # launch, and the copies the caller keeps are made while the device works through them (made first, the device sat idle for
            # 0.7 ms of a 5-shard request between its last step and its first series)
            queue_series(lambda j: job.padded_tables(mine.index(j)))
            engine.mark('series_queued')
        if mine:
            for pos, i in enumerate(mine):
                U, V = job.tables(pos)
                models[i] = (U.clone().contiguous(), V.clone().contiguous())
        if dist:
            # the only exchange of the path (sisa.py:52-58): own user rows + item table of every shard, one all-gather
            dev = engine._device()
            rows = {i: self._rows_dev(i) for i in ids}
            got = exchange_tables(models, ids, owner, rank, rows, self.n_item, self.k, dev, dist, full_u=self.n_user if want_full else 0)
            for pos, i in enumerate(ids):
                if owner[pos] == rank:
                    continue
                Ur, V = got[i]
                if want_full:      # the owner's whole pre-merge table (rows of other shards' users included: they only decayed there)
                    U = Ur.clone()
                else:              # only the shard's own rows are ever read again (merge + final test)
                    U = torch.zeros(self.n_user, self.k, dtype=torch.float32, device=dev)
                    U.index_copy_(0, rows[i], Ur)
                models[i] = (U, V.clone())
        out.update({i: MF.from_tables(*models[i]) for i in ids})
        engine.mark('models')
        logs = {}
        if dist:
            queue_series(lambda j: padded_tables(out[j])[:2])
            engine.mark('series_queued')
        # two copies for the whole call (a copy per shard and kind was a synchronisation each; tools/ab_host.py medians of 6:
        # 13.5 / 12.2 -> 12.1 / 12.2 ms learn / unlearn at 5 shards, within the noise at 16)
        if mine:
            from .. import rng
            stage = rng.SMALL.take((flat.numel(),), torch.float64)
            stage.copy_(flat, non_blocking=True)
            if stage.is_pinned():
                torch.cuda.current_stream(engine._device()).synchronize()
            host = stage.numpy().copy()
            rng.SMALL.give(stage)
            rng.device_tags_check([prepared[i][2] for i in mine])       # (the device is idle here: one tiny read)
            res_host = host[:n_res].reshape(len(mine), 2, self.epochs, 3) if keep_logs else None
            sse_host = host[n_res:].reshape(len(mine), self.epochs)
        for pos, i in enumerate(mine):
            entry = {'train_loss': np.sqrt(sse_host[pos] / prepared[i][0].N).tolist()}
            if keep_logs:
                for c, key in enumerate(('test_rmse', 'test_ndcg', 'test_hr')):
                    entry[key] = res_host[pos, 0, :, c].tolist()
                for c, key in enumerate(('total_rmse', 'total_ndcg', 'total_hr')):
                    entry[key] = res_host[pos, 1, :, c].tolist()
            else:
                nan = [float('nan')] * self.epochs
                for key in ('test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr'):
                    entry[key] = list(nan)
            logs[i] = entry
        engine.mark('logs_read')
        if job is not None:
            # the job's device memory is returned by a worker: ure_job_destroy waits for the device and takes 0.2-0.4 ms, and
            # nothing below needs it (the models are copies, the logs are on the host)
            from .. import rng
            self._closing = rng.worker_pool().submit(job.close)
        engine.mark('job_closed')
        if dist:
            gathered = [None] * world
            dist.all_gather_object(gathered, logs)
            logs = {k: v for part in gathered for k, v in part.items()}
        for i in ids:
            if rank == 0 and verbose in (1, 2):
                # the lines the sequential path prints while it trains (scratch.py:99-118), shard after shard in the reference's order
                from .scratch import print_epoch
                print('Using device:', self.device)
                e = logs[i]
                for t in range(self.epochs):
                    print_epoch(verbose, t, self.epochs, e['train_loss'][t], (e['test_rmse'][t], e['test_ndcg'][t], e['test_hr'][t]),
                                (e['total_rmse'][t], e['total_ndcg'][t], e['total_hr'][t]), True, '00:00:00')
            for key, vals in logs[i].items():
                self.log[key] += vals
            self.log['time'] += ['00:00:00'] * self.epochs
            if rank == 0:
                self.save(out[i], save_dir, i + 1)
        engine.mark('logs_appended')
        return out

    # ------------------------------------------------------------------ learn
    def learn(self, train_dlist, test_dlist, test_data, verbose, save_dir):
        '''
        train_dlist:   list of dataloader[n_group]
        '''
        assert len(train_dlist) == self.n_group
        assert len(test_dlist) == self.n_group

        if self.parallel:
            trained = self._train_parallel(list(range(self.n_group)), train_dlist, test_dlist, test_data, verbose, save_dir)
            self.model_list = [trained[i] for i in range(self.n_group)]
        else:
            for i in range(self.n_group):
                given_model = ''
                model = super(Sisa, self).train(train_dlist[i], test_dlist[i], test_data, verbose, save_dir, i + 1, given_model)
                self.model_list.append(model)

        # merge user mat (sisa.py:52-58)
        merged = torch.zeros_like(self.model_list[0].user_mat.weight.detach()).contiguous()
        self._merge(merged, range(self.n_group))
        engine.mark('merged')

        # total test
        self.test(test_data, verbose, save_dir)
        engine.mark('tested')
        return self.model_list

    # ------------------------------------------------------------------ unlearn
    def unlearn(self, model_list, train_dlist, test_dlist, test_data, del_user, verbose, save_dir):
        '''
        train_dlist:   list of dataloader[n_group]
        '''
        self.model_list = model_list

        assert len(train_dlist) == self.n_group
        assert len(test_dlist) == self.n_group

        # find deletion (sisa.py:76-81: the first group that lists the user)
        first_group = np.full(max(self.n_user, 1 + max((int(max(g)) for g in self.group_index if len(g)), default=0)), -1, dtype=np.int64)
        for i in reversed(range(self.n_group)):
            first_group[np.asarray(self.group_index[i], dtype=np.int64)] = i
        users = np.asarray(list(del_user), dtype=np.int64).reshape(-1)
        users = users[(users >= 0) & (users < len(first_group))]
        retrain_gid = set(int(g) for g in np.unique(first_group[users]) if g >= 0)

        model_before_unlearn = model_list[0]
        merged = model_before_unlearn.user_mat.weight.detach().clone().contiguous()

        if self.parallel:
            ids = list(retrain_gid)
            trained = self._train_parallel(ids, train_dlist, test_dlist, test_data, verbose, save_dir, unlearning=True) if ids else {}
            for i in ids:
                self.model_list[i] = trained[i]
        else:
            for i in retrain_gid:
                given_model = ''
                model = super(Sisa, self).train(train_dlist[i], test_dlist[i], test_data, verbose, save_dir, i + 1, given_model)
                self.model_list[i] = model

        # merge user mat (sisa.py:107-113): only the retrained shards' rows change
        self._merge(merged, list(retrain_gid))

        # total test
        self.test(test_data, verbose, save_dir)
        self.retrained = sorted(retrain_gid)
        return self.model_list
