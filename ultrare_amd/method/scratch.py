"""Scratch.train with the reference's signature (method/scratch.py:11-148), driving
the HBM-resident engine: the shard is uploaded once, every epoch is `steps` fused
kernel launches, and the two per-epoch tests run on the device.

RNG: the reference draws, per Scratch.train call, four normal fills (model init) and
per epoch three or four int64 seeds from torch's global CPU generator.
The draws are data independent, so they are all taken up front in the same order;
the epoch permutations are then expanded from their own generators.
"""
import os
import time

import numpy as np
import torch
from torch import nn

from .. import engine, rng
from ..read import as_loader
from .utils import MF, padded_tables, seed_all

PERM_THREADS = rng.perm_threads()
# end-of-epoch snapshots kept on the device for the per-epoch test series (per job).  They hold the rows with interactions in
# the shard only (engine: 'compact'): 5 GiB for 5 epochs of BASELINE.json configs[3] (32 shards, d = 128) on a 288 GB card.
def snapshot_limit():
    return int(float(os.environ.get('URE_SNAPSHOT_LIMIT_GB', '64')) * 2 ** 30)


def _is_empty(x):
    return isinstance(x, (list, tuple)) and len(x) == 0


def prepare_shard(train_loader, n_user, n_item, k, epochs, has_total, given_model='', defer=False):
    """Host part of one Scratch.train call: consume the RNG stream exactly like the
    reference (model init, then per-epoch seeds) and expand the permutations.  defer=True returns
    the permutations as a future (rng.epoch_perms_async): the expansion then runs beside the
    caller's next draws."""
    loader = as_loader(train_loader)
    if _is_empty(given_model) or given_model == '':
        # the stream of MF(...): 2 discarded + 2 kept fills (utils.py:31-40); big tables are made on the device
        U0, V0 = rng.mf_init(n_user, n_item, k, device=engine._device())
    else:
        U0 = given_model.user_mat.weight.detach().float().cpu()
        V0 = given_model.item_mat.weight.detach().float().cpu()
    seeds = rng.epoch_seeds(epochs, has_total)
    n = len(loader.dataset)
    # the permutations go to the device as batch tags (rng.epoch_tags; struct ure_shard: file_tags): half the bytes, and the device
    # does not partition them (URE_HOST_TAGS=0: as permutations)
    tags = loader.batch_size if os.environ.get('URE_HOST_TAGS', '1') != '0' and 0 < -(-n // max(loader.batch_size, 1)) <= 65535 else 0
    on_dev = None
    if loader.shuffle and tags and n < (2 ** 32 - 1) // 20:
        # ... and they are MADE on the device (rng.epoch_tags_device -> csrc/perm_chain.hip / perm_tags.hip): no sequential Fisher-Yates per
        # epoch on the host in front of the device's epochs (22.5 M rows per epoch for config.py:182-188's run at the 25 M shape)
        on_dev = rng.epoch_tags_device(seeds, n, tags, engine._device())
    if on_dev is not None:
        perms = on_dev
    elif loader.shuffle and defer:
        perms = rng.epoch_perms_async(seeds, n, threads=PERM_THREADS, pooled=True, device=engine._device(), tags_batch=tags)
    elif loader.shuffle and tags and n < (2 ** 32 - 1) // 20:
        perms = rng.epoch_tags(seeds, n, tags, threads=PERM_THREADS)
    elif loader.shuffle:
        perms = rng.epoch_perms(seeds, n, threads=PERM_THREADS, pooled=True)
    else:
        perms = torch.arange(n, dtype=torch.int32).repeat(epochs, 1)
    return loader.shard_data(n_user, n_item), (U0, V0), perms


def print_epoch(verbose, t, epochs, train_loss, test, total, has_total, epoch_time):
    """The per-epoch lines of scratch.py:99-118 (verbose 1 and 2).  test / total = (rmse, ndcg, hr)."""
    if verbose == 2:
        print(f'Epoch: [{t+1:>3d}/{epochs:>3d}] --------------------')
        print(f'Test - RMSE: {test[0]:>.4f}, NDCG: {test[1]:>.3f}, HR: {test[2]:>.3f}')
        print('Time:', epoch_time)
    elif verbose == 1:
        msg = (f'Epoch: [{t+1:>2d}/{epochs:>2d}]' + f' train loss: {train_loss:>.9f},' +
               f' train RMSE: {train_loss:>.4f},' + f' test RMSE: {test[0]:>.4f},')
        if has_total:
            msg += f' total RMSE: {total[0]:>.4f},'
        print(msg + ' time:', epoch_time)


class Scratch(object):
    def __init__(self, param, model_type):
        # model param
        self.n_user = param.n_user
        self.n_item = param.n_item
        self.k = param.k
        self.lam = param.lam
        self.model_type = model_type

        # training param
        self.seed = param.seed
        self.lr = param.lr
        self.lr_decay = param.lr_decay
        self.momentum = param.momentum
        self.epochs = param.epochs
        self.batch = getattr(param, 'batch', 30000)
        self.device = 'cuda'
        # InsParam never sets dis_type / attr; 'nor' is the only branch that
        # works with MF (utils.py:64-65)
        self.dis_type = getattr(param, 'dis_type', 'nor')
        if self.dis_type != 'nor':
            raise NotImplementedError("only dis_type='nor' exists for the MF model (utils.py:66-81 needs a non-MF model)")
        self.attr = []

        # log (one dict per object, appended by every shard it trains)
        self.log = {'train_loss': [],
                    'test_rmse': [],
                    'test_ndcg': [],
                    'test_hr': [],
                    'total_rmse': [],
                    'total_ndcg': [],
                    'total_hr': [],
                    'time': []}

        if self.model_type != 'mf':
            raise NotImplementedError("model_type is always 'mf' on the published path (config.py:185-199)")
        self.loss_fn = nn.MSELoss(reduction='sum')
        self.is_rmse = True

    def _models_before(self):
        return list(getattr(self, 'model_list', [])) if self.__class__.__name__ == 'Sisa' else []

    def train(self, train_data, test_data, test_total=[], verbose=1, save_dir='', id=0, given_model=''):
        with rng.torch_threads():          # (torch's intra-op pool capped for the duration of the request, restored afterwards)
            return self._train(train_data, test_data, test_total, verbose, save_dir, id, given_model)

    def _train(self, train_data, test_data, test_total, verbose, save_dir, id, given_model):
        print('Using device:', self.device)
        seed_all(self.seed)                                 # scratch.py:54
        has_total = not _is_empty(test_total)
        shard, init, perms = prepare_shard(train_data, self.n_user, self.n_item, self.k, self.epochs, has_total, given_model)
        batch = as_loader(train_data).batch_size
        # verbose 0: nothing is printed per epoch, so the epochs run back to back, every epoch end is kept
        # on the device (snapshots) and the two test series of scratch.py:83-97 are computed afterwards
        # in four launches each (ure_eval_series); otherwise each epoch synchronises to print
        queued = verbose == 0
        snap_mode = 'compact' if engine.LAZY_ROWS else 'full'
        series = queued and engine.TrainJob.snapshot_bytes([shard], self.epochs, self.k, snap_mode) <= snapshot_limit()
        job = engine.TrainJob([shard], [init], [perms], self.k, batch, self.epochs, self.lr, self.lam, self.momentum,
                              self.lr_decay, snapshots=snap_mode if series else False, final_only=series, epoch_reads=True)       # (tables are read at epoch ends only)
        rng.release(perms)                                  # uploaded: the host buffer goes back to the pool
        test_ev = as_loader(test_data).eval_set()
        total_ev = as_loader(test_total).eval_set() if has_total else None
        before = [padded_tables(m)[:2] for m in self._models_before()]
        n_train = shard.N

        res = torch.zeros(self.epochs, 2, 3, dtype=torch.float64, device=shard.device) if queued else None
        times = []
        if series:
            sets = (test_ev, total_ev if has_total else test_ev)
            job.run()
            res = torch.zeros(2, self.epochs, 3, dtype=torch.float64, device=shard.device)
            if has_total:
                job.evaluate_series_pair(0, sets[0], sets[1], before, res[0], res[1])
            else:
                for which, ev in enumerate(sets):
                    job.evaluate_series(0, ev, before, res[which])
            res = res.transpose(0, 1).contiguous()
            times = ['00:00:00'] * self.epochs
        for t in range(0 if not series else self.epochs, self.epochs):
            epoch_start = time.time()
            job.run_epochs(1)                               # baseTrain (utils.py:46-111) + scheduler.step()
            models = before + [job.padded_tables(0)]        # scratch.py:83-86
            if queued:
                test_ev.evaluate(models, job.d, out=res[t, 0])
                (total_ev if has_total else test_ev).evaluate(models, job.d, out=res[t, 1])
                times.append(time.strftime('%H:%M:%S', time.gmtime(time.time() - epoch_start)))
                continue
            test_rmse, test_ndcg, test_hr = test_ev.evaluate(models, job.d)
            if has_total:
                total_rmse, total_ndcg, total_hr = total_ev.evaluate(models, job.d)
            else:
                total_rmse, total_ndcg, total_hr = test_rmse, test_ndcg, test_hr
            train_loss = float(np.sqrt(job.epoch_sse(0)[t] / n_train))
            epoch_time = time.strftime('%H:%M:%S', time.gmtime(time.time() - epoch_start))
            print_epoch(verbose, t, self.epochs, train_loss, (test_rmse, test_ndcg, test_hr), (total_rmse, total_ndcg, total_hr), has_total, epoch_time)

            self.log['train_loss'].append(train_loss)
            self.log['test_rmse'].append(test_rmse)
            self.log['test_ndcg'].append(test_ndcg)
            self.log['test_hr'].append(test_hr)
            self.log['time'].append(epoch_time)
            if has_total:
                self.log['total_rmse'].append(total_rmse)
                self.log['total_ndcg'].append(total_ndcg)
                self.log['total_hr'].append(total_hr)

        if queued:
            res = res.cpu().numpy()
            self.log['train_loss'] += [float(x) for x in np.sqrt(job.epoch_sse(0) / n_train)]
            for c, key in enumerate(('test_rmse', 'test_ndcg', 'test_hr')):
                self.log[key] += [float(x) for x in res[:, 0, c]]
            self.log['time'] += times
            if has_total:
                for c, key in enumerate(('total_rmse', 'total_ndcg', 'total_hr')):
                    self.log[key] += [float(x) for x in res[:, 1, c]]

        U, V = job.tables(0)
        model = MF.from_tables(U.clone().contiguous(), V.clone().contiguous())
        job.close()
        self.save(model, save_dir, id)
        return model

    def save(self, model, save_dir, id):
        """scratch.py:131-144: model{id}.pth, user_mat{id}.npy, item_mat{id}.npy, log{id}.npy."""
        from .utils import dist_rank
        if len(save_dir) > 0 and dist_rank()[0] == 0:       # several ranks running the same call: one set of files
            torch.save(model.state_dict(), save_dir + '/model' + str(id) + '.pth')
            np.save(save_dir + '/user_mat' + str(id), model.user_mat.weight.detach().cpu().numpy())
            np.save(save_dir + '/item_mat' + str(id), model.item_mat.weight.detach().cpu().numpy())
            np.save(save_dir + '/log' + str(id), self.log)
