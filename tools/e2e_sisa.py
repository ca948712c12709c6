#!/usr/bin/env python3
"""End-to-end wall time of the SISA path on synthetic ml-1m: Sisa.learn then Sisa.unlearn
after a 2 % random user deletion (BASELINE.json metric, second half), through the
reference's operator surface.  Prints one JSON object.

    python tools/e2e_sisa.py [--shards 5] [--k 32] [--epochs 50] [--parallel 1]
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(shards=5, k=32, epochs=50, parallel=1, delper=2.0, data=None, reps=3, workload='ml1m'):
    """Wall time of Sisa.learn and Sisa.unlearn (per-epoch evals, merge and final test included) on the synthetic set: the
    MEDIAN of the repetitions after the first (which warms the allocator and the pools), every sample listed beside it
    (`learn_s_all`, `unlearn_s_all`: a single timing on a shared host is off by up to 50 %).  Every repetition is a NEW request: its own deletion set (a different 2 %
    of the users) and freshly made train loaders, so the HBM layouts of the shards it trains are built and uploaded
    INSIDE the timed calls (`layouts_built` counts them); what the earlier repetitions leave behind is a warm device
    allocator, the pinned permutation pool and the test sets (a deletion does not change them: config.py:139-172 reads
    the test files without the deletion list)."""
    a = argparse.Namespace(shards=shards, k=k, epochs=epochs, parallel=parallel, delper=delper)
    from ultrare_amd import engine, synth
    from ultrare_amd.method.sisa import Sisa
    from ultrare_amd.read import RatingData, loadData

    data = data or synth.make_dataset(**(synth.ML1M if workload == 'ml1m' else synth.ML25M))
    n_user, n_item = data['n_user'], data['n_item']
    shard_of, groups = synth.uniform_shards(n_user, a.shards)

    class P:
        k, lam, seed, batch, lr, lr_decay, momentum, epochs = a.k, 0.1, 42, 30000, 0.001, 0.95, 0.9, a.epochs
        parallel = bool(a.parallel)
    P.n_user, P.n_item = n_user, n_item

    def loaders(triple, shuffle):
        return [loadData(RatingData(np.vstack(p)), P.batch, 24, shuffle) for p in synth.split_shards(triple, shard_of, a.shards)]

    ted = loaders(data['test'], False)
    tot_arr = [np.concatenate([p[c] for p in synth.split_shards(data['test'], shard_of, a.shards)]) for c in range(3)]
    tot = loadData(RatingData(np.vstack(tot_arr)), P.batch, 24, False)
    torch.cuda.synchronize()

    out = {'shards': a.shards, 'k': a.k, 'epochs': a.epochs, 'parallel': bool(a.parallel),
           'train_rows': int(len(data['train'][0]))}
    t_learns, t_unlearns = [], []
    for rep in range(reps):       # earlier repetitions warm the allocator and the pinned pool; no layout survives them
        del_user = np.random.RandomState(1 + rep).choice(n_user, int(a.delper / 100 * n_user), replace=False)
        keep = ~np.isin(data['train'][0], del_user)
        trd = loaders(data['train'], True)
        trd_del = loaders(tuple(x[keep] for x in data['train']), True)
        sisa = Sisa(P, 'mf', a.shards, groups)
        torch.manual_seed(42)
        built0 = engine.ShardData.built
        t0 = time.perf_counter()
        ml = sisa.learn(trd, ted, tot, 0, '')
        torch.cuda.synchronize()
        t_learn = time.perf_counter() - t0
        built_learn = engine.ShardData.built - built0
        s2 = Sisa(P, 'mf', a.shards, groups)
        snap = [copy.deepcopy(m) for m in ml]
        torch.manual_seed(42)
        built0 = engine.ShardData.built
        t0 = time.perf_counter()
        s2.unlearn(snap, trd_del, ted, tot, del_user.tolist(), 0, '')
        torch.cuda.synchronize()
        t_unlearn = time.perf_counter() - t0
        built_unlearn = engine.ShardData.built - built0
        if rep > 0 or reps == 1:
            t_learns.append(t_learn)
            t_unlearns.append(t_unlearn)
    t_learn, t_unlearn = float(np.median(t_learns)), float(np.median(t_unlearns))
    n_learn = len(data['train'][0]) * a.epochs
    n_un = int(keep.sum()) * a.epochs if len(s2.retrained) == a.shards else None
    nan_shards = int(sum(1 for m in s2.model_list if not bool(torch.isfinite(m.item_mat.weight).all())))
    series = {k: np.asarray(v, dtype=np.float64) for k, v in sisa.log.items() if k != 'time'}
    epoch_logs = {'entries_per_series': int(len(series['total_rmse'])), 'finite_fraction': {k: round(float(np.isfinite(v).mean()), 4) for k, v in series.items()},
                  'first_epoch': {k: float(v[0]) for k, v in series.items() if len(v)}, 'last_epoch': {k: float(v[-1]) for k, v in series.items() if len(v)}}
    out.update(learn_s=round(t_learn, 4), unlearn_s=round(t_unlearn, 4), learn_s_all=[round(t, 4) for t in t_learns], unlearn_s_all=[round(t, 4) for t in t_unlearns],
               timed='median of the repetitions after the first; every one a new request', retrained_shards=len(s2.retrained), deleted_users=int(len(del_user)),
               deletion_set=f'RandomState({reps}).choice: a different 2 % in every repetition',
               layouts_built={'learn': built_learn, 'unlearn': built_unlearn},
               learn_interactions_per_s=round(n_learn / t_learn, 1), log0=sisa.log0, unlearn_log0=s2.log0,
               unlearn_interactions=n_un, nan_shards=nan_shards, epoch_logs_learn=epoch_logs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shards', type=int, default=5)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=50)
    ap.add_argument('--parallel', type=int, default=1)
    ap.add_argument('--delper', type=float, default=2.0)
    ap.add_argument('--workload', choices=['ml1m', 'ml25m'], default='ml1m')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(measure(a.shards, a.k, a.epochs, a.parallel, a.delper, reps=a.reps, workload=a.workload)))


if __name__ == '__main__':
    main()
