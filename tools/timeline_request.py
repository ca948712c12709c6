#!/usr/bin/env python3
"""Host marks (URE_HOST_TRACE: calling thread and workers, no profiler) of ONE new Sisa.learn and ONE new Sisa.unlearn request at either
synthetic size, after a warm-up request -- where a request's wall time goes.

    python tools/timeline_request.py [--workload ml25m] [--shards 32] [--k 128] [--epochs 5] [--reps 2]
"""
import os
os.environ.setdefault('URE_HOST_TRACE', '1')
import argparse, copy, json, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import synth, engine
from ultrare_amd.method.sisa import Sisa
from ultrare_amd.read import RatingData, loadData

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='ml25m')
ap.add_argument('--shards', type=int, default=32)
ap.add_argument('--k', type=int, default=128)
ap.add_argument('--epochs', type=int, default=5)
ap.add_argument('--reps', type=int, default=2)
a = ap.parse_args()
data = synth.make_dataset(**(synth.ML1M if a.workload == 'ml1m' else synth.ML25M))
S = a.shards
shard_of, groups = synth.uniform_shards(data['n_user'], S)


class P:
    lam, seed, batch, lr, lr_decay, momentum, parallel = 0.1, 42, 30000, 0.001, 0.95, 0.9, True
    n_user, n_item = data['n_user'], data['n_item']


P.k, P.epochs = a.k, a.epochs
loaders = lambda triple, shuffle: [loadData(RatingData(np.vstack(p)), P.batch, 24, shuffle) for p in synth.split_shards(triple, shard_of, S)]
ted = loaders(data['test'], False)
parts_te = synth.split_shards(data['test'], shard_of, S)
tot = loadData(RatingData(np.vstack([np.concatenate([p[c] for p in parts_te]) for c in range(3)])), P.batch, 24, False)


def marks(t0):
    tr = sorted(engine.HOST_TRACE, key=lambda x: x[1])
    engine.HOST_TRACE.clear()
    return [(l, round((t - t0) * 1e3, 2)) for l, t in tr]


out = {}
for rep in range(a.reps):
    del_user = np.random.RandomState(1 + rep).choice(P.n_user, int(0.02 * P.n_user), replace=False)
    keep = ~np.isin(data['train'][0], del_user)
    trd, trd_del = loaders(data['train'], True), loaders(tuple(x[keep] for x in data['train']), True)
    s = Sisa(P, 'mf', S, groups)
    torch.manual_seed(42)
    torch.cuda.synchronize()
    engine.HOST_TRACE.clear()
    t0 = time.perf_counter()
    ml = s.learn(trd, ted, tot, 0, '')
    torch.cuda.synchronize()
    s._check_closed()
    out['learn_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
    out['learn_marks'] = marks(t0)
    s2 = Sisa(P, 'mf', S, groups)
    snap = [copy.deepcopy(m) for m in ml]
    torch.manual_seed(42)
    torch.cuda.synchronize()
    engine.HOST_TRACE.clear()
    t0 = time.perf_counter()
    s2.unlearn(snap, trd_del, ted, tot, del_user.tolist(), 0, '')
    torch.cuda.synchronize()
    s2._check_closed()
    out['unlearn_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
    out['unlearn_marks'] = marks(t0)
print(json.dumps(out, indent=0))
