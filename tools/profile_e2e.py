#!/usr/bin/env python3
import os
os.environ.setdefault('URE_HOST_TRACE', '1')
"""cProfile of one shard-parallel Sisa.learn at ml-1m size (where does the host time go?)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import synth
from ultrare_amd.method.sisa import Sisa
from ultrare_amd.read import RatingData, loadData

data = synth.make_dataset(**synth.ML1M)
S = 5
shard_of, groups = synth.uniform_shards(data['n_user'], S)


class P:
    k, lam, seed, batch, lr, lr_decay, momentum, epochs, parallel = 32, 0.1, 42, 30000, 0.001, 0.95, 0.9, 50, True
    n_user, n_item = data['n_user'], data['n_item']


parts_tr = synth.split_shards(data['train'], shard_of, S)
parts_te = synth.split_shards(data['test'], shard_of, S)
ted = [loadData(RatingData(np.vstack(p)), P.batch, 24, False) for p in parts_te]
tot = loadData(RatingData(np.vstack([np.concatenate([p[c] for p in parts_te]) for c in range(3)])), P.batch, 24, False)
for rep in range(3):
    # a new request: freshly made train loaders, so the HBM layouts are built and uploaded inside the call (as bench.py times it)
    trd = [loadData(RatingData(np.vstack(p)), P.batch, 24, True) for p in parts_tr]
    s = Sisa(P, 'mf', S, groups)
    torch.manual_seed(42)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    s.learn(trd, ted, tot, 0, '')
    torch.cuda.synchronize()
    pr.disable()
    print('learn', round((time.perf_counter() - t0) * 1e3, 1), 'ms')
    from ultrare_amd import engine
    if engine.HOST_TRACE:
        tr = list(engine.HOST_TRACE)
        engine.HOST_TRACE.clear()
        print('  host timeline (ms since the call):', ', '.join(f'{lab} {round((t - t0) * 1e3, 2)}' for lab, t in tr))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
