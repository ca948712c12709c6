#!/usr/bin/env python3
"""Host timeline (URE_HOST_TRACE marks, worker threads included, no profiler) of new Sisa.learn / unlearn requests at ml-1m size.

    python tools/host_timeline.py [--reps 6] [--shards 5] [--k 32] [--ab VAR=a,b]

--ab: the environment variable VAR (one the package reads at call time) alternates between its values from request to request of ONE
process -- same box, same warm-up, interleaved --; per value the medians of the wall time and of its three parts are printed:
head = until the job exists, train = until the last launch is queued, tail = until the final test is back.
"""
import os
os.environ.setdefault('URE_HOST_TRACE', '1')
import sys, time, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import synth, engine
from ultrare_amd.method.sisa import Sisa
from ultrare_amd.read import RatingData, loadData

args = sys.argv[1:]
opt = lambda n, d: type(d)(args[args.index(n) + 1]) if n in args else d
reps, S, k = opt('--reps', 6), opt('--shards', 5), opt('--k', 32)
data = synth.make_dataset(**synth.ML1M)
shard_of, groups = synth.uniform_shards(data['n_user'], S)


class P:
    lam, seed, batch, lr, lr_decay, momentum, epochs, parallel = 0.1, 42, 30000, 0.001, 0.95, 0.9, 50, True
    n_user, n_item = data['n_user'], data['n_item']


P.k = k
parts_tr = synth.split_shards(data['train'], shard_of, S)
parts_te = synth.split_shards(data['test'], shard_of, S)
ted = [loadData(RatingData(np.vstack(p)), P.batch, 24, False) for p in parts_te]
tot = loadData(RatingData(np.vstack([np.concatenate([p[c] for p in parts_te]) for c in range(3)])), P.batch, 24, False)
ab = opt('--ab', '')
ab_var, ab_vals = (ab.split('=')[0], ab.split('=')[1].split(',')) if ab else (None, [None])
parts = {v: [] for v in ab_vals}
steps = {}
walls = []
for rep in range(reps):
    val = ab_vals[rep % len(ab_vals)]
    if ab_var:
        os.environ[ab_var] = val
    trd = [loadData(RatingData(np.vstack(p)), P.batch, 24, True) for p in parts_tr]
    s = Sisa(P, 'mf', S, groups)
    torch.manual_seed(42)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.learn(trd, ted, tot, 0, '')
    s._check_closed()
    torch.cuda.synchronize()
    w = (time.perf_counter() - t0) * 1e3
    walls.append(w)
    tr = sorted(engine.HOST_TRACE, key=lambda x: x[1])
    engine.HOST_TRACE.clear()
    at = {l.split(' (')[0]: (t - t0) * 1e3 for l, t in tr if not l.startswith('w:')}
    if rep >= len(ab_vals):
        main_marks = [(l.split(' (')[0], (t - t0) * 1e3) for l, t in tr if not l.startswith('w:')]
        steps.setdefault(val, []).append({b[0]: b[1] - a[1] for a, b in zip(main_marks[:-1], main_marks[1:])})
        parts[val].append((w, at['job_created'], at['launched'] - at['job_created'], at['tested'] - at['launched']))
    if ab_var:
        continue
    print(f'learn {w:.2f} ms')
    if rep >= reps - 2:
        main = [(l, t) for l, t in tr if not l.startswith('w:')]
        work = [(l, t) for l, t in tr if l.startswith('w:')]
        print('  main   :', ', '.join(f'{l} {(t - t0) * 1e3:.2f}' for l, t in main))
        print('  workers:', ', '.join(f'{l[3:]} {(t - t0) * 1e3:.2f}' for l, t in work))
if ab_var:
    for v in ab_vals:
        cols = list(zip(*parts[v]))
        print(f'{ab_var}={v}: n={len(parts[v])}  wall median %.2f (min %.2f)  head %.2f  train %.2f  tail %.2f' %
              (statistics.median(cols[0]), min(cols[0]), statistics.median(cols[1]), statistics.median(cols[2]), statistics.median(cols[3])))
for v, rows in steps.items():
    print(f'  {v}: median ms per step of the main thread:', ', '.join(f'{k} {statistics.median([r[k] for r in rows if k in r]):.2f}' for k in rows[0]))
if not ab_var:
    print('median of the repetitions after the first: %.2f ms' % statistics.median(walls[1:]), ' all:', ' '.join('%.2f' % w for w in walls))
