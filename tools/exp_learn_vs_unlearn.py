#!/usr/bin/env python3
"""The marks of the calling thread inside the learn and the unlearn of ultrare_amd.measure.sisa_request (what bench.py times), step by step, medians."""
import os, sys, statistics
os.environ['URE_HOST_TRACE'] = '1'
sys.path.insert(0, os.getcwd())
import numpy as np, torch, time
from ultrare_amd import engine, measure
from ultrare_amd.method import sisa as S
rows = {'learn': [], 'unlearn': []}
extra = {'learn': [], 'unlearn': []}
orig_learn, orig_unlearn = S.Sisa.learn, S.Sisa.unlearn
def wrap(kind, f):
    def g(self, *a, **k):
        engine.HOST_TRACE.clear()
        t0 = time.perf_counter()
        out = f(self, *a, **k)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self._check_closed()
        t3 = time.perf_counter()
        extra[kind].append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        tr = [(l.split(' (')[0], (t - t0) * 1e3) for l, t in engine.HOST_TRACE if not l.startswith('w:')]
        rows[kind].append({b[0]: b[1] - a_[1] for a_, b in zip([('t0', 0.0)] + tr[:-1], tr)})
        return out
    return g
S.Sisa.learn = wrap('learn', orig_learn); S.Sisa.unlearn = wrap('unlearn', orig_unlearn)
r = measure.sisa_request(5, 32, 50, 1, 2.0, reps=9)
print(r['learn_s'], r['unlearn_s'], r['learn_s_all'], r['unlearn_s_all'])
for kind in ('learn', 'unlearn'):
    rs = rows[kind][2:]
    keys = list(rs[0].keys())
    print(kind, 'call / sync / join medians:', [round(statistics.median(c), 2) for c in zip(*extra[kind][2:])], 'all calls:', [round(x[0], 1) for x in extra[kind]])
    print(kind, ', '.join(f'{k} {statistics.median([x.get(k, 0) for x in rs]):.2f}' for k in keys))
