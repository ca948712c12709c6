#!/usr/bin/env python3
"""Experiment: what would Sisa.learn take if the host RNG draws (model inits, permutations, their uploads) cost nothing?
rng.draws_batch_async is memoised by the shapes of the request, so every repetition after the first finds the inits and the
permutations on the device (NOT valid for the product: a new request must draw).  Prints learn ms with and without."""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ultrare_amd import rng
import e2e_sisa

KW = dict(shards=int(sys.argv[1]), k=int(sys.argv[2])) if len(sys.argv) > 2 else {}
real = rng.draws_batch_async
cache = {}


def memo(specs, n_workers=0, tasks=None):
    key = json.dumps([(sp.get('n_rows', 0), sp['k'], sp['epochs'], sp['n_user'], sp['n_item']) for sp in specs])
    if key not in cache:
        cache[key] = real(specs, n_workers, tasks=tasks)
        for f in cache[key]:
            f.result()
    return cache[key]


out = {}
for name, fn in (('draws as in the product', real), ('draws memoised (free)', memo)):
    rng.draws_batch_async = fn
    rng.release = (lambda perms: None) if fn is memo else rng.release
    ts = [e2e_sisa.measure(reps=4, **KW)['learn_s'] * 1e3 for _ in range(4)]
    out[name] = {'learn_ms_median': round(statistics.median(ts), 2), 'learn_ms_min': round(min(ts), 2)}
print(json.dumps(out))
