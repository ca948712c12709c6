#!/usr/bin/env python3
"""Per-rank host timeline of ONE Sisa(parallel).learn at BASELINE.json configs[3]'s shape on R ranks that share the visible GPU
(gloo; the production transport is RCCL): where a rank's wall time goes between the start of the call and the logs.

    python tools/multirank_timeline.py [--ranks 2] [--k 128] [--epochs 2] [--shards 32] > timeline.json

Prints one JSON object: {rank: {'learn_s', 'marks': [[label, ms since the call], ...], 'host_cpus', 'owned'}}."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
k, epochs, shards, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
from ultrare_amd import engine, rng, synth
from ultrare_amd.method.sisa import Sisa, assign_shards
from ultrare_amd.read import RatingData, loadData
torch.cuda.set_device(0)
dist.init_process_group('gloo')
rank = dist.get_rank()
spec = synth.ML25M
data = synth.make_dataset(**spec)
shard_of, groups = synth.uniform_shards(spec['n_user'], shards)
def arr(t):
    return np.vstack([t[0].astype(np.float64), t[1].astype(np.float64), t[2] / 5.0])
parts_te = synth.split_shards(data['test'], shard_of, shards)
trd = [loadData(RatingData(arr(p)), 30000, 24) for p in synth.split_shards(data['train'], shard_of, shards)]
ted = [loadData(RatingData(arr(p)), 30000, 24, False) for p in parts_te]
tot = loadData(RatingData(arr(tuple(np.concatenate([p[c] for p in parts_te]) for c in range(3)))), 30000, 24, False)
class P:
    lam, seed, batch, lr, lr_decay, momentum, parallel = 0.1, 42, 30000, 0.001, 0.95, 0.9, True
    n_user, n_item = spec['n_user'], spec['n_item']
P.k, P.epochs = k, epochs
res = {}
for rep in range(2):                       # the second request is the one reported (pools, library, eval sets warm)
    s = Sisa(P, 'mf', shards, groups)
    torch.manual_seed(42)
    dist.barrier()
    torch.cuda.synchronize()
    if engine.HOST_TRACE is not None:
        engine.HOST_TRACE.clear()
    t0 = time.perf_counter()
    s.learn(trd, ted, tot, 0, '')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = {'learn_s': round(t1 - t0, 4), 'marks': [[lab, round((t - t0) * 1e3, 2)] for lab, t in (engine.HOST_TRACE or [])],
           'host_cpus': rng.host_cpus(), 'owned': int(sum(1 for o in assign_shards([len(d.dataset) for d in trd], dist.get_world_size()) if o == rank)),
           'normals_drawn': rng.STATS['normals'], 'draws_skipped': rng.STATS['skipped_draws'], 'total_rmse_epoch0': s.log['total_rmse'][0]}
    rng.STATS.update(normals=0, skipped_draws=0)
with open(os.path.join(out, f'rank{rank}.json'), 'w') as f:
    json.dump(res, f)
dist.destroy_process_group()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ranks', type=int, default=2)
    ap.add_argument('--k', type=int, default=128)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--shards', type=int, default=32)
    ap.add_argument('--port', type=int, default=29671)
    a = ap.parse_args()
    import tempfile
    out = tempfile.mkdtemp()
    script = os.path.join(out, 'worker.py')
    with open(script, 'w') as f:
        f.write(WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(a.port), WORLD_SIZE=str(a.ranks), LOCAL_WORLD_SIZE=str(a.ranks), URE_HOST_TRACE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(a.k), str(a.epochs), str(a.shards), out], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.DEVNULL) for r in range(a.ranks)]
    rc = [p.wait(timeout=1000) for p in procs]
    if any(rc):
        raise SystemExit(f'ranks ended with {rc}')
    res = {'command': ' '.join(sys.argv), 'ranks': {}}
    for r in range(a.ranks):
        with open(os.path.join(out, f'rank{r}.json')) as f:
            res['ranks'][str(r)] = json.load(f)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
