#!/usr/bin/env python3
"""configs[3] shape on one GPU (32 shards, d = 128, 22.5 M rows): the default step kernel against touch mode
(csrc/mf_touch.h) on the same resident shards.  Prints one JSON object.

    python tools/exp_touch.py [--steps 4] [--shards 32] [--d 128]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--shards', type=int, default=32)
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--batch', type=int, default=30000)
    ap.add_argument('--modes', default='0,1')
    a = ap.parse_args()
    from ultrare_amd import engine, rng, synth
    t0 = time.time()
    spec = synth.ML25M
    data = synth.make_dataset(**spec)
    shard_of, _ = synth.uniform_shards(spec['n_user'], a.shards)
    parts = synth.split_shards(data['train'], shard_of, a.shards)
    sizes = [len(p[0]) for p in parts]
    spe = [(n + a.batch - 1) // a.batch for n in sizes]
    tps = max(spe)
    epochs = int(np.ceil((a.steps + 2) * tps / min(spe))) + 1
    torch.manual_seed(42)
    inits = [rng.mf_init(spec['n_user'], spec['n_item'], a.d) for _ in parts]
    perms = [rng.epoch_perms(rng.epoch_seeds(epochs, True), n, threads=8) for n in sizes]
    shards = [engine.ShardData(*p, spec['n_user'], spec['n_item']) for p in parts]
    print(f'prepared in {time.time() - t0:.0f}s: {a.shards} shards, {sum(sizes)} rows, tps {tps}, active rows {[s.n_active for s in shards[:3]]}', file=sys.stderr, flush=True)
    out = {'shards': a.shards, 'd': a.d, 'rows': int(sum(sizes)), 'ticks_per_step': tps}
    for mode in [int(x) for x in a.modes.split(',')]:
        job = engine.TrainJob(shards, inits, perms, a.d, a.batch, epochs, 1e-3, 0.1, 0.9, 0.95, touch=bool(mode))
        job.run(tps)                      # warm-up: one epoch of the largest shard (touch: includes the first preparation)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        job.run(a.steps * tps)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
        sm, ns, am, na = job.run_profiled(tps)
        n_inter = 0
        for n, st in zip(sizes, spe):
            for t in range(tps, tps + a.steps * tps):
                if t < st * epochs:
                    n_inter += (n - (st - 1) * a.batch) if t % st == st - 1 else a.batch
        out['touch' if mode else 'default'] = {
            'touch': job.touch, 'ms_per_tick_incl_epoch_prep': round(ms / (a.steps * tps), 4),
            'interactions_per_s': round(n_inter / (ms * 1e-3), 1),
            'step_kernel_us_events': round(sm / max(ns, 1) * 1e3, 1), 'prep_us_per_launch_pair': round(am / max(na, 1) * 1e3, 1), 'prep_launch_groups': na}
        job.close()
        del job
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
