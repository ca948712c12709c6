"""Full MF at the 25 M shape in touch_mode 3 (csrc/mf_index.h), one library build per process: step-launch time (event pair per launch),
epoch-start time, interactions/s.  The synthetic set is cached under /tmp so that several builds can be compared in one session.
    URE_LIB=tools/ab/lib_x.so URE_ALLOW_STALE_LIB=1 python tools/exp_index.py [--d 128] [--epochs 2] [--batch 30000]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dataset():
    from ultrare_amd import synth
    cache = '/tmp/ure_ml25m_train.npz'
    if os.path.exists(cache):
        z = np.load(cache)
        return (z['u'], z['i'], z['r']), synth.ML25M
    data = synth.make_dataset(**synth.ML25M, seed=synth.SEED)
    u, i, r = data['train']
    np.savez(cache, u=u, i=i, r=r)
    return (u, i, r), synth.ML25M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--batch', type=int, default=30000)
    ap.add_argument('--label', default=os.environ.get('URE_LIB', 'product'))
    a = ap.parse_args()
    from ultrare_amd import engine, rng
    part, spec = dataset()
    part = (part[0], part[1], (part[2] / 5.0).astype(np.float32))
    n = len(part[0])
    torch.manual_seed(42)
    init = rng.mf_init(spec['n_user'], spec['n_item'], a.d)
    if a.d > 64:
        init = tuple(t * 0.3 for t in init)
    E = 2 * a.epochs + 1
    tags = rng.epoch_tags(rng.epoch_seeds(E, True), n, a.batch, threads=min(16, os.cpu_count() or 1))
    sh = engine.ShardData(*part, spec['n_user'], spec['n_item'])
    job = engine.TrainJob([sh], [init], [tags], a.d, a.batch, E, 1e-3, 0.1, 0.9, 0.95, final_only=True)
    steps = job.steps_per_epoch(0)
    job.run(steps)                       # warm-up epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms, n_step, prep_ms, n_prep = job.run_profiled(a.epochs * steps)
    wall = time.perf_counter() - t0
    out = {'label': a.label, 'd': a.d, 'touch': job.touch, 'index': getattr(job, 'index', False), 'steps_per_epoch': steps,
           'step_us': round(step_ms / n_step * 1e3, 2), 'epoch_start_ms': round(prep_ms / a.epochs, 3), 'prep_launch_groups': n_prep,
           'rows_per_step': job.touch_rows_per_step(), 'epoch_ms_device': round((step_ms + prep_ms) / a.epochs, 3),
           'minter_per_s': round(n * a.epochs / ((step_ms + prep_ms) * 1e-3) / 1e6, 1), 'wall_s_profiled': round(wall, 3)}
    # the same number of epochs without an event pair per launch: device time of whole epochs, their starts included
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    job.run(a.epochs * steps)
    ev1.record()
    torch.cuda.synchronize()
    out['epoch_ms_plain'] = round(ev0.elapsed_time(ev1) / a.epochs, 3)
    out['minter_per_s_plain'] = round(n * a.epochs / (ev0.elapsed_time(ev1) * 1e-3) / 1e6, 1)
    out['overlap'] = os.environ.get('URE_INDEX_OVERLAP', '1')
    job.close()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
