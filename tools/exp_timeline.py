#!/usr/bin/env python3
"""Inside one step launch: when every workgroup starts and ends (diagnostic build).

    python -m ultrare_amd.build --timeline tools/ab/libtimeline.so      # here (cross-compiles)
    python tools/exp_timeline.py [--lib tools/ab/libtimeline.so]         # on the GPU box

Prints, for 14 consecutive launches of the bench workload (ml-1m shape, 5 shards, d=32), the start
/ duration / end distribution of the workgroups by kind: `multi` = unit workgroups holding a row cut
into several units, `single` = unit workgroups of one-unit rows, `rider` = batch-tag preparation.
"""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=os.path.join(ROOT, 'tools', 'ab', 'libtimeline.so'))
ap.add_argument('--shards', type=int, default=5)
a = ap.parse_args()
from ultrare_amd import _native as nv
nv.LIB_PATH = os.path.abspath(a.lib)
from ultrare_amd import engine, rng, synth

spec = synth.ML1M
data = synth.make_dataset(**spec)
S, d, B, E = a.shards, 32, 30000, 12
shard_of, _ = synth.uniform_shards(spec['n_user'], S)
parts = synth.split_shards(data['train'], shard_of, S)
torch.manual_seed(42)
inits = [rng.mf_init(spec['n_user'], spec['n_item'], d) for _ in parts]
perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
shards = [engine.ShardData(*p, spec['n_user'], spec['n_item']) for p in parts]
job = engine.TrainJob(shards, inits, perms, d, B, E, 1e-3, 0.1, 0.9)
SLOTS = 16384
buf = torch.zeros(16 * SLOTS * 8, dtype=torch.int64, device='cuda')
L = nv.lib()
L.ure_debug_timeline.argtypes = [ctypes.c_void_p]
job.run(28)
torch.cuda.synchronize()
assert L.ure_debug_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
job.run(14)
torch.cuda.synchronize()
L.ure_debug_timeline(ctypes.c_void_p(0))
tl = buf.cpu().numpy().reshape(16, SLOTS, 8)
upb = 256 // (d // 4 if d <= 32 else d // 8)
heavy = [((sh.units(d).cpu().numpy()[:, 3] >> 30) & 1).reshape(-1, upb)[:, 0] for sh in shards]
spans = []
for tick in range(28, 42):
    t = tl[tick & 15]
    idx = np.nonzero(t[:, 0] > 0)[0]
    base = t[idx, 0].min()
    st, en = (t[idx, 0] - base) / 100.0, (t[idx, 1] - base) / 100.0        # us (100 MHz counter)
    x, j = idx & 7, idx >> 3                                               # sliced mapping of mf_step (URE_SHARD_FAST=2)
    sl = S * x + j % S
    sh, wg = sl >> 3, (j // S) * 8 + (sl & 7)
    kind = np.full(len(idx), 'rider ', dtype='U6')
    for k in range(S):
        m = (sh == k) & (wg < len(heavy[k]))
        kind[m] = np.where(heavy[k][wg[m]] > 0, 'multi ', 'single')
    spans.append(en.max())
    if tick == 34:      # per XCD (workgroup id mod 8): which shards it serves, how long its row workgroups take
        for x in range(8):
            m = ((idx & 7) == x) & (kind != 'rider ')
            dx = en[m] - st[m]
            print(f'      xcd {x}: shards {sorted(set(sh[m].tolist()))} row workgroups {m.sum()} dur med {np.median(dx):.2f} p90 {np.percentile(dx, 90):.2f} '
                  f'max {dx.max():.2f} end max {en[m].max():.2f} | riders {(((idx & 7) == x) & (kind == "rider ")).sum()}')
    if tick == 34:      # the slowest workgroups of one launch: which units do they hold?
        dur_all = en - st
        for q in np.argsort(-dur_all)[:12]:
            if kind[q] == 'rider ':
                print(f'      slow: rider shard {sh[q]} wg {wg[q]} dur {dur_all[q]:.2f} start {st[q]:.2f}')
                continue
            un = shards[sh[q]].units(d).cpu().numpy()[wg[q] * upb:(wg[q] + 1) * upb]
            lens = un[:, 2] - un[:, 1]
            print(f'      slow: shard {sh[q]} wg {wg[q]} dur {dur_all[q]:.2f} start {st[q]:.2f} rows {len(set(un[un[:, 0] >= 0, 0].tolist()))} '
                  f'max unit {lens.max()} slots {lens.sum()} max count {((un[:, 3] >> 16) & 0x1FFF).max()}')
    print(f'tick {tick}: {len(idx)} workgroups, span {en.max():.2f} us')
    for kd in ('multi ', 'single', 'rider '):
        m = kind == kd
        if not m.any():
            continue
        dur = en[m] - st[m]
        if kd != 'rider ':
            ph = (t[idx[m]][:, 2:5] - base) / 100.0
            ok = (t[idx[m]][:, 2:5] > 0).all(axis=1)
            if ok.any():
                a0, a1, a2 = ph[ok, 0] - st[m][ok], ph[ok, 1] - ph[ok, 0], ph[ok, 2] - ph[ok, 1]
                a3 = en[m][ok] - ph[ok, 2]
                print(f'   {kd} phases (first lane, med / p90 us): descriptor+row {np.median(a0):.2f}/{np.percentile(a0, 90):.2f}  scan {np.median(a1):.2f}/{np.percentile(a1, 90):.2f}  '
                      f'gathers {np.median(a2):.2f}/{np.percentile(a2, 90):.2f}  combine+update {np.median(a3):.2f}/{np.percentile(a3, 90):.2f}')
        print(f'   {kd} n={m.sum():5d} start med {np.median(st[m]):6.2f} max {st[m].max():6.2f} | dur med {np.median(dur):6.2f} '
              f'p90 {np.percentile(dur, 90):6.2f} max {dur.max():6.2f} | end med {np.median(en[m]):6.2f} max {en[m].max():6.2f}')
print('mean span', round(float(np.mean(spans)), 2), 'us')
job.close()
