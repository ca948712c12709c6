#!/usr/bin/env python3
"""Where the host time of one shard's ingest goes (ml-1m size, 5 shards): triples(), the native layout builder, the upload,
the device fills -- one shard at a time, then all five on the worker pool as Sisa does.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import engine, rng, synth, _native as nv
from ultrare_amd.read import RatingData, loadData

data = synth.make_dataset(**synth.ML1M)
S = 5
shard_of, groups = synth.uniform_shards(data['n_user'], S)
parts = synth.split_shards(data['train'], shard_of, S)
nu, ni = data['n_user'], data['n_item']
dev = engine._device()


def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 3), r


out = {}
ds = RatingData(np.vstack(parts[0]))
out['triples_ms'], tr = t(ds.triples)
u, i, r = tr
out['minmax_ms'], _ = t(lambda: (u.min(), u.max(), i.min(), i.max()))
out['build_layout_numpy_ms'], _ = t(lambda: nv.build_layout(u, i, r, nu, ni))
out['shard_data_ms'], sh = t(lambda: engine.ShardData(u, i, r, nu, ni))
out['shard_data_x5_serial_ms'], _ = t(lambda: [engine.ShardData(*RatingData(np.vstack(p)).triples(), nu, ni) for p in parts])


def batched(wait=True):
    from ultrare_amd.read import shard_layouts
    lo = [loadData(RatingData(np.vstack(p)), 30000, 24, True) for p in parts]
    t0 = time.perf_counter()
    shard_layouts(lo, nu, ni, dev)
    if wait:
        torch.cuda.synchronize()
    return time.perf_counter() - t0


out['shard_layouts_x5_batched_ms'] = round(min(batched() for _ in range(5)) * 1e3, 3)
out['shard_layouts_x5_batched_host_only_ms'] = round(min(batched(False) for _ in range(5)) * 1e3, 3)
raw = [(RatingData(np.vstack(p)).users, RatingData(np.vstack(p)).items, RatingData(np.vstack(p)).ratings) for p in parts]
regs = [np.empty(nv.layout_region_words(len(r[0]), nu, ni), dtype=np.int32) for r in raw]
out['native_build_layouts_x5_ms'], _ = t(lambda: nv.build_layouts(raw, nu, ni, regs, threads=5))
out['native_build_layouts_x5_1thread_ms'], _ = t(lambda: nv.build_layouts(raw, nu, ni, regs, threads=1))
n = len(u)
out['device_fills_ms'], _ = t(lambda: (torch.full((2, sh.n_slots), -1, dtype=torch.int16, device=dev), torch.full((n,), -1, dtype=torch.int16, device=dev),
                                       torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(8000, dtype=torch.int32, device=dev)))
print(json.dumps(out))
