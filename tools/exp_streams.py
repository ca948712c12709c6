#!/usr/bin/env python3
"""Experiment: the bench workload's 5 shards as G independent jobs on G streams (are the
latency-bound step kernels better overlapped across streams than batched in one launch?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import engine, rng, synth

spec = synth.ML1M
data = synth.make_dataset(**spec)
S, d, B, E = 5, 32, 30000, 40
shard_of, _ = synth.uniform_shards(spec['n_user'], S)
parts = synth.split_shards(data['train'], shard_of, S)
torch.manual_seed(42)
inits = [rng.mf_init(spec['n_user'], spec['n_item'], d) for _ in parts]
perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
for groups in ([[0, 1, 2, 3, 4]], [[0, 1, 2], [3, 4]], [[0, 1], [2, 3], [4]], [[0], [1], [2], [3], [4]]):
    jobs, streams = [], []
    for g in groups:
        jobs.append(engine.TrainJob([engine.ShardData(*parts[i], spec['n_user'], spec['n_item']) for i in g],
                                    [inits[i] for i in g], [perms[i] for i in g], d, B, E, 1e-3, 0.1, 0.9))
        streams.append(torch.cuda.Stream())
    ticks = 7 * 25
    for j, st in zip(jobs, streams):
        j.run(14, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # interleave enqueues so that no stream runs ahead of the host
    for t in range(0, ticks, 7):
        for j, st in zip(jobs, streams):
            j.run(7, stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(min(len(p[0]), 0) for p in parts)
    inter = 0
    for p in parts:
        steps = (len(p[0]) + B - 1) // B
        for t in range(14, 14 + ticks):
            inter += (len(p[0]) - (steps - 1) * B) if t % steps == steps - 1 else B
    print(len(groups), 'streams:', round(dt * 1e3, 3), 'ms for', ticks, 'ticks ->', round(inter / dt / 1e9, 2), 'G inter/s', flush=True)
    for j in jobs:
        j.close()
