#!/usr/bin/env python3
"""bench.py -- training interactions/s of the SISA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ml-1m-shaped synthetic ratings (6040 x 3416,
896,914 train rows; the real file is not shipped with the reference), 5 SISA shards
(the reference's uniform grouping), d = 32, batch 30,000, SGD-momentum-L2 -- all five
shards resident in HBM and trained side by side, one optimizer step of each per launch.
A "step" of this bench = one epoch of the largest shard = `ticks_per_step` launches; the
number of interactions processed in the timed ticks is counted exactly.
N > 1: weak scaling -- every rank trains its own 5-shard job (shards are independent,
sisa.py:33-36; there is no data-path collective), value = all ranks' interactions /
max-over-ranks time.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['ml1m', 'ml25m'], default='ml1m',
                    help="ml1m: BASELINE configs[1] (default); ml25m: configs[3] shape (32 shards, d=128) on one GPU")
    ap.add_argument('--shards', type=int, default=None)
    ap.add_argument('--d', type=int, default=None)
    ap.add_argument('--batch', type=int, default=30000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unlearn', action='store_true', help='skip the learn/unlearn wall-time leg')
    ap.add_argument('--cpu-budget', type=float, default=15.0, help='seconds of CPU baseline work')
    ap.add_argument('--roofline-steps', type=int, default=3)
    ap.add_argument('--backend', default='nccl', help='process-group backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    ap.add_argument('--force-device', type=int, default=None, help='rehearsal only: put every rank on this device')
    return ap.parse_args()


def interactions_in_ticks(sizes, batch, t0, t1, epochs, dense_bytes=None):
    """Exact number of interactions the ticks [t0, t1) process over all shards, and the
    algorithmic bytes of every tick ((16 + 16 d) per interaction + 20 P per dense
    optimizer step, P counted over the rows actually streamed)."""
    total = 0
    per_tick = np.zeros(max(t1 - t0, 0), dtype=np.int64)
    dense = np.zeros(max(t1 - t0, 0), dtype=np.int64)
    for k, n in enumerate(sizes):
        steps = (n + batch - 1) // batch
        last = n - (steps - 1) * batch
        for t in range(t0, min(t1, steps * epochs)):
            bs = last if (t % steps) == steps - 1 else batch
            per_tick[t - t0] += bs
            dense[t - t0] += dense_bytes[k] if dense_bytes is not None else 0
            total += bs
    return total, per_tick, dense


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if a.force_device is not None:
        local = a.force_device
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(a.backend)

    from ultrare_amd import engine, rng, synth
    from ultrare_amd import _native as nv

    # ---- inputs: generated and made resident before anything is timed ----------
    spec = synth.ML1M if a.workload == 'ml1m' else synth.ML25M
    a.shards = a.shards or (5 if a.workload == 'ml1m' else 32)
    a.d = a.d or (32 if a.workload == 'ml1m' else 128)
    data = synth.make_dataset(**spec, seed=synth.SEED + rank)
    shard_of, _ = synth.uniform_shards(spec['n_user'], a.shards)
    parts = synth.split_shards(data['train'], shard_of, a.shards)
    sizes = [len(p[0]) for p in parts]
    steps_per_epoch = [(n + a.batch - 1) // a.batch for n in sizes]
    tps = max(steps_per_epoch)                                     # ticks per bench step
    n_bench_steps = a.warmup + a.steps + a.roofline_steps
    epochs = int(np.ceil(n_bench_steps * tps / min(steps_per_epoch))) + 1
    torch.manual_seed(42 + rank)
    inits, perms = [], []
    t_rng = time.perf_counter()
    for p in parts:
        inits.append(rng.mf_init(spec['n_user'], spec['n_item'], a.d))
        perms.append(rng.epoch_perms(rng.epoch_seeds(epochs, True), len(p[0]), threads=min(8, os.cpu_count() or 1)))
    t_rng = time.perf_counter() - t_rng
    shards = [engine.ShardData(*p, spec['n_user'], spec['n_item']) for p in parts]
    job = engine.TrainJob(shards, inits, perms, a.d, a.batch, epochs, 1e-3, 0.1, 0.9, 0.95)
    del perms

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- warmup -----------------------------------------------------------------
    job.run(a.warmup * tps)
    torch.cuda.synchronize()
    # ---- timed region: exactly K steps --------------------------------------------
    t0_tick = job.done
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0 = time.perf_counter()
    ev0.record()
    job.run(a.steps * tps)
    ev1.record()
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    barrier()
    wall = w1 - w0
    dev_ms = ev0.elapsed_time(ev1)
    n_inter, _, _ = interactions_in_ticks(sizes, a.batch, t0_tick, job.done, epochs)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        c = torch.tensor([n_inter], dtype=torch.int64, device='cuda')
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_total = int(c.item())
    else:
        n_total = n_inter

    # ---- roofline: the timed region consists of step-kernel launches only (the next epoch's
    # batch tags ride inside them), so the kernel's average duration is the HIP-event time of the
    # region / launches, on the stream the launches went to.  A second pass with one event pair
    # per launch (ure_job_train_profiled) is reported beside it.
    dp = engine.pad_dim(a.d)
    rows_streamed = [sh.n_active if job.lazy_rows else spec['n_user'] + spec['n_item'] for sh in shards]
    _, per_tick, dense = interactions_in_ticks(sizes, a.batch, t0_tick, t0_tick + a.steps * tps, epochs,
                                               dense_bytes=[20 * r * dp for r in rows_streamed])
    n_launch = a.steps * tps
    b_sparse = 16 + 16 * dp
    alg_bytes = float((per_tick * b_sparse + dense).sum()) / n_launch                          # per launch
    avg_ms = dev_ms / n_launch
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    step_ms, n_step, assign_ms, n_assign = job.run_profiled(a.roofline_steps * tps)
    # HBM-side bytes per launch come from rocprofv3 PMC passes of this same command (they cannot
    # be collected from inside the process); the committed summary is quoted with its source
    traffic, traffic_src = None, None
    import glob
    import re
    pmc = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*pmc_hbm_traffic.json')),
                 key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.relpath(f, ROOT))])   # v9 < v10
    if pmc and a.workload == 'ml1m' and a.d == 32 and a.shards == 5 and a.batch == 30000:
        try:
            with open(pmc[-1]) as f:
                k = [v for n, v in json.load(f)['kernels'].items() if 'mf_step_kernel' in n][0]
            traffic, traffic_src = k['traffic_bytes_per_launch'], os.path.relpath(pmc[-1], ROOT)
        except Exception:
            pass
    roofline = {'bound': 'hbm', 'kernel': 'mf_step_kernel', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                'alg_bytes_per_launch': round(alg_bytes), 'avg_launch_us': round(avg_ms * 1e3, 2),
                'launches_timed': n_launch, 'per_launch_event_us': round(step_ms / max(n_step, 1) * 1e3, 2),
                'dense_rows_streamed_per_shard': rows_streamed, 'lazy_rows': bool(job.lazy_rows)}

    # ---- CPU baseline (rank 0, N = 1): the torch DataLoader port of the reference ---
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import torch_port
        workers = min(24, os.cpu_count() or 1)
        torch.manual_seed(42)
        _, seen, spent, _ = torch_port.train_shard(parts[0], spec['n_user'], spec['n_item'], a.d, a.batch, 50,
                                                   workers=workers, budget_s=a.cpu_budget)
        pre = torch_port.prebatched_rate(parts[0], spec['n_user'], spec['n_item'], a.d, a.batch, 2)
        cpu = {'value': round(seen / spent, 1), 'unit': 'interactions/s', 'cores': workers, 'kind': 'port',
               'sample': f'shard 0 ({sizes[0]} rows), {seen // sizes[0]} epoch(s) = {seen} interactions in {spent:.1f}s; '
                         f'per-sample Dataset + DataLoader({workers} workers) + nn.Embedding + SGD, '
                         f'{torch.get_num_threads()} torch threads, host has {os.cpu_count()} cpus',
               'prebatched_value': round(pre, 1)}

    # ---- second half of the metric (rank 0, N = 1): Sisa.learn, then Sisa.unlearn after a 2 %
    # random user deletion, 50 epochs, through the operator surface, wall clock with the per-epoch
    # evaluations, row merge and final test included (outside the timed region above)
    unlearn = None
    if rank == 0 and world == 1 and a.workload == 'ml1m' and not a.no_unlearn:
        job.close()
        import importlib.util
        sp = importlib.util.spec_from_file_location('e2e_sisa', os.path.join(ROOT, 'tools', 'e2e_sisa.py'))
        e2e = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(e2e)
        r = e2e.measure(a.shards, a.d, 50, 1, 2.0, data=data)
        unlearn = {'learn_wall_s': r['learn_s'], 'unlearn_wall_s': r['unlearn_s'], 'epochs': 50,
                   'deleted_users': r['deleted_users'], 'retrained_shards': r['retrained_shards'],
                   'unlearn_interactions': r['unlearn_interactions'],
                   'includes': 'host RNG + layout, 50 epochs of all retrained shards side by side, per-epoch shard/total '
                               'evaluations, row merge, final test; inputs as in-memory loaders',
                   'final_test': {'learn': r['log0'], 'unlearn': r['unlearn_log0']}}
        # BASELINE.json configs[4]: 16 shards (d = the reference's default k = 16), 2 % random deletion
        r16 = e2e.measure(16, 16, 50, 1, 2.0, data=data)
        unlearn['config4_16_shards_k16'] = {'learn_wall_s': r16['learn_s'], 'unlearn_wall_s': r16['unlearn_s'],
                                            'retrained_shards': r16['retrained_shards'], 'deleted_users': r16['deleted_users']}

    if rank == 0:
        arch = ''
        try:
            import ctypes
            buf = ctypes.create_string_buffer(64)
            nv.lib().ure_device_info(local, None, None, buf, 64)
            arch = buf.value.decode()
        except Exception:
            pass
        out = {
            'metric': 'training interactions/sec + unlearn retrain wall-time, ml-1m 5-shard SISA' if a.workload == 'ml1m' else f'training interactions/sec, synthetic ml-25m-scale {a.shards}-shard SISA',
            'value': round(n_total / wall, 1), 'unit': 'interactions/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(wall * 1e3 / a.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{a.workload}-shaped synthetic {spec["n_user"]}x{spec["n_item"]}, {spec["n_train"]} train rows, '
                                   f'{a.shards}-shard SISA (uniform grouping), d={a.d}, batch={a.batch}, SGD-momentum-L2, '
                                   f'all shards of a rank side by side',
                       'shards_per_gpu': a.shards, 'shard_rows': sizes, 'ticks_per_step': tps, 'parallelism': f'shards x{world}',
                       'arch': arch},
            'device_ms_timed': round(dev_ms, 3), 'interactions_timed': n_total,
            'host_rng_prep_s': round(t_rng, 3),
            'roofline': roofline, 'cpu_baseline': cpu, 'unlearn': unlearn,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
