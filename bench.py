#!/usr/bin/env python3
"""bench.py -- training interactions/s of the SISA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ml-1m-shaped synthetic ratings (6040 x 3416,
896,914 train rows; the real file is not shipped with the reference), 5 SISA shards
(the reference's uniform grouping), d = 32, batch 30,000, SGD-momentum-L2 -- all five
shards resident in HBM and trained side by side, one optimizer step of each per launch.
A "step" of this bench = one epoch of the largest shard = `ticks_per_step` launches; the
number of interactions processed in the timed ticks is counted exactly.

N > 1 -- one process per GPU over RCCL.  Started by the driver under torch.distributed.run
(RANK / WORLD_SIZE in the environment) or, when WORLD_SIZE is unset, by this script itself:
it starts N rank processes before anything touches the GPU and relays rank 0's JSON line.
  * headline `value`: weak scaling -- every rank trains its own 5-shard ml-1m job (shards are
    independent, sisa.py:33-36; no data-path collective), value = all ranks' interactions /
    max-over-ranks time;
  * `exchange`: the path's one collective (sisa.py:52-58), the all-gather of every shard's own
    user rows + item table, timed over RCCL on the tables just trained;
  * `north_star_splits`: ONE job's shards placed across the ranks by longest-processing-time-first
    (strong scaling) for BASELINE.json configs[2] (8 shards, d = 64), configs[3] (25 M ratings,
    32 shards, d = 128) and configs[4] (16-shard learn + unlearn through Sisa, wall time).
  --split 1 runs a workload in that split mode as the headline instead.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['ml1m', 'ml25m'], default='ml1m',
                    help="ml1m: BASELINE configs[1] (default); ml25m: configs[3] shape (32 shards, d=128)")
    ap.add_argument('--shards', type=int, default=None)
    ap.add_argument('--d', type=int, default=None)
    ap.add_argument('--batch', type=int, default=30000)
    ap.add_argument('--split', type=int, default=0,
                    help='1: ONE job, its shards placed across the ranks (strong scaling); 0: every rank its own job (weak)')
    ap.add_argument('--splits', default='auto',
                    help="extra legs at N > 1: comma list of config2,config3,config4 ; 'auto' = all three ; 'none'")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unlearn', action='store_true', help='skip the learn/unlearn wall-time leg')
    ap.add_argument('--no-hbm-leg', action='store_true', help='N = 1: skip the configs[3]-shape leg (roofline_hbm)')
    ap.add_argument('--no-ot', action='store_true', help='N = 1: skip the OT grouping leg (ot)')
    ap.add_argument('--no-cold', action='store_true', help='N = 1: skip the cold unlearning request (unlearn.cold_request_s)')
    ap.add_argument('--extras-budget', type=float, default=240.0, help='N = 1: seconds after which no further optional leg is started')
    ap.add_argument('--cpu-budget', type=float, default=15.0, help='seconds of CPU baseline work')
    ap.add_argument('--roofline-steps', type=int, default=3)
    ap.add_argument('--no-resident', action='store_true', help='skip the second timed region (every tag resident): the process then launches the step kernel in the inclusive region, the warm-up and the event-pair pass only (profiles/rNN/bench_headline_kernel_stats.csv)')
    ap.add_argument('--extras-timeout', type=float, default=420.0, help='N > 1: seconds allowed for the exchange / split legs')
    ap.add_argument('--backend', default='nccl', help='process-group backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    ap.add_argument('--force-device', type=int, default=None, help='rehearsal only: put every rank on this device')
    ap.add_argument('--force-dist', action='store_true', help='rehearsal only: create the process group even for one rank, so that '
                                                               'the RCCL calls of the N > 1 path run on a one-GPU box')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, BEFORE this process touches the GPU
# ------------------------------------------------------------------------------------------------
SPAWN_GRACE_S = 10.0


def spawn_ranks(n):
    """`python bench.py --gpus N` with no WORLD_SIZE: N children of this script, one per GPU, with the
    torch.distributed environment of a single-node launch.  Children are fresh processes (never an
    exec of a process that initialised HIP); rank 0 inherits stdout and prints the JSON line."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool support dmabuf IPC only; with the legacy mode RCCL's
    # intra-node transport setup fails in hipIpcGetMemHandle ("invalid argument").  The image exports it already;
    # it is repeated here so that a caller's scrubbed environment cannot drop it.
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=None if r == 0 else subprocess.DEVNULL)
             for r in range(n)]
    rc, deadline = 0, None
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    # one rank failed: the others would wait in a collective for ever.  They get a grace period
                    # to fail (and say why) by themselves before they are terminated.
                    rc, deadline = code, time.time() + SPAWN_GRACE_S
            if deadline is not None and time.time() > deadline:
                for q in pending:
                    procs[q].terminate()
                deadline = float('inf')
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def interactions_in_ticks(sizes, batch, t0, t1, epochs, dense_bytes=None):
    """Exact number of interactions the ticks [t0, t1) process over the given shards, and the
    algorithmic bytes of every tick ((16 + 16 d) per interaction + the dense optimizer
    bytes of the rows actually streamed)."""
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


class Dist:
    """The process group (or its absence) behind three helpers."""

    def __init__(self, a):
        self.rank = int(os.environ.get('RANK', 0))
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.local = int(os.environ.get('LOCAL_RANK', 0)) if a.force_device is None else a.force_device
        torch.cuda.set_device(self.local)
        self.pg = None
        if self.world > 1 or a.force_dist:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29655')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
            if a.backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', self.local))
            else:
                dist.init_process_group(a.backend)
            self.pg = dist

    def barrier(self):
        if self.pg is not None:
            self.pg.barrier()

    def max(self, x):
        if self.pg is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device='cuda')
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x):
        if self.pg is None:
            return int(x)
        t = torch.tensor([x], dtype=torch.int64, device='cuda')
        self.pg.all_reduce(t, op=self.pg.ReduceOp.SUM)
        return int(t.item())


NO_RESIDENT = False


def train_leg(D, workload, n_shards, d, batch, steps, warmup, split, roofline_steps=0, keep_job=False, data=None):
    """Make the workload resident, run W warm-up and K timed bench steps of the rank's shards, return the
    measurements.  split: ONE job (same data on every rank), shards placed by assign_shards; otherwise
    every rank generates and trains its own job."""
    from ultrare_amd import engine, rng, synth
    from ultrare_amd.method.sisa import assign_shards
    spec = synth.ML1M if workload == 'ml1m' else synth.ML25M
    data = data or synth.make_dataset(**spec, seed=synth.SEED + (0 if split else D.rank))
    shard_of, groups = synth.uniform_shards(spec['n_user'], n_shards)
    parts = synth.split_shards(data['train'], shard_of, n_shards)
    all_sizes = [len(p[0]) for p in parts]
    owner = assign_shards(all_sizes, D.world) if split else [D.rank] * n_shards
    mine = [s for s in range(n_shards) if owner[s] == D.rank]
    sizes = [all_sizes[s] for s in mine]
    steps_per_epoch = [(n + batch - 1) // batch for n in (all_sizes if split else sizes)]
    tps = max(steps_per_epoch)                                     # ticks per bench step (job-wide)
    # The epochs' batches are part of what is timed (the reference's baseTrain iterates the DataLoader, utils.py:58 --
    # read.py:127-133 shuffles inside it): the tags come from the PRODUCT's path, rng.device_tags (csrc/perm_tags.hip: torch.randperm's
    # permutations made on the device from the epochs' seeds, on a side stream), and the shuffles of the epochs the timed ticks read are
    # launched AFTER the clock has started.  A second timed region of K steps with every tag resident gives `value_resident_tags`
    # (round 4's headline).  Shards beyond 2^20 rows / URE_DEVICE_TAGS=0: host-made tags, resident, one region.
    dev_tags = bool(mine) and rng.device_tags_wanted() and all(all_sizes[s] <= rng.DEVICE_TAGS_MAX_ROWS and -(-all_sizes[s] // batch) <= 65535 for s in mine)
    resident_steps = steps if dev_tags and not NO_RESIDENT else 0
    n_bench_steps = warmup + steps + resident_steps + roofline_steps
    min_steps = min(steps_per_epoch)
    epochs = int(np.ceil(n_bench_steps * tps / min_steps)) + 2
    t_rng = time.perf_counter()
    inits, perms, seeds = [], [], []
    if not split:
        torch.manual_seed(42 + D.rank)
    for s in mine:
        if split:
            torch.manual_seed(42 + 1000 * s)                       # own shards only: any fixed init serves a throughput leg
        inits.append(rng.mf_init(spec['n_user'], spec['n_item'], d))
        seeds.append(rng.epoch_seeds(epochs, True))
    t_rng = time.perf_counter() - t_rng
    t_tags = time.perf_counter()
    fire, fired, bounds, b1, timed_to, perms_methods = None, 0, None, 0, 0, []
    if dev_tags:
        dev = engine._device()
        tasks = [rng._task_of(dict(start_state=None, n_user=spec['n_user'], n_item=spec['n_item'], k=d, epochs=epochs, with_total_test=True,
                                   n_rows=all_sizes[s], shuffle=True, device=dev, tags_batch=batch, seeds=seeds[k]), buffers=False) for k, s in enumerate(mine)]
        # tick t reads the tags of epoch t // steps + 1 of a shard (prepared one epoch ahead; touch_mode 2: two) -- the smallest shard is the
        # furthest along.  Chunk 0: what the warm-up reads; chunk 1: what the timed ticks read beyond that; chunk 2: the rest.
        ahead = 2
        t0w, t1w = warmup * tps, (warmup + steps) * tps
        b1 = min(epochs, ((t0w - 1) // min_steps + 1 + ahead) if t0w > 0 else ahead)
        b2 = min(epochs, (t1w - 1) // min_steps + 1 + ahead)
        # ... each cut into the chunks the product launches a request's shuffles in (rng.default_tag_bounds: two small ones first)
        n_max = max(all_sizes[s] for s in mine)
        cuts = [0]
        for lo, hi in zip([0, b1, b2], [b1, b2, epochs]):
            if hi > lo:
                cuts += [lo + x for x in rng.default_tag_bounds(hi - lo, len(mine), n_max)[1:]]
        bounds = sorted(set(cuts))
        timed_to = b2
        fire = rng.device_tags(tasks, bounds=bounds, defer=True)
        assert fire, 'rng.device_tags refused the bench shards'
        perms = [t.perms_value for t in tasks]
        perms_methods = list(perms[0]._ure_methods)
    else:
        for k, s in enumerate(mine):
            # host-made batch tags (struct ure_shard: file_tags), resident before the clock starts; URE_HOST_TAGS=0: the permutations themselves
            if os.environ.get('URE_HOST_TAGS', '1') != '0' and -(-all_sizes[s] // batch) <= 65535:
                perms.append(rng.epoch_tags(seeds[k], all_sizes[s], batch, threads=min(8, os.cpu_count() or 1)))
            else:
                perms.append(rng.epoch_perms(seeds[k], all_sizes[s], threads=min(8, os.cpu_count() or 1)))
    t_tags = time.perf_counter() - t_tags

    def fire_until(epoch_end):
        nonlocal fired
        while fire is not None and fired + 1 < len(bounds) and bounds[fired + 1] <= epoch_end:
            fire(fired)
            fired += 1
    fire_until(b1 if fire else 0)
    job, shards = None, []
    if mine:
        shards = [engine.ShardData(*parts[s], spec['n_user'], spec['n_item']) for s in mine]
        job = engine.TrainJob(shards, inits, perms, d, batch, epochs, 1e-3, 0.1, 0.9, 0.95, final_only=True)     # (as Sisa runs it: tables read at the end)
    del perms, parts

    if job is not None:
        job.run(warmup * tps)
    torch.cuda.synchronize()

    def timed(n_steps, fire_to=None):
        """Exactly n_steps bench steps between barrier + synchronize on both sides; fire_to: the shuffles of the epochs up to there are launched
        right behind the region's first event."""
        t0_tick = job.done if job is not None else 0
        D.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        ev0.record()
        if fire_to is not None:
            fire_until(fire_to)
        if job is not None:
            job.wait_marks = []                                 # (event pairs around the stream's waits for tag chunks)
            job.run(n_steps * tps)
        ev1.record()
        torch.cuda.synchronize()
        w1 = time.perf_counter()
        D.barrier()
        t1_tick = job.done if job is not None else 0
        waited = 0.0
        if job is not None:
            waited = sum(e0.elapsed_time(e1) for e0, e1 in job.wait_marks)
            job.wait_marks = None
        n_inter, _, _ = interactions_in_ticks(sizes, batch, t0_tick, t1_tick, epochs)
        return {'wall': D.max(w1 - w0), 'my_wall': w1 - w0, 'dev_ms': ev0.elapsed_time(ev1), 'n_total': D.sum(n_inter), 'n_inter': n_inter,
                't0_tick': t0_tick, 'n_launch': t1_tick - t0_tick, 'waited_ms': waited}
    # ---- timed region: exactly K steps, barrier + synchronize on both sides; the epochs' shuffles inside it ----------------------
    shuffles_timed = (timed_to - b1) * len(mine) if fire else 0
    r = timed(steps, timed_to if fire else None)
    resident = None
    if resident_steps:
        fire_until(epochs)
        torch.cuda.synchronize()
        resident = timed(resident_steps)
    out = {'spec': spec, 'data': data, 'groups': groups, 'mine': mine, 'owner': owner, 'sizes': sizes, 'all_sizes': all_sizes,
           'tps': tps, 'epochs': epochs, 'wall': r['wall'], 'my_wall': r['my_wall'], 'dev_ms': r['dev_ms'], 'n_total': r['n_total'], 'n_inter': r['n_inter'],
           't_rng': t_rng, 't_tags': t_tags, 't0_tick': r['t0_tick'], 'n_launch': r['n_launch'], 'shards': shards, 'job': job, 'waited_ms': r['waited_ms'],
           'tag_chunks': ([{'epochs': [a_, b_], 'by': m} for a_, b_, m in zip(bounds[:-1], bounds[1:], perms_methods)] if fire else None),
           'value': r['n_total'] / r['wall'] if r['wall'] > 0 else 0.0,
           'batch_tags': ('device (rng.device_tags -> csrc/perm_chain.hip on a side stream, in the product\'s chunks), launched inside the timed region' if dev_tags else
                          'host-made, resident before the timed region'),
           'shuffles_in_timed_region': shuffles_timed, 'resident': resident}
    if not keep_job and job is not None:
        job.close()
        out['job'] = None
        out['shards'] = None
    return out


def source_hash():
    from ultrare_amd import build as lib_build
    return lib_build.step_kernel_hash()


def roofline_of(a, leg, d, batch):
    """Algorithmic bytes per launch / the step kernel's average launch duration (HIP events on the
    launch stream around the timed region: it holds step-kernel launches only -- the next epoch's batch
    tags ride inside them), the per-launch-event pass beside it, and the HBM-side traffic of the matching
    rocprofv3 PMC passes when their source hash equals this tree's."""
    from ultrare_amd import engine
    job, shards, spec = leg['job'], leg['shards'], leg['spec']
    dp = engine.pad_dim(d)
    rows_streamed = [sh.n_active if job.lazy_rows else spec['n_user'] + spec['n_item'] for sh in shards]
    if job.touch:      # touch mode: only the rows a step trains are read and rewritten (measured on the epoch's row masks)
        rows_streamed = [round(x, 1) for x in job.touch_rows_per_step()]
    n_launch = leg['n_launch']
    # SURVEY 8d counts 20 B per dense element (w, m, g read; w, m written); this kernel never materialises
    # g, so it moves 16 B: both figures are reported, `achieved` uses the survey's
    _, per_tick, dense20 = interactions_in_ticks(leg['sizes'], batch, leg['t0_tick'], leg['t0_tick'] + n_launch, leg['epochs'],
                                                 dense_bytes=[int(20 * r * dp) for r in rows_streamed])
    b_sparse = 16 + 16 * dp
    alg20 = float((per_tick * b_sparse + dense20).sum()) / n_launch
    alg16 = float((per_tick * b_sparse + dense20 * 16 // 20).sum()) / n_launch
    # the region holds the epochs' shuffles on the side stream and, on the launch stream, the waits for their chunks: the step kernel's
    # own duration is the region minus those waits (event pairs around them) over its launches -- what rocprofv3 reads for the kernel
    avg_ms = (leg['dev_ms'] - leg.get('waited_ms', 0.0)) / n_launch
    res = leg.get('resident')
    step_ms, n_step, _, _ = job.run_profiled(a.roofline_steps * leg['tps'])
    if job.touch and n_step:
        # touch mode: the timed region also holds the three preparation launches of every epoch start, so the step kernel's
        # own duration comes from the pass with one event pair per launch (kernels of ~0.5 ms: the events cost nothing)
        avg_ms = step_ms / n_step
    kernel_name = 'mf_index_step_kernel' if getattr(job, 'index', False) else 'mf_touch_step_kernel' if job.touch else 'mf_step_kernel'
    here = source_hash()
    traffic, traffic_src, traffic_note = pmc_traffic_of(f'{a.workload}_s{len(leg["all_sizes"])}_d{d}_b{batch}', kernel_name)
    achieved = alg20 / (avg_ms * 1e-3) / 1e9
    fabric = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    # the ml-1m working set (~45 MB) is cache resident: the step kernel is bound by its three dependent memory levels, not by
    # HBM bytes; only the touch-mode regime (tables beyond the Infinity Cache) is an HBM-bound kernel
    return {'bound': 'hbm' if job.touch else 'latency (cache-resident working set; frac is algorithmic bytes against the HBM peak)',
            'kernel': kernel_name, 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic, 'traffic_source': traffic_src, 'traffic_note': traffic_note,
            'fabric_gbs': round(fabric, 1) if fabric else None, 'fabric_frac': round(fabric / HBM_PEAK_GBS, 4) if fabric else None,
            'alg_bytes_per_launch': round(alg20), 'alg_bytes_per_launch_16B_dense': round(alg16),
            'alg_gbs_16B_dense': round(alg16 / (avg_ms * 1e-3) / 1e9, 1),
            'avg_launch_us': round(avg_ms * 1e3, 2), 'avg_launch_from': 'event pair per launch' if job.touch else 'events around the timed region minus the stream\'s waits for tag chunks (event pairs), / launches',
            'stream_waited_for_tags_ms': round(leg.get('waited_ms', 0.0), 3), 'avg_launch_us_region': round(leg['dev_ms'] / n_launch * 1e3, 2),
            'launches_timed': n_launch,
            'avg_launch_us_resident_tags': round(res['dev_ms'] / max(res['n_launch'], 1) * 1e3, 2) if res else None,
            # (the pass with an event pair around EVERY launch reads higher than region / launches: the two event records a launch is
            # bracketed with take ~1.5-2 us of their own on the stream; it is what splits step launches from epoch-start launches)
            'event_pair_pass_us': round(step_ms / max(n_step, 1) * 1e3, 2),
            **({} if job.touch else latency_floor(job)),
            'dense_rows_streamed_per_shard': rows_streamed if len(rows_streamed) <= 8 else {'shards': len(rows_streamed), 'mean': round(float(np.mean(rows_streamed)), 1)},
            'lazy_rows': bool(job.lazy_rows), 'touch_mode': (3 if getattr(job, 'index', False) else 2 if job.ahead else 1) if job.touch else 0, 'source_hash': here,
            'note': 'frac = algorithmic bytes (SURVEY 8d) / time / 8 TB/s; fabric_frac = PMC bytes that crossed the L2 / time / 8 TB/s '
                    '(ml-1m tables are cache resident, so fabric_frac is the HBM-side utilisation)'}


def pmc_traffic_of(tag, kernel_name):
    """HBM-side bytes per launch of `kernel_name` from the newest committed rocprofv3 PMC file of workload `tag` (tools/pmc_traffic.py ->
    profiles/rNN/) -- quoted only when the file's source hash equals this tree's step-kernel sources.  -> (bytes or None, file, note)."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'*pmc_hbm_traffic*{tag}*.json')) +
                   (glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*pmc_hbm_traffic.json')) if tag == 'ml1m_s5_d32_b30000' else []),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.relpath(f, ROOT))])
    here, note = source_hash(), None
    for f in reversed(cands):
        try:
            with open(f) as fh:
                j = json.load(fh)
            if j.get('source_hash') != here:
                note = note or f'{os.path.relpath(f, ROOT)} was taken at source hash {j.get("source_hash")}, tree is {here}: not quoted'
                continue
            k = [v for n, v in j['kernels'].items() if kernel_name + '<' in n][0]
            return k['traffic_bytes_per_launch'], os.path.relpath(f, ROOT), None
        except Exception:
            continue
    return None, None, note


def latency_floor(job):
    """The cache-resident step kernel (ml-1m: ~45 MB working set) is bound by its chain of dependent memory levels, not by HBM bytes:
    its figure of merit is the distance to a latency floor, built from MI355X_MICROARCH.md's idle-chip constants -- a dependent kernel
    boundary (1.45 us), the grid's dispatch ramp (1,024 workgroups start within 0.34-0.69 us), and the kernel's dependent levels at the
    Infinity-Cache hit latency (227 ns each: shard descriptor -> work unit -> own row + slots -> gathered rows -> the row's store).
    Under load each level takes several times its idle latency (tools/exp_timeline.py: ~10 us per workgroup with all ~1,300 resident):
    that, not bandwidth, is the distance between `avg_launch_us` and the floor."""
    boundary, ramp, level, levels = 1.45, 0.5, 0.227, 5
    floor = boundary + ramp + levels * level
    return {'latency_floor_us': round(floor, 2),
            'latency_floor_terms': {'kernel_boundary_us': boundary, 'dispatch_ramp_us': ramp, 'dependent_levels': levels, 'level_latency_us_idle_chip': level,
                                    'source': 'MI355X_MICROARCH.md (price list: boundary; dispatch; cycle constants: Infinity Cache hit latency)'}}


def cpu_baseline_of(a, leg, d, batch):
    """The torch DataLoader port of the reference (oracle/torch_port.py), timed on this host: a thread
    sweep for the arithmetic, then the reference's structure (per-sample Dataset + DataLoader workers)
    for a bounded number of epochs, and its end-to-end flavour with the two per-epoch tests."""
    from oracle import torch_port
    spec, data = leg['spec'], leg['data']
    from ultrare_amd import synth
    shard_of, _ = synth.uniform_shards(spec['n_user'], len(leg['all_sizes']))
    part0 = synth.split_shards(data['train'], shard_of, len(leg['all_sizes']))[0]
    test0 = synth.split_shards(data['test'], shard_of, len(leg['all_sizes']))[0]
    from ultrare_amd import rng
    ncpu = rng.host_cpus()          # the CPUs this container may really use (cgroup quota), not the host's count
    # ---- arithmetic only (pre-batched tensors): sweep the intra-op thread count, keep the best
    sweep = {}
    for t in [x for x in (1, 4, 8, 16, 32, 64) if x <= ncpu]:
        torch.set_num_threads(t)
        sweep[t] = round(torch_port.prebatched_rate(part0, spec['n_user'], spec['n_item'], d, batch, 1), 1)
    best_t = max(sweep, key=sweep.get)
    torch.set_num_threads(best_t)
    workers = min(24, max(ncpu - best_t, 1))
    budget = a.cpu_budget
    torch.manual_seed(42)
    _, seen, spent, _ = torch_port.train_shard(part0, spec['n_user'], spec['n_item'], d, batch, 50, workers=workers,
                                               budget_s=budget * 0.6)
    torch.manual_seed(42)
    e2e_seen, e2e_spent = torch_port.train_shard_with_tests(part0, test0, spec['n_user'], spec['n_item'], d, batch, 50,
                                                            workers=workers, budget_s=budget * 0.4)
    n0 = len(part0[0])
    return {'value': round(seen / spent, 1), 'unit': 'interactions/s', 'cores': min(ncpu, workers + best_t), 'kind': 'port',
            'threads': best_t, 'workers': workers, 'host_cpus': ncpu, 'machine_cpus': os.cpu_count(),
            'sample': f'shard 0 ({n0} rows), {seen // n0} epoch(s) = {seen} interactions in {spent:.1f}s; '
                      f'per-sample Dataset + DataLoader({workers} worker processes) + nn.Embedding + SGD with {best_t} torch threads',
            'prebatched_value': sweep[best_t], 'prebatched_thread_sweep': sweep,
            'end_to_end_value': round(e2e_seen / e2e_spent, 1),
            'end_to_end_sample': f'the same plus the per-epoch shard test and total test in Python (scratch.py:72-97), '
                                 f'{e2e_seen // n0} epoch(s) in {e2e_spent:.1f}s'}


def unlearn_leg(a, data, shards, d):
    """Second half of the metric: Sisa.learn, then Sisa.unlearn after a 2 % random user deletion, 50
    epochs, through the operator surface; wall clock with per-epoch evaluations, row merge, final test."""
    from ultrare_amd import measure as e2e
    r = e2e.sisa_request(shards, d, 50, 1, 2.0, data=data, reps=6)
    out = {'learn_wall_s': r['learn_s'], 'unlearn_wall_s': r['unlearn_s'], 'learn_wall_s_all': r['learn_s_all'], 'unlearn_wall_s_all': r['unlearn_s_all'],
           'timed': r['timed'], 'epochs': 50,
           'deleted_users': r['deleted_users'], 'retrained_shards': r['retrained_shards'],
           'unlearn_interactions': r['unlearn_interactions'],
           'layouts_built_in_timed_call': r['layouts_built'],
           'includes': 'a NEW request on a warm allocator: its own deletion set and freshly made in-memory loaders, so the timed call '
                       'builds the HBM layouts of the shards it trains and uploads them (layouts_built_in_timed_call), draws the model inits '
                       "and the epochs' seeds from the host RNG streams, makes every epoch's batches (torch.randperm's permutations: on the device, "
                       'csrc/perm_tags.hip), runs 50 epochs of all retrained shards side by side with the per-epoch shard / total evaluations, merges '
                       'the rows, runs the final test and destroys the job (its teardown is joined before the clock stops); the test sets '
                       'stay resident (a deletion does not change them)',
           'batch_tags': 'device' if __import__('ultrare_amd.rng', fromlist=['rng']).device_tags_wanted() else 'host',
           'final_test': {'learn': r['log0'], 'unlearn': r['unlearn_log0']}, 'nan_shards': r['nan_shards']}
    return out, e2e


def ot_leg(a):
    """`ot_cluster` (utils.py:628-656, the `group.py` half of the north-star path) on the driver's clock: n = 6,040 / k = 5 / d = 32 to
    convergence or 10 rounds, labels asserted equal to the reference's own run (tests/golden/ot_ml1m.npz), and ONE round at n = 162,000 /
    k = 32 / d = 128; the CPU oracle (oracle.cpu_ref.ot_cluster: the same algorithm with an exact HiGHS LP where the reference calls POT's
    ot.emd -- SURVEY 8c) timed beside the first within a budget."""
    from ultrare_amd import measure, synth
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ot_ml1m.npz'))
    n, d, seed = int(g['n']), int(g['d']), int(g['seed'])
    out = {'ml1m_k5': measure.ot_request(n, d, 5, seed, 10, want_label=g['k5_label'])}
    out['ml25m_k32_one_round'] = measure.ot_request(162000, 128, 32, 20240608, 1)
    if not a.no_cpu_baseline:
        from oracle import cpu_ref as O
        X = synth.ot_embedding(n, d, seed)
        rounds = 2
        np.random.seed(0)
        np.random.choice(n, int(2 / 100 * n), replace=False)
        t0 = time.perf_counter()
        O.ot_cluster(X, 5, max_iters=rounds)
        spent = time.perf_counter() - t0
        gpu_round = out['ml1m_k5']['wall_s'] / max(out['ml1m_k5']['rounds'], 1)
        out['cpu_baseline'] = {'kind': 'port', 'value': round(spent / rounds, 3), 'unit': 's per round', 'cores': 1,
                               'sample': f'oracle.cpu_ref.ot_cluster, n={n}, k=5, d={d}: {rounds} rounds in {spent:.1f}s (numpy cost matrix, exact LP by HiGHS, numpy centroids)',
                               'gpu_s_per_round': round(gpu_round, 5)}
    return out


def hbm_leg(a, D):
    """The HBM-bound regime on the driver's clock: BASELINE.json configs[3]'s shape on ONE GPU (synthetic 162 k x 60 k, 22.5 M
    train rows, 32 shards side by side, d = 128: 14.5 GB of tables, touch mode), a few timed steps and the roofline of
    mf_touch_step_kernel from an event pair per launch.  -> the `roofline_hbm` object."""
    import argparse
    b = argparse.Namespace(**vars(a))
    from ultrare_amd import synth
    b.workload, b.shards, b.d, b.roofline_steps = 'ml25m', 32, 128, 2
    t0 = time.perf_counter()
    data = synth.make_dataset(**synth.ML25M, seed=synth.SEED)

    def one(d):
        b.d = d
        leg = train_leg(D, 'ml25m', 32, d, a.batch, 3, 1, False, b.roofline_steps + 2, keep_job=True, data=data)
        job = leg['job']
        r = roofline_of(b, leg, d, a.batch)
        # share of the epoch-start preparation (two launches per epoch start at 24-27 steps per epoch) in device time, from a pass
        # with an event pair around every launch that covers whole epochs
        step_ms, n_step, prep_ms, n_prep = job.run_profiled(2 * leg['tps'])
        # does the reference's arithmetic stay finite on this set?  (d = 128: N(0, 1) rows predict +-11 at the start and every shard
        # diverges from epoch 2 on, in the oracle as well -- DESIGN.md 2; the kernel has no data-dependent exit, so its time stands)
        sse = job.epoch_sse_all()
        done = [min(job.done, job.shard_steps[s_]) // job.steps_per_epoch(s_) for s_ in range(len(job.shards))]
        bad = [int(np.flatnonzero(~np.isfinite(sse[s_][:done[s_]]))[0]) for s_ in range(len(done)) if not np.isfinite(sse[s_][:done[s_]]).all()]
        r.update({'finite_tables': not bad, 'shards_diverged': len(bad), 'first_nonfinite_epoch': (min(bad) if bad else None), 'epochs_run': int(max(done))})
        job.close()
        return leg, r, step_ms, n_step, prep_ms, n_prep
    leg, r, step_ms, n_step, prep_ms, n_prep = one(128)
    # the same shape at k = 16, where the tables stay finite (the width a `--group 32` run of the reference would train: config.py:19)
    leg16, r16, s16, ns16, p16, np16 = one(16)
    r['k16'] = {'workload': 'the same 32 shards at k = 16 (finite tables)', 'value': round(leg16['value'], 1), 'unit_value': 'interactions/s',
                'avg_launch_us': r16['avg_launch_us'], 'achieved': r16['achieved'], 'frac': r16['frac'], 'alg_bytes_per_launch': r16['alg_bytes_per_launch'],
                'touch_mode': r16['touch_mode'], 'finite_tables': r16['finite_tables'], 'epochs_run': r16['epochs_run'],
                'prep_share_of_device_time': round(p16 / max(p16 + s16, 1e-9), 4),
                'traffic': r16['traffic'], 'traffic_source': r16['traffic_source'], 'traffic_note': r16['traffic_note'], 'fabric_frac': r16['fabric_frac'],
                'traffic_over_algorithmic': round(r16['traffic'] / r16['alg_bytes_per_launch'], 3) if r16['traffic'] else None}
    r['full_mf'] = fullmf_leg(a, D, data)
    r.update({'workload': 'BASELINE.json configs[3] shape on one GPU: synthetic 162000x60000, 22500000 train rows, 32-shard SISA '
                          '(uniform grouping), d=128, batch=30000, all shards side by side, touch mode',
              'value': round(leg['value'], 1), 'unit_value': 'interactions/s', 'steps': 3, 'warmup': 1,
              'ms_per_step': round(leg['wall'] * 1e3 / 3, 4), 'ticks_per_step': leg['tps'],
              'prep_share_of_device_time': round(prep_ms / max(prep_ms + step_ms, 1e-9), 4), 'prep_launches': int(n_prep),
              'prep_ms_per_epoch_start': round(prep_ms / max(n_prep // 2, 1), 4),
              'leg_wall_s': round(time.perf_counter() - t0, 1)})
    return r


def fullmf_leg(a, D, data):
    """Full MF at the 25 M shape (config.py:182-188 runFull, the stage whose user_mat0.npy every `--group N` run clusters): ONE shard of
    22.5 M rows, d = 128, 750 optimizer steps per epoch -- touch_mode 3, the epoch's slots sorted by step (csrc/mf_index.h).  One warm-up
    epoch, then one epoch with an event pair per launch: the step launch (with the combine launch of split rows), the epoch start."""
    from ultrare_amd import engine, rng
    u, i, r = data['train']
    spec_u, spec_i, n = data['n_user'], data['n_item'], len(u)
    t0 = time.perf_counter()
    torch.manual_seed(42)
    init = tuple(t * 0.3 for t in rng.mf_init(spec_u, spec_i, 128))          # (0.3: predictions of +-1 at the start, the tables stay finite)
    E = 3
    # the epochs' batch tags as Scratch.train gets them: made on the device (rng.epoch_tags_device -> csrc/perm_chain.hip; 22.5 M rows are
    # beyond perm_tags.hip's 2^20), a launch per epoch on the side stream, inside this leg; the host's Fisher-Yates only under URE_DEVICE_TAGS=0
    seeds = rng.epoch_seeds(E, True)
    torch.cuda.synchronize()
    t_sh = time.perf_counter()
    tags = rng.epoch_tags_device(seeds, n, a.batch, engine._device())
    tags_by = 'device (csrc/perm_chain.hip)'
    if tags is None:
        tags, tags_by = rng.epoch_tags(seeds, n, a.batch, threads=min(16, os.cpu_count() or 1)), 'host threads'
    torch.cuda.synchronize()
    t_sh = (time.perf_counter() - t_sh) / E
    sh = engine.ShardData(u, i, (r / 5.0).astype(np.float32), spec_u, spec_i)
    job = engine.TrainJob([sh], [init], [tags], 128, a.batch, E, 1e-3, 0.1, 0.9, 0.95, final_only=True)
    steps = job.steps_per_epoch(0)
    job.run(steps)
    torch.cuda.synchronize()
    step_ms, n_step, prep_ms, n_prep = job.run_profiled(steps)
    rows = job.touch_rows_per_step()[0]
    alg = n / steps * (16 + 16 * 128) + 20 * rows * 128
    us = step_ms / n_step * 1e3
    job.run()
    finite = bool(np.isfinite(job.epoch_sse(0)).all())
    job.close()
    traffic, traffic_src, traffic_note = pmc_traffic_of(f'ml25m_s1_d128_b{a.batch}', 'mf_index_step_kernel' if job.index else 'mf_touch_step_kernel')
    return {'workload': f'full MF, synthetic {spec_u}x{spec_i}, {n} train rows, ONE shard, d=128, batch={a.batch}: {steps} optimizer steps per epoch; '
                        'inits = 0.3 x the reference\'s N(0, 1) draws and ratings / 5 (with N(0, 1) at d = 128 the reference\'s own arithmetic is NaN from epoch 2)',
            'batch_tags': tags_by, 'shuffle_ms_per_epoch': round(t_sh * 1e3, 2),
            'traffic': traffic, 'traffic_source': traffic_src, 'traffic_note': traffic_note,
            'traffic_over_algorithmic': round(traffic / alg, 3) if traffic else None,
            'fabric_frac': round(traffic / us / 1e3 / HBM_PEAK_GBS, 4) if traffic else None,
            'kernel': 'mf_index_step_kernel' if job.index else 'mf_touch_step_kernel', 'touch_mode': 3 if job.index else 1,
            'avg_launch_us': round(us, 2), 'avg_launch_from': 'event pair per launch (step kernel + the combine launch of split rows)',
            'alg_bytes_per_launch': round(alg), 'achieved': round(alg / us / 1e3, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(alg / us / 1e3 / HBM_PEAK_GBS, 4),
            'rows_trained_per_step': round(rows, 1), 'epoch_start_ms': round(prep_ms, 3), 'epoch_start_share_of_device_time': round(prep_ms / (prep_ms + step_ms), 4),
            'value': round(n / ((step_ms + prep_ms) * 1e-3), 1), 'unit_value': 'interactions/s (device time of one epoch incl. its start)',
            'finite_tables': finite, 'leg_wall_s': round(time.perf_counter() - t0, 1)}


def exchange_leg(D, leg, d, reps=5):
    """The path's one collective on the tables just trained: a global job of 5 x world shards, shard j
    owned by rank j // 5; every rank contributes its shards' own user rows + item tables."""
    from ultrare_amd.method.sisa import exchange_tables
    job, spec = leg['job'], leg['spec']
    S = len(leg['mine'])
    ids = list(range(S * D.world))
    owner = [j // S for j in ids]
    dev = torch.device('cuda', torch.cuda.current_device())
    rows = {j: torch.as_tensor(np.asarray(leg['groups'][j % S], dtype=np.int64)).to(dev) for j in ids}
    if job.touch:
        job.run()          # (a job in touch mode keeps its rows valid for their NEXT step: the tables exist at the end of training / at epoch ends only)
    models = {D.rank * S + s: tuple(t.contiguous() for t in job.tables(s)) for s in range(S)}
    times = []
    for _ in range(reps + 1):
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = exchange_tables(models, ids, owner, D.rank, rows, spec['n_item'], d, dev, D.pg)
        torch.cuda.synchronize()
        times.append(D.max(time.perf_counter() - t0))
    peer = (D.rank + 1) % D.world
    ok = bool(torch.isfinite(got[peer * S][1]).all().item()) and got[peer * S][0].shape == (len(leg['groups'][0]), d)
    seg = sum((len(leg['groups'][s]) + spec['n_item']) * d * 4 for s in range(S))
    return {'collective': 'all_gather_into_tensor (padded all-gather-v of own user rows + item tables)', 'backend': D.pg.get_backend(),
            'ms': round(float(np.median(times[1:])) * 1e3, 3), 'first_call_ms': round(times[0] * 1e3, 3),
            'bytes_per_rank': seg, 'bytes_gathered': seg * D.world, 'shards': len(ids), 'received_ok': ok}


def split_legs(a, D, which):
    """BASELINE.json configs[2..4] as ONE job spread over the ranks."""
    out = {}
    short = dict(steps=min(a.steps, 5), warmup=1)
    if 'config2' in which:
        leg = train_leg(D, 'ml1m', 8, 64, a.batch, a.steps, a.warmup, split=True)
        out['config2'] = {'workload': 'ml-1m 8-shard SISA, d=64, shards placed over the ranks', 'value': round(leg['value'], 1),
                          'unit': 'interactions/s', 'scaling': 'strong', 'ms_per_step': round(leg['wall'] * 1e3 / a.steps, 4),
                          'shards_per_rank': [leg['owner'].count(r) for r in range(D.world)]}
    if 'config4' in which:
        from ultrare_amd import synth
        data = synth.make_dataset(**synth.ML1M)
        r, _ = unlearn_leg(a, data, 16, 16)
        out['config4'] = {'workload': 'ml-1m 16-shard learn + unlearn (2 % random deletion), k=16, 50 epochs, through Sisa(parallel) '
                                      'across the ranks: RNG replay, isolated training, one all-gather, merge, final test',
                          'learn_wall_s': D.max(r['learn_wall_s']), 'unlearn_wall_s': D.max(r['unlearn_wall_s']),
                          'retrained_shards': r['retrained_shards'], 'deleted_users': r['deleted_users']}
    if 'config3' in which:
        leg = train_leg(D, 'ml25m', 32, 128, a.batch, short['steps'], short['warmup'], split=True)
        out['config3'] = {'workload': 'synthetic 162k x 60k x 22.5M train rows, 32-shard SISA, d=128, shards placed over the ranks',
                          'value': round(leg['value'], 1), 'unit': 'interactions/s', 'scaling': 'strong', 'steps': short['steps'],
                          'ms_per_step': round(leg['wall'] * 1e3 / short['steps'], 4),
                          'ms_per_launch': round(leg['wall'] * 1e3 / max(short['steps'] * leg['tps'], 1), 4),
                          'shards_per_rank': [leg['owner'].count(r) for r in range(D.world)]}
    return out


def main():
    t_start = time.perf_counter()
    a = parse()
    global NO_RESIDENT
    NO_RESIDENT = bool(a.no_resident)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    # the driver reads ONE JSON line from stdout, and RCCL / gloo / the engine's progress prints write there too:
    # from here on file descriptor 1 is stderr, and the line goes to a private copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    D = Dist(a)
    rank, world = D.rank, D.world
    from ultrare_amd import _native as nv

    a.shards = a.shards or (5 if a.workload == 'ml1m' else 32)
    a.d = a.d or (32 if a.workload == 'ml1m' else 128)
    split = bool(a.split) and D.pg is not None
    leg = train_leg(D, a.workload, a.shards, a.d, a.batch, a.steps, a.warmup, split, a.roofline_steps, keep_job=True)
    spec, job = leg['spec'], leg['job']

    roofline = roofline_of(a, leg, a.d, a.batch) if job is not None and (rank == 0) else None
    if job is not None and rank != 0:
        job.run_profiled(a.roofline_steps * leg['tps'])          # every rank does the same work outside the timed region

    arch = ''
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        nv.lib().ure_device_info(D.local, None, None, buf, 64)
        arch = buf.value.decode()
    except Exception:
        pass
    mode = (f'ONE job, shards placed over {world} ranks (LPT)' if split else
            'all shards of a rank side by side' + (f'; every rank its own job (x{world})' if world > 1 else ''))
    out = {
        'metric': 'training interactions/sec + unlearn retrain wall-time, ml-1m 5-shard SISA' if a.workload == 'ml1m' else f'training interactions/sec, synthetic ml-25m-scale {a.shards}-shard SISA',
        'value': round(leg['value'], 1), 'unit': 'interactions/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(leg['wall'] * 1e3 / a.steps, 4), 'higher_is_better': True, 'scaling': 'strong' if split else 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{a.workload}-shaped synthetic {spec["n_user"]}x{spec["n_item"]}, {spec["n_train"]} train rows, '
                               f'{a.shards}-shard SISA (uniform grouping), d={a.d}, batch={a.batch}, SGD-momentum-L2, {mode}',
                   'shards_per_gpu': len(leg['mine']), 'shard_rows': leg['sizes'], 'ticks_per_step': leg['tps'],
                   'parallelism': f'shards x{world}', 'backend': a.backend if world > 1 else None, 'arch': arch},
        'device_ms_timed': round(leg['dev_ms'], 3), 'interactions_timed': leg['n_total'],
        # `value` times the launches AND the epochs' shuffles (the product's device path, launched inside the region); the same K steps with
        # every tag resident beforehand -- what rounds 1-4 reported as `value` -- stand beside it
        'batch_tags': leg['batch_tags'], 'shuffles_in_timed_region': leg['shuffles_in_timed_region'], 'tag_chunks': leg.get('tag_chunks'),
        'value_resident_tags': round(leg['resident']['n_total'] / leg['resident']['wall'], 1) if leg.get('resident') else None,
        'ms_per_step_resident_tags': round(leg['resident']['wall'] * 1e3 / a.steps, 4) if leg.get('resident') else None,
        'host_rng_prep_s': round(leg['t_rng'], 3), 'host_rng_prep_covers': "model inits and the epochs' seeds (no shuffle is made on the host)",
        'tag_setup_s': round(leg['t_tags'], 4),
        'roofline': roofline, 'cpu_baseline': None, 'unlearn': None,
    }
    printed = threading.Lock()

    def finite(x):      # strict JSON has no NaN: a diverged shard's metric is reported as null (+ nan_shards beside it)
        if isinstance(x, float):
            return x if np.isfinite(x) else None
        if isinstance(x, dict):
            return {k: finite(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [finite(v) for v in x]
        return x

    def emit():
        if rank == 0 and printed.acquire(blocking=False):
            real_stdout.write(json.dumps(finite(out)) + '\n')
            real_stdout.flush()

    if D.pg is not None:
        # ---- N > 1 extras: the exchange over RCCL and the north-star splits.  A watchdog prints the
        # line with what is there and ends the process if a collective never returns.
        def expire():
            out.setdefault('north_star_splits', {})['status'] = f'timeout after {a.extras_timeout:.0f}s'
            emit()
            os._exit(0 if rank == 0 else 3)
        dog = threading.Timer(a.extras_timeout, expire)
        dog.daemon = True
        dog.start()
        # (a leg that raises on this rank costs its own object, not the line; ranks left waiting in a collective are ended by their watchdog)
        def attempt(key, fn):
            try:
                out[key] = fn()
            except BaseException as e:             # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                out[key] = {'error': f'{type(e).__name__}: {e}'[:400]}
        if not split and job is not None:
            attempt('exchange', lambda: exchange_leg(D, leg, a.d))
        job and job.close()
        leg['job'] = leg['shards'] = job = None
        torch.cuda.empty_cache()
        which = [] if a.splits == 'none' else (['config2', 'config4', 'config3'] if a.splits == 'auto' else a.splits.split(','))
        if which:
            attempt('north_star_splits', lambda: split_legs(a, D, which))
        dog.cancel()
    else:
        # ---- N = 1 extras (rank 0 is the only rank): CPU baseline and the unlearn half of the metric
        # (an extra that fails -- a box short of memory, a guard of a measurement tripping -- costs its own object, never the line: the
        # headline and its roofline are already in `out`)
        def guarded(key, fn):
            try:
                out[key] = fn()
            except BaseException as e:             # noqa: BLE001  (KeyboardInterrupt included: the line is what the driver reads)
                import traceback
                traceback.print_exc(file=sys.stderr)
                prev = out.get(key)                # (what the leg had put into the line before it failed stays)
                out[key] = dict(prev if isinstance(prev, dict) else {}, error=f'{type(e).__name__}: {e}'[:400])
        if not a.no_cpu_baseline:
            guarded('cpu_baseline', lambda: cpu_baseline_of(a, leg, a.d, a.batch))
        def unlearn_extras():
            nonlocal job
            job.close()
            leg['job'] = leg['shards'] = job = None
            un, e2e = unlearn_leg(a, leg['data'], a.shards, a.d)
            # BASELINE.json configs[4]: 16 shards (d = the reference's default k = 16), 2 % random deletion
            r16 = e2e.sisa_request(16, 16, 50, 1, 2.0, data=leg['data'], reps=4)
            un['config4_16_shards_k16'] = {'learn_wall_s': r16['learn_s'], 'unlearn_wall_s': r16['unlearn_s'],
                                           'learn_wall_s_all': r16['learn_s_all'], 'unlearn_wall_s_all': r16['unlearn_s_all'],
                                           'retrained_shards': r16['retrained_shards'], 'deleted_users': r16['deleted_users'],
                                           'layouts_built_in_timed_call': r16['layouts_built'],
                                           'final_test': {'learn': r16['log0'], 'unlearn': r16['unlearn_log0']},
                                           # (on this synthetic set some k = 16 shards diverge in the reference's own arithmetic:
                                           # MSELoss(sum), lr 1e-3, users with ~2,800 ratings)
                                           'nan_shards': r16['nan_shards']}
            out['unlearn'] = un
            # the whole learn request on the same clock as the headline: interactions of a 50-epoch learn / its wall time
            out['value_request'] = round(leg['spec']['n_train'] * 50 / un['learn_wall_s'], 1)
            out['value_request_note'] = 'Sisa.learn through the operator surface (layouts, inits, shuffles, 50 epochs, per-epoch tests, merge, final test): interactions / learn_wall_s'
            cb = out.get('cpu_baseline')
            if cb and cb.get('end_to_end_value') and un.get('unlearn_interactions'):
                # both halves of the metric get a stated CPU figure: the unlearn request's interactions at the CPU path's end-to-end rate
                cb['unlearn_wall_s_extrapolated'] = round(un['unlearn_interactions'] / cb['end_to_end_value'], 1)
                cb['unlearn_wall_s_extrapolated_from'] = (f"{un['unlearn_interactions']} interactions of the unlearn request (the retrained shards' rows x 50 epochs) "
                                                          f"/ end_to_end_value; the reference retrains shard after shard on the CPU (sisa.py:66-118)")
            if not a.no_cold and time.perf_counter() - t_start < a.extras_budget:
                # the whole request as the reference's CLI runs it (config.py:139-172), nothing resident beforehand
                c = e2e.cold_request(a.shards, a.d, 50, data=leg['data'])
                un['cold_request_s'] = c['unlearn']['total_s']
                un['cold_request'] = {'unlearn': c['unlearn'], 'learn': c['learn'], 'retrained_shards': c['retrained'],
                                      'deleted_users': c['deleted_users'], 'flow': c['flow'], 'final_test': c['unlearn_log0']}
            if time.perf_counter() - t_start < a.extras_budget:
                # the full-MF stage as a request (config.py:182-188; BASELINE.json configs[0]'s shape): its epochs' shuffles made on the device
                un['run_full'] = e2e.full_request(a.d, 50, data=leg['data'], reps=4)
            return un
        if a.workload == 'ml1m' and not a.no_unlearn:
            guarded('unlearn', unlearn_extras)
        if a.workload == 'ml1m' and not a.no_ot and time.perf_counter() - t_start < a.extras_budget:
            guarded('ot', lambda: ot_leg(a))
        if a.workload == 'ml1m' and not a.no_hbm_leg and time.perf_counter() - t_start < a.extras_budget:
            def hbm():
                nonlocal job
                if job is not None:
                    job.close()
                    leg['job'] = leg['shards'] = job = None
                leg['data'] = None
                torch.cuda.empty_cache()
                return hbm_leg(a, D)
            guarded('roofline_hbm', hbm)
    emit()
    if D.pg is not None:
        D.pg.destroy_process_group()


if __name__ == '__main__':
    main()
