#!/usr/bin/env python3
"""HBM-side traffic of the step kernel from rocprofv3 PMC passes, in the form bench.py quotes.

    python tools/pmc_traffic.py OUT_DIR [--tag r02] [-- bench.py arguments ...]

Runs `rocprofv3 --pmc <GROUP> --kernel-trace --output-format csv -- python3 bench.py <args> --no-cpu-baseline
--no-unlearn --steps 5 --warmup 1`, one pass per counter group (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum), never combined with another tracing domain, and writes
    OUT_DIR/<tag>_pmc_hbm_traffic_<workload>_s<shards>_d<d>_b<batch>.json
with per-kernel means, the corrected bytes per launch (gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide
reads -- MI355X_MICROARCH.md, HBM section -- so read bytes = 2 x FETCH_SIZE), the command, and the source hash of
the tree (ultrare_amd.build.source_hash): bench.py quotes `traffic` only when that hash equals the running tree's.
Copy the file into profiles/rNN/ to have it judged.
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GROUPS = ['FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum']


def main():
    argv = sys.argv[1:]
    bench_args = []
    if '--' in argv:
        i = argv.index('--')
        argv, bench_args = argv[:i], argv[i + 1:]
    out_dir = os.path.abspath(argv[0])
    tag = argv[argv.index('--tag') + 1] if '--tag' in argv else 'r02'
    os.makedirs(out_dir, exist_ok=True)
    from ultrare_amd import build

    def opt(name, default):
        return bench_args[bench_args.index(name) + 1] if name in bench_args else default
    workload = opt('--workload', 'ml1m')
    shards = opt('--shards', '5' if workload == 'ml1m' else '32')
    d = opt('--d', '32' if workload == 'ml1m' else '128')
    batch = opt('--batch', '30000')
    tail = ['--no-cpu-baseline', '--no-unlearn', '--no-hbm-leg'] + ([] if '--steps' in bench_args else ['--steps', '5']) + ([] if '--warmup' in bench_args else ['--warmup', '1'])
    cmd_tail = ['python3', os.path.join(ROOT, 'bench.py')] + bench_args + tail
    env = dict(os.environ, TMPDIR='/tmp')
    sums = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for n, group in enumerate(GROUPS):
        pdir = os.path.join(out_dir, f'{tag}_{workload}_s{shards}_d{d}_pass{n}')
        import shutil
        shutil.rmtree(pdir, ignore_errors=True)
        cmd = ['rocprofv3', '--pmc'] + group.split() + ['--kernel-trace', '--output-format', 'csv', '-d', pdir, '--'] + cmd_tail
        print('pass', n, group, flush=True)
        with open(pdir + '.log', 'w') as log:
            rc = subprocess.run(cmd, cwd='/tmp', env=env, stdout=log, stderr=subprocess.STDOUT, timeout=900).returncode
        if rc != 0:
            raise SystemExit(f'pass {n} failed (rc {rc}); see {pdir}.log')
        for path in glob.glob(os.path.join(pdir, '**', '*counter_collection.csv'), recursive=True):
            seen = set()
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = row['Kernel_Name']
                    if 'ure::' not in k:
                        continue
                    k = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
                    c = sums[k][row['Counter_Name']]
                    c[0] += float(row['Counter_Value'])
                    c[1] += 1
                    if row['Dispatch_Id'] not in seen:
                        seen.add(row['Dispatch_Id'])
                        dur[k][0] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
                        dur[k][1] += 1
    kernels = {}
    for k, cs in sums.items():
        e = {}
        for c, (tot, cnt) in cs.items():
            if c in ('FETCH_SIZE', 'WRITE_SIZE'):
                e[c] = {'dispatches': cnt, 'mean_KiB': round(tot / cnt, 2)}
            else:
                e[c] = round(tot / cnt, 2)
        if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
            f, w = e['FETCH_SIZE']['mean_KiB'] * 1024, e['WRITE_SIZE']['mean_KiB'] * 1024
            e['traffic_bytes_per_launch'] = round(2 * f + w)
            e['traffic_bytes_per_launch_uncorrected'] = round(f + w)
        if 'TCC_HIT_sum' in e and 'TCC_MISS_sum' in e:
            e['l2_hit_rate'] = round(e['TCC_HIT_sum'] / max(e['TCC_HIT_sum'] + e['TCC_MISS_sum'], 1), 4)
        e['mean_us_under_profiler'] = round(dur[k][0] / max(dur[k][1], 1), 2)
        kernels[k] = e
    res = {'command': 'python tools/pmc_traffic.py ... -- ' + ' '.join(bench_args) + '   (rocprofv3 --pmc <GROUP> --kernel-trace --output-format csv -- ' +
                      ' '.join(['python3', 'bench.py'] + bench_args + tail) +
                      '; one pass per counter group: ' + ' | '.join(GROUPS) + ')',
           'unit_note': 'FETCH_SIZE / WRITE_SIZE in KiB per dispatch; gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) reads '
                        '(MI355X_MICROARCH.md, HBM section): read bytes = 2 * FETCH_SIZE * 1024',
           'source_hash': build.step_kernel_hash(), 'library_hash': build.source_hash(), 'workload': f'{workload}_s{shards}_d{d}_b{batch}', 'kernels': kernels}
    name = os.path.join(out_dir, f'{tag}_pmc_hbm_traffic_{workload}_s{shards}_d{d}_b{batch}.json')
    with open(name, 'w') as f:
        json.dump(res, f, indent=1)
    print(name)
    print(json.dumps({k: {x: v for x, v in e.items() if x.startswith('traffic') or x in ('l2_hit_rate', 'mean_us_under_profiler')} for k, e in kernels.items()}, indent=1))


if __name__ == '__main__':
    main()
