#!/usr/bin/env python3
"""Times the two device shuffles (csrc/perm_tags.hip: one workgroup per permutation; csrc/perm_chain.hip: many) on the shapes a request
makes: the first epoch of 5 shards, a 5-shard x 50-epoch request, full MF at ml-1m size (50 epochs of 896,914 rows), one epoch of 22.5 M rows.

    python tools/exp_shuffle.py [--reps 5] > exp_shuffle.json          (under rocprofv3 --kernel-trace --stats for the kernels' shares)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import _native as nv, rng     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--skip-big', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--rl', type=int, default=0, help='range_log2 of ure_device_shuffle_tags (0: the library chooses)')
    ap.add_argument('--which', default='chain,reservations')
    a = ap.parse_args()
    L = nv.lib()
    dev = torch.device('cuda:0')
    sizes = [179718, 184837, 171049, 170284, 191026]
    shapes = {'first_epoch_5x180k': [(n, 30000) for n in sizes],
              'request_5x50x180k': [(n, 30000) for n in sizes for _ in range(50)],
              'chunk_5x8x180k': [(n, 30000) for n in sizes for _ in range(8)],
              'config4_16x50x56k': [(56057, 30000)] * 800,
              'fullmf_ml1m_50x897k': [(896914, 30000)] * 50,
              'fullmf_ml1m_1x897k': [(896914, 30000)]}
    if not a.skip_big:
        shapes['one_epoch_22.5M'] = [(22_500_000, 30000)]
        shapes['one_epoch_4M'] = [(4_000_000, 30000)]
    res = {}
    for name, perms in shapes.items():
        if a.only and name not in a.only.split(','):
            continue
        n_max = max(n for n, _ in perms)
        outs = torch.empty(sum(n for n, _ in perms), dtype=torch.int16, device=dev)
        at, table = 0, []
        for i, (n, b) in enumerate(perms):
            table.append((1000 + i, outs.data_ptr() + 2 * at, n, b))
            at += n
        tab = torch.from_numpy(np.array(table, dtype=rng.PERM_DTYPE).view(np.uint8)).to(dev)
        row = {'perms': len(perms), 'rows': at}
        for which in ('chain', 'reservations'):
            if which not in a.which.split(','):
                continue
            if which == 'reservations' and n_max > (1 << 20):
                continue
            if which == 'chain':
                words = int(L.ure_device_shuffle_tags_scratch(n_max, len(perms)))
            else:
                groups = min(256, len(perms))
                words = int(L.ure_device_randperm_tags_scratch(n_max, groups))
            scratch = torch.zeros(words, dtype=torch.int32, device=dev)
            ts = []
            for r in range(a.reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if which == 'chain':
                    nv.check(L.ure_device_shuffle_tags(tab.data_ptr(), len(perms), n_max, scratch.data_ptr(), words, a.rl, nv.stream_handle()), 'shuffle')
                else:
                    nv.check(L.ure_device_randperm_tags(tab.data_ptr(), len(perms), n_max, scratch.data_ptr(), words, groups, nv.stream_handle()), 'randperm')
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
            row[which + '_ms'] = round(float(np.median(ts)), 4)
            row[which + '_ms_all'] = [round(t, 4) for t in ts]
            if which == 'chain':
                keep = outs.clone()
            elif 'chain' in a.which.split(','):
                row['equal'] = bool(torch.equal(keep, outs))
            del scratch
        res[name] = row
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
