#!/usr/bin/env python3
"""OT grouping at BASELINE sizes on the GPU: per-round times of the cost kernel, the potentials (warm start), the
device-to-host copy and the exact host solver, for n = 6040 (d = 32; k = 5, 8, 16) and n = 162,000 (d = 128, k = 32),
with the cold solver beside it on the first round (labels must be equal).  Prints one JSON object; run it under
`rocprofv3 --kernel-trace --stats` for the kernels' own times.

    python tools/profile_ot.py [--rounds 3] [--no-25m]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ot_embedding(n, d, seed):
    rs = np.random.RandomState(seed)
    centers = rs.standard_normal((12, d)) * 0.8
    which = rs.randint(0, 12, n)
    X = centers[which] + rs.standard_normal((n, d)) * 0.6
    return X.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--no-25m', action='store_true')
    a = ap.parse_args()
    from ultrare_amd import _native as nv
    from ultrare_amd.method.utils import ot_warm_iters
    L, st = nv.lib(), nv.stream_handle()
    out = {}
    cases = [(6040, 32, 5, 20240607), (6040, 32, 8, 20240607), (6040, 32, 16, 20240607)]
    if not a.no_25m:
        cases.append((162000, 128, 32, 20240608))
    for n, d, k, seed in cases:
        X = ot_embedding(n, d, seed)
        np.random.seed(0)
        np.random.choice(n, int(2 / 100 * n), replace=False)
        centroid = X[np.random.choice(n, size=k, replace=False)]
        Xd = torch.from_numpy(X).cuda()
        dist_d = torch.empty(k, n, dtype=torch.float32, device='cuda')
        label_d = torch.empty(n, dtype=torch.int32, device='cuda')
        cent_d = torch.empty(k, d, dtype=torch.float32, device='cuda')
        counts_d = torch.empty(k, dtype=torch.int32, device='cuda')
        rounds = []
        pi = np.zeros(k)
        for r in range(a.rounds):
            cd = torch.from_numpy(np.ascontiguousarray(centroid)).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nv.check(L.ure_ot_cost(nv.ptr(Xd), nv.ptr(cd), n, k, d, nv.ptr(dist_d), st), 'cost')
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mis = ctypes.c_int64()
            nv.check(L.ure_ot_potentials(nv.ptr(dist_d), n, k, ot_warm_iters(n), pi.ctypes.data, ctypes.byref(mis), st), 'pot')
            t2 = time.perf_counter()
            dist = dist_d.cpu().numpy()
            t3 = time.perf_counter()
            label, _, obj, aug = nv.ot_assign_warm(dist, pi, want_plan=False)
            t4 = time.perf_counter()
            order = torch.from_numpy(np.argsort(label.astype(np.uint8 if k <= 256 else np.int64), kind='stable').astype(np.int32)).cuda()     # (as method/utils.py: a radix sort)
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(label, minlength=k))]).astype(np.int64)).cuda()
            nv.check(L.ure_ot_centroids_members(nv.ptr(Xd), nv.ptr(order), nv.ptr(off), n, k, d, nv.ptr(cent_d), nv.ptr(counts_d), st), 'cent')
            centroid = cent_d.cpu().numpy()
            t5 = time.perf_counter()
            e = {'cost_ms': round((t1 - t0) * 1e3, 3), 'potentials_ms': round((t2 - t1) * 1e3, 3), 'd2h_ms': round((t3 - t2) * 1e3, 3),
                 'solver_ms': round((t4 - t3) * 1e3, 3), 'augmentations': int(aug), 'misplaced_after_ascent': int(mis.value), 'centroids_ms': round((t5 - t4) * 1e3, 3),
                 'round_ms': round((t5 - t0) * 1e3, 3)}
            if r == 0:
                t = time.perf_counter()
                cold, _, obj0 = nv.ot_assign(dist)
                e['cold_solver_ms'] = round((time.perf_counter() - t) * 1e3, 1)
                e['labels_equal_cold'] = bool(np.array_equal(cold, label)) and obj0 == obj
            rounds.append(e)
        out[f'n{n}_d{d}_k{k}'] = rounds
    print(json.dumps(out))


if __name__ == '__main__':
    main()
