#!/usr/bin/env python3
"""Soak of csrc/perm_chain.hip against the host's tags: random sizes (1 .. 400 k rows, a few beyond 2^20 and 2^21), random seeds and batch sizes,
many permutations per call, every range width -- one process, `--calls` calls.  Prints one line per call and a summary; exit code 1 on a mismatch."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import _native as nv, rng     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--calls', type=int, default=24)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    L = nv.lib()
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(a.seed)
    bad = 0
    for call in range(a.calls):
        big = call % 8 == 7
        n_perms = int(rs.randint(1, 6)) if big else int(rs.randint(1, 120))
        hi = (1 << 21) + 300000 if big else 400000
        sizes = [int(rs.randint(1, hi)) if rs.rand() > 0.1 else int(rs.choice([1, 2, 3, 623, 624, 625, 1023, 1024, 1025, 16383, 16384, 16385])) for _ in range(n_perms)]
        batches = [int(rs.choice([1, 7, 3000, 30000])) for _ in sizes]
        batches = [max(b, -(-n // 65535)) for n, b in zip(sizes, batches)]
        seeds = rs.randint(0, 2 ** 62, size=n_perms).astype(np.int64)
        total = sum(sizes)
        out = torch.full((total,), -1, dtype=torch.int16, device=dev)
        want = torch.empty(total, dtype=torch.int16)
        table, at = [], 0
        for n, b, sd in zip(sizes, batches, seeds):
            nv.check(L.ure_host_randperm_tags(np.array([sd]).ctypes.data, 1, n, b, want.data_ptr() + 2 * at, 0), 'host')
            table.append((int(sd), out.data_ptr() + 2 * at, n, b))
            at += n
        tab = torch.from_numpy(np.array(table, dtype=rng.PERM_DTYPE).view(np.uint8)).to(dev)
        n_max = max(sizes)
        rl = int(rs.choice([0, 10, 11, 12, 14]))
        words = int(L.ure_device_shuffle_tags_scratch(n_max, n_perms))
        scratch = torch.randint(-2 ** 31, 2 ** 31 - 1, (words,), dtype=torch.int32, device=dev)
        flag = int(L.ure_device_shuffle_tags_flag(n_max, n_perms))
        scratch[flag] = 0
        nv.check(L.ure_device_shuffle_tags(tab.data_ptr(), n_perms, n_max, scratch.data_ptr(), words, rl, nv.stream_handle()), 'device')
        torch.cuda.synchronize()
        ok = bool(torch.equal(out.cpu(), want)) and int(scratch[flag]) == 0
        bad += not ok
        print(f'call {call}: {n_perms} permutations, {total} rows, n_max {n_max}, range_log2 {rl}: {"ok" if ok else "MISMATCH"}', flush=True)
    print(f'{a.calls - bad} of {a.calls} calls equal to the host')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
