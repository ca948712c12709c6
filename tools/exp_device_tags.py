#!/usr/bin/env python3
"""ure_device_randperm_tags against ure_host_randperm_tags (bit for bit) and its time: python tools/exp_device_tags.py [--n 180000] [--perms 250] [--groups 16]"""
import ctypes, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd import _native as nv

args = sys.argv[1:]
opt = lambda k, d: type(d)(args[args.index(k) + 1]) if k in args else d
n, P, B = opt('--n', 180000), opt('--perms', 250), opt('--batch', 30000)
L = nv.lib()
dev = torch.device('cuda:0')
seeds = np.random.RandomState(1).randint(0, 2 ** 62, size=P).astype(np.int64)
host = torch.empty(P, n, dtype=torch.int16)
t0 = time.perf_counter()
nv.check(L.ure_host_randperm_tags(seeds.ctypes.data, P, n, B, host.data_ptr(), 16), 'host')
t_host = time.perf_counter() - t0
from ultrare_amd import rng
out = {'n': n, 'perms': P, 'batch': B, 'host_ms_16_threads': round(t_host * 1e3, 2)}
for groups in [int(g) for g in opt('--groups', '16,32,64').split(',')]:
    words = int(L.ure_device_randperm_tags_scratch(n, groups))
    scratch = torch.zeros(words, dtype=torch.int32, device=dev)
    tags = torch.zeros(P, n, dtype=torch.int16, device=dev)
    table = np.zeros(P, dtype=rng.PERM_DTYPE)
    table['seed'], table['n'], table['batch'] = seeds, n, B
    table['tags'] = tags.data_ptr() + 2 * n * np.arange(P, dtype=np.int64)
    table_d = torch.from_numpy(table.view(np.uint8)).to(dev)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nv.check(L.ure_device_randperm_tags(table_d.data_ptr(), P, n, scratch.data_ptr(), words, groups, nv.stream_handle()), 'device')
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    flags = scratch[2 * ((n + 63) // 64 * 64) * groups:][:groups].cpu().tolist()
    same = bool(torch.equal(tags.cpu(), host))
    out[f'groups_{groups}'] = {'ms': round(min(ts) * 1e3, 3), 'equal_to_host': same, 'gave_up_flags': [f for f in flags if f]}
    if not same:
        d = (tags.cpu() != host)
        out[f'groups_{groups}']['mismatching_perms'] = int(d.any(dim=1).sum())
print(json.dumps(out))
