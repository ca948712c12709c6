#!/usr/bin/env python3
"""What the host of a GPU box gives this process: CPUs (count, affinity, cgroup quota) and how the permutation
expander (ure_host_randperm: 250 permutations of 180 k rows = one 5-shard, 50-epoch SISA call) scales with threads.
Prints one JSON object.  No GPU needed."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = {'os_cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0))}
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
        try:
            out[f] = open(f).read().strip()
        except OSError:
            pass
    try:
        out['loadavg'] = open('/proc/loadavg').read().split()[:3]
        model = [l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')]
        out['cpu_model'], out['cpuinfo_cpus'] = model[0], len(model)
    except Exception:
        pass
    import torch
    from ultrare_amd import _native as nv
    L = nv.lib()
    n, P = 180000, 250
    seeds = np.arange(1, P + 1, dtype=np.int64)
    buf = torch.empty(P, n, dtype=torch.int32)
    scale = {}
    for t in (1, 2, 4, 8, 16, 32, 64, 128):
        if t > 2 * out['affinity']:
            break
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            nv.check(L.ure_host_randperm(seeds.ctypes.data, P, n, buf.data_ptr(), t), 'randperm')
            best = min(best, time.perf_counter() - t0)
        scale[t] = round(best * 1e3, 2)
    out['randperm_250x180k_ms_by_threads'] = scale
    print(json.dumps(out))


if __name__ == '__main__':
    main()
