"""touch_mode 3: the device-built slot index of an epoch against a numpy restatement (tests/test_gpu_touch.py uses check()).
python tools/check_index.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def expected_index(part, sched_host, n_active, n_user, perm, B):
    """-> (step_begin, list per step of (row id, [opposite ids in file order], [ratings]) in schedule order)"""
    u, i, r = (np.asarray(x) for x in part)
    n = len(u)
    step_of = np.empty(n, dtype=np.int64)
    step_of[perm] = np.arange(n) // B
    steps = (n + B - 1) // B
    sched_idx = np.full(int(sched_host[:, 0].max()) + 1, -1, dtype=np.int64)
    sched_idx[sched_host[:n_active, 0]] = np.arange(n_active)
    # every interaction twice: its user row (opposite = item id) and its item row (opposite = user id)
    rows = np.concatenate([u, n_user + i]).astype(np.int64)
    opp = np.concatenate([i, u]).astype(np.int64)
    rat = np.concatenate([r, r]).astype(np.float32)
    j = np.concatenate([np.arange(n), np.arange(n)])
    st = np.concatenate([step_of, step_of])
    order = np.lexsort((j, sched_idx[rows], st))
    return steps, st[order], rows[order], opp[order], rat[order]


def check(job, s, part, perm, B):
    """Compares shard s's index (of the epoch the job last started) with the numpy restatement; returns the number of items."""
    sh = job.shards[s]
    steps, st, rows, opp, rat = expected_index(part, sh._sched_host, sh.n_active, sh.n_user, np.asarray(perm), B)
    sb = job.index_array(s, 'step_begin')
    exp_sb = np.searchsorted(st, np.arange(steps + 1))
    assert np.array_equal(sb, exp_sb), ('step_begin', sb[:8], exp_sb[:8])
    ss = job.index_array(s, 'sslot')
    assert len(ss) == len(st)
    assert np.array_equal(ss[:, 3] & 0xFFFF, st), 'sorted slots: steps'
    assert np.array_equal(ss[:, 0] & 0x7FFFFFFF, opp), 'sorted slots: opposite ids'
    assert np.array_equal(ss[:, 1].view(np.float32), rat), 'sorted slots: ratings'
    assert np.array_equal(ss[:, 2], rows), 'sorted slots: rows'
    # runs
    start = np.ones(len(st), dtype=bool)
    start[1:] = (st[1:] != st[:-1]) | (rows[1:] != rows[:-1])
    begin = np.flatnonzero(start)
    end = np.append(begin[1:], len(st))
    items = job.index_array(s, 'items')
    assert len(items) == len(begin), (len(items), len(begin))
    assert np.array_equal(items[:, 1], begin) and np.array_equal(items[:, 2], end), 'items: runs'
    assert np.array_equal(items[:, 0] & 0x7FFFFFFF, rows[begin]), 'items: rows'
    si = job.index_array(s, 'step_item')
    assert np.array_equal(si, np.searchsorted(st[begin], np.arange(steps + 1))), 'step_item'
    # buffers and gaps: a row alternates between the two buffers with each of its own steps, starting in buffer 0
    irow, istep = rows[begin], st[begin]
    o = np.lexsort((istep, irow))
    rank = np.zeros(len(o), dtype=np.int64)
    same = np.zeros(len(o), dtype=bool)
    same[1:] = irow[o][1:] == irow[o][:-1]
    for k in range(1, len(o)):
        if same[k]:
            rank[k] = rank[k - 1] + 1
    nxt = np.full(len(o), steps, dtype=np.int64)
    nxt[:-1][same[1:]] = istep[o][1:][same[1:]]
    buf = np.empty(len(o), dtype=np.int64)
    gap = np.empty(len(o), dtype=np.int64)
    buf[o] = rank & 1
    gap[o] = nxt - istep[o] - 1
    assert np.array_equal((items[:, 0].view(np.uint32) >> 31).astype(np.int64), buf), 'items: buffers'
    assert np.array_equal(items[:, 3] & 0xFFFF, gap), 'items: gaps'
    # the buffer of a slot's opposite row at that step
    key = irow * (steps + 1) + istep
    srt = np.argsort(key)
    opp_row = np.where(rows >= sh.n_user, opp, sh.n_user + opp)
    at = np.searchsorted(key[srt], opp_row * (steps + 1) + st)
    assert np.array_equal(key[srt][at], opp_row * (steps + 1) + st)
    assert np.array_equal((ss[:, 0] >> 31).astype(np.int64), buf[srt][at]), 'sorted slots: buffers of the opposite rows'
    return len(items)


if __name__ == '__main__':
    from oracle import cpu_ref as O
    from ultrare_amd import engine, rng
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'toy', '0_train.csv')
    raw = O.load_csv(G)
    part = O.partition(*raw, O.uniform_groups(1508, 1))[0]
    k, E = 16, 4
    for B in (437, 3000, 37):       # 65 steps per epoch (two mask words); 10 (the short-epoch sort: chunks of 1,024); 762 (thirteen words)
        print('B', B)
        torch.manual_seed(11)
        init = rng.mf_init(1508, 2071, k)
        perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(part[0]))
        job = engine.TrainJob([engine.ShardData(*part, 1508, 2071)], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch='index')
        steps = job.steps_per_epoch(0)
        for e in range(E):          # the index a shard's steps read is the one of its current epoch (two sets, by epoch parity)
            job.run(1)
            print('epoch', e, 'items', check(job, 0, part, perms[e].numpy(), B))
            job.run(steps - 1)
        job.close()
