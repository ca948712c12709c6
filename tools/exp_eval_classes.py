"""Microbenchmark: ure_eval_users on test sets whose users all have the same number of entries -- what a wave of each class of
the ranking launch costs (quarter waves: <= 16 entries, in-register: <= 64, multi: <= 512, memory: > 512)."""
import json
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ultrare_amd import engine, _native as nv


def run(cnt, users, reps=20):
    uid = np.repeat(np.arange(users, dtype=np.int32), cnt)
    rng = np.random.default_rng(cnt)
    iid = rng.integers(0, 3000, len(uid)).astype(np.int32)
    rating = (rng.integers(1, 6, len(uid)) / 5).astype(np.float32)
    es = engine.EvalSet(uid, iid, rating)
    es.pred.copy_(torch.from_numpy(rng.standard_normal(len(uid)).astype(np.float32)))
    L, st = nv.lib(), nv.stream_handle()
    call = lambda: nv.check(L.ure_eval_users(nv.ptr(es.off), es.n_users, nv.ptr(es.pred), nv.ptr(es.rating), nv.ptr(es.log2), nv.ptr(es.hits),
                                             nv.ptr(es.ndcg), nv.ptr(es.top_rating), es.n_wide, es.n_half, st), 'ure_eval_users')
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    waves = es.n_wide + (es.n_half + 1) // 2 + (es.n_users - es.n_wide - es.n_half + 3) // 4
    us = a.elapsed_time(b) * 1e3 / reps
    return dict(cnt=cnt, users=users, waves=waves, us_per_call=round(us, 2), ns_per_wave_x_1024=round(us * 1e3 / waves * 1024, 1))


if __name__ == '__main__':
    out = []
    for cnt, users in ((4, 120000), (16, 120000), (17, 30000), (32, 30000), (64, 30000), (65, 30000), (128, 30000), (343, 30000), (600, 8000),
                       (17, 300), (64, 300), (128, 300), (343, 300)):
        out.append(run(cnt, users))
        print(json.dumps(out[-1]), flush=True)
