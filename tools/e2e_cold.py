#!/usr/bin/env python3
"""Cold end-to-end unlearning request at ml-1m size, from CSV files on disk to the final
ensemble test (the reference's Instance.__group path, config.py:123-174, uniform groups):
read + partition with the deletion set, build layouts, retrain the affected shards, merge, test.
Prints one JSON object with the wall time of each phase."""
import argparse, copy, json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shards', type=int, default=5)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=50)
    a = ap.parse_args()
    from ultrare_amd import synth
    from ultrare_amd.method.sisa import Sisa
    from ultrare_amd.read import RatingData, loadData, readRating

    data = synth.make_dataset(**synth.ML1M)
    tmp = tempfile.mkdtemp()
    tr_csv, te_csv = os.path.join(tmp, 'train.csv'), os.path.join(tmp, 'test.csv')
    synth.write_csv(tr_csv, data['train'])
    synth.write_csv(te_csv, data['test'])
    n_user, n_item = data['n_user'], data['n_item']
    del_user = np.random.RandomState(1).choice(n_user, int(0.02 * n_user), replace=False).tolist()

    class P:
        k, lam, seed, batch, lr, lr_decay, momentum, epochs, parallel = a.k, 0.1, 42, 30000, 0.001, 0.95, 0.9, a.epochs, True
    P.n_user, P.n_item = n_user, n_item

    def request(dels, models):
        t = {}
        t0 = time.perf_counter()
        tr, idx = readRating(tr_csv, n_user, 5, dels, [], a.shards, [])
        te, _ = readRating(te_csv, n_user, 5, [], [], a.shards, idx)
        t['read_partition_s'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        trd = [loadData(RatingData(x), P.batch, 24) for x in tr]
        ted = [loadData(RatingData(x), P.batch, 24, False) for x in te]
        tot = loadData(RatingData(np.hstack(te)), P.batch, 24, False)
        t['loaders_s'] = time.perf_counter() - t0
        s = Sisa(P, 'mf', a.shards, idx)
        torch.manual_seed(42)
        t0 = time.perf_counter()
        if models is None:
            ml = s.learn(trd, ted, tot, 0, '')
        else:
            ml = s.unlearn(models, trd, ted, tot, dels, 0, '')
        torch.cuda.synchronize()
        t['train_merge_test_s'] = time.perf_counter() - t0
        t['total_s'] = sum(t.values())
        return ml, s, {k: round(v, 4) for k, v in t.items()}

    request([], None)                                   # warm-up: library load, allocator, pinned pool
    ml, s, t_learn = request([], None)
    ml2, s2, t_un = request(del_user, [copy.deepcopy(m) for m in ml])
    print(json.dumps({'shards': a.shards, 'k': a.k, 'epochs': a.epochs, 'learn': t_learn, 'unlearn': t_un,
                      'retrained': len(s2.retrained), 'log0': s.log0, 'unlearn_log0': s2.log0}))


if __name__ == '__main__':
    main()
