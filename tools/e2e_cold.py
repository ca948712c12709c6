#!/usr/bin/env python3
"""Cold end-to-end unlearning request at ml-1m size, from CSV files on disk to the final
ensemble test (the reference's Instance.__group path, config.py:123-174, uniform groups):
read + partition with the deletion set, build layouts, retrain the affected shards, merge, test.
Prints one JSON object with the wall time of each phase."""
import argparse, copy, json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(shards=5, k=32, epochs=50, data=None):
    """-> dict: phases of a cold learn request and of a cold unlearn request (a fresh 2 % deletion set), after one
    warm-up request that loads the library and warms the device allocator and the pinned pool.  Nothing of a request's
    data survives into the next: the CSV files are read and partitioned again, loaders, HBM layouts and test sets are rebuilt."""
    import shutil
    from ultrare_amd import engine, synth
    from ultrare_amd.method.sisa import Sisa
    from ultrare_amd.read import RatingData, loadData, readRating

    data = data or synth.make_dataset(**synth.ML1M)
    tmp = tempfile.mkdtemp()
    tr_csv, te_csv = os.path.join(tmp, 'train.csv'), os.path.join(tmp, 'test.csv')
    synth.write_csv(tr_csv, data['train'])
    synth.write_csv(te_csv, data['test'])
    n_user, n_item = data['n_user'], data['n_item']
    del_user = np.random.RandomState(1).choice(n_user, int(0.02 * n_user), replace=False).tolist()

    class P:
        lam, seed, batch, lr, lr_decay, momentum, parallel = 0.1, 42, 30000, 0.001, 0.95, 0.9, True
    P.k, P.epochs, P.n_user, P.n_item = k, epochs, n_user, n_item

    def request(dels, models):
        t = {}
        built0 = engine.ShardData.built
        t0 = time.perf_counter()
        tr, idx = readRating(tr_csv, n_user, 5, dels, [], shards, [])
        te, _ = readRating(te_csv, n_user, 5, [], [], shards, idx)
        t['read_partition_s'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        trd = [loadData(RatingData(x), P.batch, 24) for x in tr]
        ted = [loadData(RatingData(x), P.batch, 24, False) for x in te]
        tot = loadData(RatingData(np.hstack(te)), P.batch, 24, False)
        t['loaders_s'] = time.perf_counter() - t0
        s = Sisa(P, 'mf', shards, idx)
        torch.manual_seed(42)
        t0 = time.perf_counter()
        if models is None:
            ml = s.learn(trd, ted, tot, 0, '')
        else:
            ml = s.unlearn(models, trd, ted, tot, dels, 0, '')
        torch.cuda.synchronize()
        t['train_merge_test_s'] = time.perf_counter() - t0
        t['total_s'] = sum(t.values())
        t = {k: round(v, 4) for k, v in t.items()}
        t['layouts_built'] = engine.ShardData.built - built0
        return ml, s, t

    try:
        request([], None)                                   # warm-up: library load, allocator, pinned pool
        ml, s, t_learn = request([], None)
        ml2, s2, t_un = request(del_user, [copy.deepcopy(m) for m in ml])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {'shards': shards, 'k': k, 'epochs': epochs, 'learn': t_learn, 'unlearn': t_un, 'retrained': len(s2.retrained),
            'deleted_users': len(del_user), 'log0': s.log0, 'unlearn_log0': s2.log0,
            'flow': 'config.py:139-172: CSV files on disk -> readRating (partition with the deletion set) -> loaders -> HBM layouts '
                    '(uploaded over PCIe) -> Sisa.learn / unlearn (50 epochs, per-epoch logs) -> merge -> final test'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shards', type=int, default=5)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=50)
    a = ap.parse_args()
    print(json.dumps(measure(a.shards, a.k, a.epochs)))


if __name__ == '__main__':
    main()
