"""The CPU oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle; the HIP path is then
compared with the oracle (tests/test_gpu_*.py) and with the same goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

G = os.path.join(os.path.dirname(__file__), 'golden')
N_USER, N_ITEM = 1508, 2071


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope='module')
def toy():
    tr = O.load_csv(os.path.join(G, 'toy', '0_train.csv'))
    te = O.load_csv(os.path.join(G, 'toy', '0_test.csv'))
    return tr, te


def test_rng_stream_matches_reference():
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    torch.manual_seed(42)
    U0, V0 = O.mf_init(N_USER, N_ITEM, 16)
    assert np.array_equal(U0[:8], g['U0_head'])
    assert np.array_equal(V0[:8], g['V0_head'])
    assert float(U0.astype(np.float64).sum()) == float(g['U0_sum'])
    assert float(V0.astype(np.float64).sum()) == float(g['V0_sum'])
    O.draw_seed()
    perm = O.epoch_perm(O.draw_seed(), int(g['perm0_n']))
    assert np.array_equal(perm[:16], g['perm0_head'])
    chk = int((perm.astype(object) * np.arange(1, len(perm) + 1).astype(object)).sum() % (2 ** 61 - 1))
    assert chk == int(g['perm0_check'])


@pytest.mark.parametrize('E', [1, 3, 50])
def test_full_mf_matches_reference(toy, E):
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    (tu, ti, tr), (eu, ei, er) = toy
    full = [list(range(N_USER))]
    train = O.partition(tu, ti, tr, full)[0]
    test = O.partition(eu, ei, er, full)[0]
    assert len(train[0]) == int(g['train_n']) and len(test[0]) == int(g['test_n'])
    h = O.Hyper(k=16, batch=3000, epochs=E)
    torch.manual_seed(h.seed)
    U, V, log = O.scratch_train(h, N_USER, N_ITEM, train, test)
    assert rel(U, g[f'E{E}_U']) < 2e-6
    assert rel(V, g[f'E{E}_V']) < 2e-6
    np.testing.assert_allclose(log['train_loss'], g[f'E{E}_train_loss'], rtol=2e-6)
    np.testing.assert_allclose(log['test_rmse'], g[f'E{E}_test_rmse'], rtol=2e-6)
    np.testing.assert_allclose(log['test_hr'], g[f'E{E}_test_hr'], rtol=1e-4)
    np.testing.assert_allclose(log['test_ndcg'], g[f'E{E}_test_ndcg'], rtol=1e-4)
    np.testing.assert_allclose([log['test_rmse'][-1], log['test_ndcg'][-1], log['test_hr'][-1]],
                               g[f'E{E}_final_stable'], rtol=1e-4)


def _sisa_setup(toy, S):
    g = np.load(os.path.join(G, 'sisa_toy.npz'))
    (tu, ti, tr), (eu, ei, er) = toy
    idx = O.uniform_groups(N_USER, S)
    for i in range(S):
        assert np.array_equal(np.array(idx[i]), g[f'S{S}_index{i}'])
    return g, idx, (tu, ti, tr), (eu, ei, er)


@pytest.mark.parametrize('S,E', [(3, 2), (4, 3)])
def test_sisa_learn_unlearn_matches_reference(toy, S, E):
    g, idx, (tu, ti, tr), (eu, ei, er) = _sisa_setup(toy, S)
    tag = f'S{S}'
    train_l = O.partition(tu, ti, tr, idx)
    test_l = O.partition(eu, ei, er, idx)
    for i in range(S):
        assert len(train_l[i][0]) == int(g[f'{tag}_ntrain{i}'])
        assert len(test_l[i][0]) == int(g[f'{tag}_ntest{i}'])
    total = O.hstack(test_l)
    h = O.Hyper(k=16, batch=3000, epochs=E)
    torch.manual_seed(h.seed)
    res = O.sisa_learn(h, N_USER, N_ITEM, idx, train_l, test_l, total)
    for i in range(S):
        assert rel(res['models'][i][1], g[f'{tag}_learn_V{i}']) < 2e-6
        assert rel(res['U_pre'][i], g[f'{tag}_learn_Upre{i}']) < 2e-6
    assert rel(res['merged'], g[f'{tag}_learn_Umerged']) < 2e-6
    np.testing.assert_allclose(res['log0'], g[f'{tag}_learn_log0'], rtol=1e-4)
    # D8: the reference log is one dict appended by every shard
    for key in ('train_loss', 'test_rmse', 'total_rmse', 'total_ndcg', 'total_hr', 'test_ndcg', 'test_hr'):
        mine = np.concatenate([np.asarray(l[key]) for l in res['logs']])
        np.testing.assert_allclose(mine, g[f'{tag}_learn_log_{key}'], rtol=1e-4, err_msg=key)

    for name in ('A', 'B'):
        t = f'{tag}_un{name}'
        del_user = g[t + '_del_user'].tolist()
        train_d = O.partition(tu, ti, tr, idx, del_user)
        assert [len(a[0]) for a in train_d] == g[t + '_ntrain'].tolist()
        torch.manual_seed(h.seed)
        un = O.sisa_unlearn(h, N_USER, N_ITEM, idx, res['models'], train_d, test_l, total, del_user)
        assert len(un['retrained']) == int(g[t + '_n_retrained'])
        assert rel(un['merged'], g[t + '_Umerged']) < 2e-6
        for i in range(S):
            assert rel(un['models'][i][1], g[f'{t}_V{i}']) < 2e-6
        np.testing.assert_allclose(un['log0'], g[t + '_log0'], rtol=1e-4)


def test_eval_unit_vectors():
    g = np.load(os.path.join(G, 'eval_vectors.npz'))
    for c in range(int(g['n_cases'])):
        u, r, scores = g[f'c{c}_u'].astype(np.int32), g[f'c{c}_r'], g[f'c{c}_scores']
        S = scores.shape[0]
        acc = np.zeros(scores.shape[1], dtype=np.float32)
        for m in range(S):
            acc = acc + scores[m]
        pred = acc / np.float32(S)
        got = O.eval_from_pred(u, r, pred, int(g[f'c{c}_batch']))
        np.testing.assert_allclose(got, g[f'c{c}_expect'], rtol=1e-6, err_msg=f'case {c}')
    for v, n, want in zip(g['ndcg_in'], g['ndcg_len'], g['ndcg_out']):
        assert O.ndcg_at_k(v[:n]) == want


@pytest.mark.parametrize('k', [4, 5, 7])
def test_ot_cluster_matches_reference(k):
    g = np.load(os.path.join(G, 'ot_toy.npz'))
    X = g['X']
    tag = f'k{k}'
    # cost kernel restatement is bit-exact with numpy's (utils.py:637)
    C0 = X[g[tag + '_cent_idx']]
    assert np.array_equal(O.ot_cost(X, C0).T, g[tag + '_round0_dist'])
    np.random.seed(0)
    np.random.choice(N_USER, int(2 / 100 * N_USER), replace=False)    # config.py:47-49 with D1 fixed
    trace = []
    inertia, label = O.ot_cluster(X, k, trace=trace)
    assert len(trace) == int(g[tag + '_rounds'])
    assert np.array_equal(np.array([t['label'] for t in trace]), g[tag + '_round_labels'])
    assert np.array_equal(trace[-1]['dist'].T, g[tag + '_last_dist'])
    assert np.array_equal(label, g[tag + '_label'])
    assert np.float64(inertia) == g[tag + '_inertia']


def _ml1m_inputs():
    from ultrare_amd import synth
    g = np.load(os.path.join(G, 'ml1m_synth.npz'))
    data = synth.make_dataset(**synth.ML1M, seed=int(g['seed']))
    assert len(data['train'][0]) == int(g['n_train']) and len(data['test'][0]) == int(g['n_test'])
    assert int((data['train'][0] * 7 + data['train'][1]).sum()) == int(g['train_check'])      # same generator output
    return g, data


def _check_rows(g, tag, U, V, tol):
    assert rel(U[g['rows_u']], g[tag + '_U_rows']) < tol and rel(V[g['rows_i']], g[tag + '_V_rows']) < tol
    assert abs(float(np.abs(U.astype(np.float64)).sum()) / float(g[tag + '_U_abs']) - 1) < tol
    assert abs(float(np.abs(V.astype(np.float64)).sum()) / float(g[tag + '_V_abs']) - 1) < tol


def test_ml1m_size_full_mf_matches_reference():
    """BASELINE configs[0] shape (6040 x 3416, 896,914 rows, d=32, B=30,000), one epoch through the
    real reference (tests/golden/ml1m_synth.npz) vs the oracle."""
    from ultrare_amd import synth
    g, data = _ml1m_inputs()
    full = np.zeros(data['n_user'], dtype=np.int64)
    train = synth.split_shards(data['train'], full, 1)[0]
    test = synth.split_shards(data['test'], full, 1)[0]
    h = O.Hyper(k=32, batch=30000, epochs=1)
    torch.manual_seed(h.seed)
    U, V, log = O.scratch_train(h, data['n_user'], data['n_item'], train, test)
    _check_rows(g, 'full', U, V, 5e-6)
    np.testing.assert_allclose(log['train_loss'], g['full_train_loss'], rtol=1e-5)
    np.testing.assert_allclose([log['test_rmse'][0], log['test_ndcg'][0], log['test_hr'][0]], g['full_test'], rtol=1e-4)


@pytest.mark.parametrize('k,balanced', [(4, False), (4, True), (5, False), (5, True)])
def test_kmeans_oracle_vs_reference_golden(k, balanced):
    """The comparison clusterers (utils.py:354-418) as run by the real reference on the toy user
    embedding (tests/golden/make_golden.py kmeans): labels of every init and of the best-of-3 run,
    inertia bit for bit."""
    g = np.load(os.path.join(G, 'kmeans_toy.npz'))
    tag = f'k{k}_{"bal" if balanced else "plain"}'
    for t in range(3):
        label, inertia = O.single_kmeans(k, g['X'], balanced, 10, g[tag + '_inits'][t])
        assert np.array_equal(label, g[tag + '_single_labels'][t])
        assert inertia == g[tag + '_single_inertia'][t]
    np.random.seed(7)
    assert np.array_equal(O.kmeans(k, g['X'], balanced, 3, 10), g[tag + '_label'])
