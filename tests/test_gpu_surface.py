"""GPU parity at the reference's operator surface: Scratch.train, Sisa.learn/unlearn,
baseTest, ot_cluster, Instance.runFull/runGroup -- written like the calls the
reference's own config.py makes, compared with goldens produced by the real reference
(tests/golden/make_golden.py) and with the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TRAIN, TEST = os.path.join(G, 'toy', '0_train.csv'), os.path.join(G, 'toy', '0_test.csv')
N_USER, N_ITEM = 1508, 2071
RTOL = 1e-4     # BASELINE.json: 1e-4 relative on learned embeddings and metrics


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / np.abs(b).max())


class Param:
    """The InsParam fields Scratch / Sisa read (config.py:17-49)."""

    def __init__(self, epochs, k=16, batch=3000, parallel=False):
        self.k, self.lam, self.seed, self.batch = k, 0.1, 42, batch
        self.lr, self.lr_decay, self.momentum, self.epochs = 0.001, 0.95, 0.9, epochs
        self.n_user, self.n_item, self.parallel = N_USER, N_ITEM, parallel


def _loaders(train_arr, test_arr, batch):
    from ultrare_amd.read import RatingData, loadData
    return loadData(RatingData(train_arr), batch, 24), loadData(RatingData(test_arr), batch, 24, False)


@pytest.mark.parametrize('E', [3, 50])
def test_scratch_train_matches_reference(E, tmp_path):
    from ultrare_amd.method.scratch import Scratch
    from ultrare_amd.read import readRating
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    tr, idx = readRating(TRAIN, N_USER, 5, [], [], 1, [])
    te, _ = readRating(TEST, N_USER, 5, [], [], 1, idx)
    train, test = _loaders(tr[0], te[0], 3000)
    sc = Scratch(Param(E), 'mf')
    torch.manual_seed(42)
    model = sc.train(train, test, [], 0, str(tmp_path))
    assert rel(model.user_mat.weight, g[f'E{E}_U']) < RTOL
    assert rel(model.item_mat.weight, g[f'E{E}_V']) < RTOL
    np.testing.assert_allclose(sc.log['train_loss'], g[f'E{E}_train_loss'], rtol=RTOL)
    np.testing.assert_allclose(sc.log['test_rmse'], g[f'E{E}_test_rmse'], rtol=RTOL)
    np.testing.assert_allclose(sc.log['test_hr'], g[f'E{E}_test_hr'], rtol=RTOL)
    np.testing.assert_allclose(sc.log['test_ndcg'], g[f'E{E}_test_ndcg'], rtol=RTOL)
    # artifacts (scratch.py:131-144)
    sd = torch.load(tmp_path / 'model0.pth', map_location='cpu')
    assert sorted(sd) == ['item_mat.weight', 'user_mat.weight']
    assert np.load(tmp_path / 'user_mat0.npy').shape == (N_USER, 16)
    assert np.load(tmp_path / 'item_mat0.npy').shape == (N_ITEM, 16)
    log = np.load(tmp_path / 'log0.npy', allow_pickle=True).item()
    assert len(log['train_loss']) == E and len(log['time']) == E


def _sisa_inputs(S, del_user=()):
    from ultrare_amd.read import RatingData, loadData, readRating
    tr, idx = readRating(TRAIN, N_USER, 5, list(del_user), [], S, [])
    te, _ = readRating(TEST, N_USER, 5, [], [], S, idx)
    trd = [loadData(RatingData(a), 3000, 24) for a in tr]
    ted = [loadData(RatingData(a), 3000, 24, False) for a in te]
    tot = loadData(RatingData(np.hstack(te)), 3000, 24, False)
    return idx, trd, ted, tot


@pytest.mark.parametrize('S,E', [(3, 2), (4, 3)])
@pytest.mark.parametrize('parallel', [False, True])
def test_sisa_learn_unlearn_matches_reference(S, E, parallel, tmp_path):
    check_sisa_against_reference(S, E, parallel, tmp_path)


def check_sisa_against_reference(S, E, parallel, tmp_path):
    """Sisa.learn and two Sisa.unlearn calls against the goldens of the real reference (sisa.py:25-118); also run by
    tests/test_gpu_surface_touch.py with the engine forced into its touch modes."""
    import copy
    from ultrare_amd.method.sisa import Sisa
    g = np.load(os.path.join(G, 'sisa_toy.npz'))
    tag = f'S{S}'
    idx, trd, ted, tot = _sisa_inputs(S)
    for i in range(S):
        assert np.array_equal(np.array(idx[i]), g[f'{tag}_index{i}'])
        assert len(trd[i].dataset) == int(g[f'{tag}_ntrain{i}'])
    sisa = Sisa(Param(E, parallel=parallel), 'mf', S, idx)
    torch.manual_seed(42)
    ml = sisa.learn(trd, ted, tot, 0, str(tmp_path))
    for i in range(S):
        assert rel(ml[i].item_mat.weight, g[f'{tag}_learn_V{i}']) < RTOL
        assert rel(np.load(tmp_path / f'user_mat{i + 1}.npy'), g[f'{tag}_learn_Upre{i}']) < RTOL
        assert ml[i].user_mat.weight.data_ptr() == ml[0].user_mat.weight.data_ptr()      # shared merged matrix
    assert rel(ml[0].user_mat.weight, g[f'{tag}_learn_Umerged']) < RTOL
    log0 = np.load(tmp_path / 'log0.npy', allow_pickle=True).item()
    np.testing.assert_allclose([log0['total_rmse'], log0['total_ndcg'], log0['total_hr']], g[f'{tag}_learn_log0'], rtol=RTOL)
    # per-epoch series (scratch.py:83-128): in-loop in the sequential mode, rebuilt from snapshots
    # in the parallel mode
    for key in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr'):
        np.testing.assert_allclose(sisa.log[key], g[f'{tag}_learn_log_{key}'], rtol=RTOL, err_msg=key)
    assert len(np.load(tmp_path / f'log{S}.npy', allow_pickle=True).item()['total_hr']) == S * E      # D8

    for name in ('A', 'B'):
        t = f'{tag}_un{name}'
        del_user = g[t + '_del_user'].tolist()
        idx2, trd2, ted2, tot2 = _sisa_inputs(S, del_user)
        assert [len(d.dataset) for d in trd2] == g[t + '_ntrain'].tolist()
        s2 = Sisa(Param(E, parallel=parallel), 'mf', S, idx2)
        out = tmp_path / name
        out.mkdir()
        torch.manual_seed(42)
        ml2 = s2.unlearn([copy.deepcopy(m) for m in ml], trd2, ted2, tot2, del_user, 0, str(out))
        assert len(s2.retrained) == int(g[t + '_n_retrained'])
        assert rel(ml2[0].user_mat.weight, g[t + '_Umerged']) < RTOL
        for i in range(S):
            assert rel(ml2[i].item_mat.weight, g[f'{t}_V{i}']) < RTOL
        l0 = np.load(out / 'log0.npy', allow_pickle=True).item()
        np.testing.assert_allclose([l0['total_rmse'], l0['total_ndcg'], l0['total_hr']], g[t + '_log0'], rtol=RTOL)
        for key in ('train_loss', 'test_rmse', 'total_rmse', 'total_ndcg', 'total_hr'):
            np.testing.assert_allclose(s2.log[key], g[f'{t}_log_{key}'], rtol=RTOL, err_msg=key)


def test_parallel_equals_sequential_bitwise(tmp_path):
    from ultrare_amd.method.sisa import Sisa
    idx, trd, ted, tot = _sisa_inputs(3)
    res = []
    for par in (False, True):
        s = Sisa(Param(2, parallel=par), 'mf', 3, idx)
        torch.manual_seed(42)
        ml = s.learn(trd, ted, tot, 0, '')
        res.append(([m.item_mat.weight.detach().cpu() for m in ml], ml[0].user_mat.weight.detach().cpu(), s.log0))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]


def test_basetest_and_forward_on_reference_models():
    from ultrare_amd.method.utils import MF, baseTest
    from ultrare_amd.read import RatingData, loadData, readRating
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    te, _ = readRating(TEST, N_USER, 5, [], [], 1, [])
    loader = loadData(RatingData(te[0]), 3000, 24, False)
    model = MF.from_tables(torch.from_numpy(g['E50_U']).cuda(), torch.from_numpy(g['E50_V']).cuda())
    np.testing.assert_allclose(baseTest(loader, [model], None, 'cuda', 0), g['E50_final_stable'], rtol=RTOL)
    u = torch.from_numpy(te[0][0][:100].astype(np.int64))
    i = torch.from_numpy(te[0][1][:100].astype(np.int64))
    want = (g['E50_U'][u.numpy()] * g['E50_V'][i.numpy()]).sum(1)
    assert rel(model(u, i), want) < 1e-5


@pytest.mark.parametrize('k', [4, 5, 7])
def test_ot_cluster_bit_exact(k):
    """Group assignments are bit-exact (BASELINE.json north_star) with the reference's
    ot_cluster driven by an exact LP, under the CLI's numpy RNG sequence."""
    from ultrare_amd.method.utils import ot_cluster
    g = np.load(os.path.join(G, 'ot_toy.npz'))
    np.random.seed(0)
    np.random.choice(N_USER, int(2 / 100 * N_USER), replace=False)
    inertia, label = ot_cluster(g['X'], k)
    assert np.array_equal(label, g[f'k{k}_label'])
    assert np.float64(inertia) == g[f'k{k}_inertia']


@pytest.mark.parametrize('n,k,d', [(1508, 5, 16), (3000, 8, 32), (1000, 3, 20), (777, 6, 128), (500, 2, 200)])
def test_ot_kernels_bit_exact_vs_oracle(n, k, d):
    import ctypes
    from ultrare_amd import _native as nv
    rs = np.random.RandomState(n + d)
    X = rs.standard_normal((n, d)).astype(np.float32)
    C = X[rs.choice(n, k, replace=False)]
    Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(C).cuda()
    dist = torch.empty(k, n, dtype=torch.float32, device='cuda')
    nv.check(nv.lib().ure_ot_cost(nv.ptr(Xd), nv.ptr(Cd), n, k, d, nv.ptr(dist), nv.stream_handle()), 'cost')
    want = ((X - C[:, np.newaxis]) ** 2).sum(axis=2)           # the reference expression itself (utils.py:637)
    assert np.array_equal(dist.cpu().numpy(), want)
    label = rs.randint(0, k, n).astype(np.int32)
    cent = torch.empty(k, d, dtype=torch.float32, device='cuda')
    cnt = torch.empty(k, dtype=torch.int32, device='cuda')
    nv.check(nv.lib().ure_ot_centroids(nv.ptr(Xd), nv.ptr(torch.from_numpy(label).cuda()), n, k, d, nv.ptr(cent),
                                       nv.ptr(cnt), nv.stream_handle()), 'centroids')
    want_c = np.array([X[label == i].mean(axis=0) for i in range(k)])  # utils.py:648
    assert np.array_equal(cent.cpu().numpy(), want_c)
    assert np.array_equal(cnt.cpu().numpy(), np.bincount(label, minlength=k))


def test_instance_run_full_then_group_end_to_end(tmp_path):
    """main.py --group 0 then --group 3 on the toy set: artifact tree of config.py:62-77,
    scratch.py:131-144, sisa.py:23 and the OT label cache of group.py:61-64."""
    import shutil
    from ultrare_amd.config import InsParam, Instance
    data = tmp_path / 'data'
    (data / 'toy').mkdir(parents=True)
    shutil.copy(TRAIN, data / 'toy' / '0_train.csv')
    shutil.copy(TEST, data / 'toy' / '0_test.csv')
    save = tmp_path / 'result'
    torch.manual_seed(42)
    p0 = InsParam('toy', 3, 24, [32], 0, 2, 'rand', data_dir=str(data))
    assert len(p0.del_user) == int(0.02 * N_USER)                        # D1
    Instance(p0, save_dir=str(save)).runFull(is_save=True, verbose=0)
    base = save / '2' / 'rand' / 'toy_g0'
    for f in ('param.pkl', 'deletion.npy', 'MF_full_train/user_mat0.npy', 'MF_full_train/model0.pth',
              'MF_retrain/item_mat0.npy', 'MF_retrain/log0.npy'):
        assert (base / f).exists(), f
    p3 = InsParam('toy', 2, 24, [32], 3, 2, 'rand', data_dir=str(data))
    ins = Instance(p3, save_dir=str(save))
    models = ins.runGroup(is_save=True, learn_type='sisa', group_type='emb-ot', n_group=3, verbose=0)
    assert len(models) == 3
    g3 = save / '2' / 'rand' / 'toy_g3'
    for f in ('MF_emb-ot_sisa_learn/log0.npy', 'MF_emb-ot_sisa_learn/user_mat3.npy', 'MF_emb-ot_sisa_unlearn/log0.npy'):
        assert (g3 / f).exists(), f
    groups = np.load(data / 'toy' / 'val' / 'emb-ot3.npy', allow_pickle=True)
    sizes = sorted(len(x) for x in groups)
    assert sum(sizes) == N_USER and sizes[-1] - sizes[0] <= 1              # balanced
    # the grouping equals the oracle's ot_cluster on the same embedding and numpy state
    X = np.load(base / 'MF_full_train' / 'user_mat0.npy')
    np.random.seed(0)
    np.random.choice(N_USER, int(0.02 * N_USER), replace=False)
    _, lab = O.ot_cluster(X, 3)
    assert [np.flatnonzero(lab == c).tolist() for c in range(3)] == [list(x) for x in groups]
    # shards are ordered ascending by rating count (read.py:45-50) and every deleted user's shard retrained
    assert ins.last.retrained == sorted({i for i, gidx in enumerate(ins.last.group_index)
                                         if set(gidx) & set(int(u) for u in p3.del_user)})


@pytest.mark.gpu
@pytest.mark.parametrize('k,balanced', [(4, False), (4, True), (5, False), (5, True)])
def test_kmeans_vs_reference_golden(k, balanced):
    """utils.py:354-418 through the package's kmeans / singleKmeans (HIP distances and centroids, host
    assignment) against the real reference's labels on the toy user embedding."""
    from scipy.sparse import csr_matrix
    from ultrare_amd.method.utils import kmeans, singleKmeans
    g = np.load(os.path.join(G, 'kmeans_toy.npz'))
    tag = f'k{k}_{"bal" if balanced else "plain"}'
    X = g['X']
    np.random.seed(7)
    for t in range(3):
        label, inertia = singleKmeans(k, len(X), csr_matrix(X), balanced, 10)
        assert np.array_equal(label, g[tag + '_single_labels'][t])
        assert inertia == g[tag + '_single_inertia'][t]
    np.random.seed(7)
    assert np.array_equal(kmeans(k, len(X), X, balanced=balanced, n_init=3, max_iter=10), g[tag + '_label'])


@pytest.mark.gpu
@pytest.mark.parametrize('n,k,d,balanced', [(300, 3, 200, True), (1000, 7, 33, False), (640, 4, 8, True)])
def test_kmeans_kernels_vs_oracle_random(n, k, d, balanced):
    """ure_kmeans_cost / ure_kmeans_centroids / ure_host_kmeans_assign on random data of odd shapes (any d, e.g. a
    rating matrix as embedding) against the numpy statements of the oracle, one full singleKmeans run."""
    from ultrare_amd.method.utils import singleKmeans
    rs = np.random.RandomState(n + d)
    X = (rs.randn(n, d) * (rs.rand(n, d) < 0.6)).astype(np.float32)        # sparse-ish, like ratings
    np.random.seed(3)
    want_label, want_inertia = O.single_kmeans(k, X, balanced, 6)
    np.random.seed(3)
    label, inertia = singleKmeans(k, n, X, balanced, 6)
    assert np.array_equal(label, want_label) and inertia == want_inertia


@pytest.mark.gpu
def test_late_permutation_chunks_do_not_change_results(monkeypatch):
    """The parallel path uploads the epoch permutations in chunks of 8 epochs while training already runs; TrainJob.run
    must not launch a step whose batch tags need a chunk that has not arrived.  With the workers slowed down so that
    every chunk is late (the device would otherwise run far ahead), the models and logs stay bitwise the same."""
    from ultrare_amd import rng
    from ultrare_amd.method.sisa import Sisa
    S, E = 3, 20
    idx, trd, ted, tot = _sisa_inputs(S)
    monkeypatch.setenv('URE_DEVICE_TAGS', '0')          # (the host's chunk workers: with the tags made on the device there is nothing to be late)
    out = []
    for delay in (0.0, 0.03):
        monkeypatch.setattr(rng, '_TEST_CHUNK_DELAY_S', delay)
        sisa = Sisa(Param(E, parallel=True), 'mf', S, idx)
        torch.manual_seed(42)
        ml = sisa.learn(trd, ted, tot, 0, '')
        out.append(([m.item_mat.weight.detach().clone() for m in ml], ml[0].user_mat.weight.detach().clone(), dict(sisa.log), dict(sisa.log0)))
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(out[0][1], out[1][1])
    assert out[0][2]['train_loss'] == out[1][2]['train_loss'] and out[0][2]['total_ndcg'] == out[1][2]['total_ndcg']
    assert out[0][3] == out[1][3]


def test_a_request_returns_its_device_memory_without_the_cyclic_collector():
    """The job of a request -- tables, snapshots, batch tags, the scratch that made them -- goes back to the allocator when the request
    is over, by reference counts alone: with the cyclic collector off, the memory in use after the fifth request is what it was after
    the second (a job -> state -> job cycle once kept 0.7 GB per ml-1m request alive until a collection)."""
    import gc
    from ultrare_amd.method.sisa import Sisa
    S, E = 3, 4
    idx, trd, ted, tot = _sisa_inputs(S)
    gc.collect()
    gc.disable()
    try:
        used = []
        for rep in range(5):
            sisa = Sisa(Param(E, parallel=True), 'mf', S, idx)
            torch.manual_seed(42)
            sisa.learn(trd, ted, tot, 0, '')
            sisa._check_closed()
            torch.cuda.synchronize()
            del sisa
            used.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert used[4] <= used[1], used


def test_tags_made_on_the_device_train_like_the_hosts(monkeypatch):
    """Sisa(parallel) with the epochs' batch tags made on the device (ure_device_randperm_tags, the default) and with the host's
    expansion threads (URE_DEVICE_TAGS=0): the same models, the same logs, bit for bit."""
    from ultrare_amd.method.sisa import Sisa
    S, E = 4, 6
    idx, trd, ted, tot = _sisa_inputs(S)
    out = []
    for mode in ('1', '0'):
        monkeypatch.setenv('URE_DEVICE_TAGS', mode)
        sisa = Sisa(Param(E, parallel=True), 'mf', S, idx)
        torch.manual_seed(42)
        ml = sisa.learn(trd, ted, tot, 0, '')
        out.append(([m.item_mat.weight.detach().clone() for m in ml], ml[0].user_mat.weight.detach().clone(), dict(sisa.log), dict(sisa.log0)))
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]


@pytest.mark.parametrize('shuffle', ['auto', 'chain', 'reservations'])
def test_every_device_shuffle_trains_like_the_hosts(monkeypatch, shuffle):
    """The same with the kernel family forced (URE_SHUFFLE: csrc/perm_chain.hip for every chunk, csrc/perm_tags.hip for every chunk, the
    product's choice per chunk) -- and for Scratch.train (config.py:182-188's full-MF stage), whose tags are made on the device since round 5
    (rng.epoch_tags_device), epoch by epoch (verbose 1) and queued (verbose 0)."""
    from ultrare_amd.method.scratch import Scratch
    from ultrare_amd.method.sisa import Sisa
    S, E = 3, 5
    idx, trd, ted, tot = _sisa_inputs(S)
    runs = []
    for tags, how in (('1', shuffle), ('0', 'auto')):
        monkeypatch.setenv('URE_DEVICE_TAGS', tags)
        monkeypatch.setenv('URE_SHUFFLE', how)
        sisa = Sisa(Param(E, parallel=True), 'mf', S, idx)
        torch.manual_seed(42)
        ml = sisa.learn(trd, ted, tot, 0, '')
        got = [[m.item_mat.weight.detach().clone() for m in ml] + [ml[0].user_mat.weight.detach().clone()], dict(sisa.log)]
        for verbose in (0, 1):
            sc = Scratch(Param(E), 'mf')
            torch.manual_seed(42)
            m = sc.train(trd[0], ted[0], tot, verbose, '')
            got[0] += [m.user_mat.weight.detach().clone(), m.item_mat.weight.detach().clone()]
            got.append({k: v for k, v in sc.log.items() if k != 'time'})
        runs.append(got)
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    assert runs[0][1:] == runs[1][1:]


def test_cli_default_trains_shards_side_by_side_and_prints_the_same_lines(tmp_path, capsys):
    """main.py:60-70.  `python main.py --group 3` takes the shard-parallel path by default: models, logs, log0 and
    the per-epoch lines of scratch.py:99-118 equal those of `--parallel 0`, which trains shard after shard as the reference does."""
    import re
    import shutil
    from ultrare_amd.main import main, parser
    assert parser.get_default('parallel') == 1
    data = tmp_path / 'data'
    (data / 'toy').mkdir(parents=True)
    shutil.copy(TRAIN, data / 'toy' / '0_train.csv')
    shutil.copy(TEST, data / 'toy' / '0_test.csv')
    main(['--dataset', 'toy', '--epoch', '2', '--group', '0', '--data-dir', str(data), '--save-dir', str(tmp_path / 'r0'), '--verbose', '0'])
    shutil.copytree(tmp_path / 'r0', tmp_path / 'r1')
    outs = []
    for par, save in (('0', 'r0'), (None, 'r1')):
        capsys.readouterr()
        main(['--dataset', 'toy', '--epoch', '2', '--group', '3', '--data-dir', str(data), '--save-dir', str(tmp_path / save)] +
             (['--parallel', par] if par is not None else []))
        outs.append([re.sub(r'time: \S+', 'time: -', ln) for ln in capsys.readouterr().out.splitlines() if ln.startswith(('Epoch', 'Using'))])
    assert outs[0] == outs[1] and sum(ln.startswith('Epoch') for ln in outs[0]) >= 3 * 2
    for run in ('MF_emb-ot_sisa_learn', 'MF_emb-ot_sisa_unlearn'):
        a, b = (tmp_path / r / '2' / 'rand' / 'toy_g3' / run for r in ('r0', 'r1'))
        assert np.load(a / 'log0.npy', allow_pickle=True).item() == np.load(b / 'log0.npy', allow_pickle=True).item()
        for i in (1, 2, 3):
            if (a / f'item_mat{i}.npy').exists():
                assert np.array_equal(np.load(a / f'item_mat{i}.npy'), np.load(b / f'item_mat{i}.npy'))
                la, lb = (np.load(x / f'log{i}.npy', allow_pickle=True).item() for x in (a, b))
                # (verbose 1: the sequential path reads every epoch's per-user results back and reduces them with numpy, the parallel
                # path reduces on the device in a fixed order -- equal to rounding; bit-identical at verbose 0, test_parallel_equals_...)
                for k in la:
                    if k != 'time':
                        np.testing.assert_allclose(la[k], lb[k], rtol=1e-6, err_msg=k)
