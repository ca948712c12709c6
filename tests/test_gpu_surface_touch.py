"""The operator surface (Scratch.train, Sisa.learn / unlearn: sisa.py:25-118, scratch.py:51-148) with the engine in its touch modes
(csrc/mf_touch.h, csrc/mf_index.h).  The auto rule selects them only above 256 MB of live rows, so the golden tests of
test_gpu_surface.py / test_gpu_scale.py run the default kernel; here the same checks run with URE_TOUCH=1 -- touch_mode 2 (masks one
epoch ahead) where the caller reads the tables at the end, touch_mode 1 with URE_TOUCH_AHEAD=0, touch_mode 3 for epochs of more than
63 steps -- and BASELINE.json configs[3]'s shape runs through Sisa(parallel) against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture
def jobs_seen(monkeypatch):
    """(touch, ahead, index) of every engine.TrainJob the operator surface creates."""
    from ultrare_amd import engine
    seen = []
    init = engine.TrainJob.__init__

    def spy(self, *a, **k):
        init(self, *a, **k)
        seen.append((self.touch, self.ahead, self.index))
    monkeypatch.setattr(engine.TrainJob, '__init__', spy)
    return seen


def _force(monkeypatch, mode):
    monkeypatch.setenv('URE_TOUCH', '1')
    monkeypatch.setenv('URE_TOUCH_AHEAD', '1' if mode == 'ahead' else '0')


@pytest.mark.parametrize('mode', ['ahead', 'windows'])
@pytest.mark.parametrize('S,E', [(3, 2), (4, 3)])
def test_sisa_learn_unlearn_matches_reference_in_touch_mode(S, E, mode, tmp_path, monkeypatch, jobs_seen):
    from test_gpu_surface import check_sisa_against_reference
    _force(monkeypatch, mode)
    check_sisa_against_reference(S, E, True, tmp_path)
    assert jobs_seen and all(t and a == (mode == 'ahead') and not i for t, a, i in jobs_seen), jobs_seen


@pytest.mark.parametrize('mode', ['ahead', 'windows'])
def test_sequential_sisa_matches_reference_in_touch_mode(mode, tmp_path, monkeypatch, jobs_seen):
    from test_gpu_surface import check_sisa_against_reference
    _force(monkeypatch, mode)
    check_sisa_against_reference(3, 2, False, tmp_path)
    assert jobs_seen and all(t for t, _, _ in jobs_seen)


@pytest.mark.parametrize('mode', ['ahead', 'windows'])
def test_ml1m_size_vs_reference_golden_in_touch_mode(mode, tmp_path, monkeypatch, jobs_seen):
    from ultrare_amd import synth
    from test_gpu_scale import check_ml1m_size_against_reference
    _force(monkeypatch, mode)
    check_ml1m_size_against_reference(synth.make_dataset(**synth.ML1M), tmp_path)
    assert jobs_seen and all(t and a == (mode == 'ahead') for t, a, _ in jobs_seen), jobs_seen


@pytest.mark.parametrize('verbose', [0, 1])
def test_scratch_train_with_long_epochs_in_touch_mode_3(verbose, tmp_path, monkeypatch, jobs_seen, capsys):
    """Scratch.train (scratch.py:51-148) with 65 optimizer steps per epoch: touch_mode 3 (the epoch's slots sorted by step), epoch by
    epoch (verbose 1: tables read and tested at every epoch end) and as one queued run (verbose 0: series from snapshots), against the
    oracle's Scratch.train on the same stream."""
    from ultrare_amd.method.scratch import Scratch
    from ultrare_amd.read import RatingData, loadData, readRating
    from test_gpu_surface import N_USER, N_ITEM, TEST, TRAIN, Param
    monkeypatch.setenv('URE_TOUCH', '1')
    E, B = 3, 437
    tr, idx = readRating(TRAIN, N_USER, 5, [], [], 1, [])
    te, _ = readRating(TEST, N_USER, 5, [], [], 1, idx)
    train, test = loadData(RatingData(tr[0]), B, 24), loadData(RatingData(te[0]), B, 24, False)
    sc = Scratch(Param(E, batch=B), 'mf')
    torch.manual_seed(42)
    model = sc.train(train, test, [], verbose, str(tmp_path))
    assert jobs_seen == [(True, False, True)], jobs_seen
    part = tuple(np.ascontiguousarray(a) for a in (tr[0][0].astype(np.int32), tr[0][1].astype(np.int32), tr[0][2].astype(np.float32)))
    tpart = tuple(np.ascontiguousarray(a) for a in (te[0][0].astype(np.int32), te[0][1].astype(np.int32), te[0][2].astype(np.float32)))
    torch.manual_seed(42)
    U, V, log = O.scratch_train(O.Hyper(k=16, batch=B, epochs=E), N_USER, N_ITEM, part, tpart)
    assert rel(model.user_mat.weight, U) < 2e-5 and rel(model.item_mat.weight, V) < 2e-5
    for key in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr'):
        np.testing.assert_allclose(sc.log[key], log[key], rtol=1e-4, err_msg=key)


_ML25M = []


def _ml25m():
    from ultrare_amd import synth
    if not _ML25M:
        _ML25M.append(synth.make_dataset(**synth.ML25M))
    return _ML25M[0]


@pytest.mark.parametrize('index_env', ['1', '0'])
def test_configs3_shape_through_sisa_parallel_against_the_oracle(index_env, monkeypatch, jobs_seen):
    """BASELINE.json configs[3]'s shape (162,000 x 60,000, 22.5 M ratings, 32 shards) through Sisa(parallel) at k = 16, where the
    reference's arithmetic stays finite: the auto rule puts the job into touch mode (373 MB of live rows, tables read at the end) -- touch_mode 3
    at this row width (engine.INDEX_SHORT_EPOCH_MAX_D), touch_mode 2 in the second case (URE_TOUCH_INDEX=0) --, the per-epoch logs come from
    compact snapshots through ure_eval_series_compact, the shards' own rows are merged.  Shards 0 and 1 are
    checked against the oracle's Scratch.train on the same stream: item tables, own user rows of the merged matrix, and both epochs of
    every log series (shard 1's tests average shard 0's final model in: scratch.py:83-86)."""
    from ultrare_amd import synth
    from ultrare_amd.method.sisa import Sisa
    from ultrare_amd.read import RatingData, loadData
    monkeypatch.delenv('URE_TOUCH', raising=False)
    monkeypatch.setenv('URE_TOUCH_INDEX', index_env)
    spec = synth.ML25M
    data = _ml25m()
    S, E, B, k = 32, 2, 30000, 16

    class P:
        lam, seed, batch, lr, lr_decay, momentum, epochs, parallel = 0.1, 42, B, 0.001, 0.95, 0.9, E, True
        n_user, n_item = spec['n_user'], spec['n_item']
    P.k = k
    shard_of, groups = synth.uniform_shards(P.n_user, S)
    parts = synth.split_shards(data['train'], shard_of, S)
    parts_te = synth.split_shards(data['test'], shard_of, S)

    def arr(t):
        return np.vstack([t[0].astype(np.float64), t[1].astype(np.float64), t[2] / 5.0])
    trd = [loadData(RatingData(arr(p)), B, 24) for p in parts]
    ted = [loadData(RatingData(arr(p)), B, 24, False) for p in parts_te]
    total = O.hstack(parts_te)
    tot = loadData(RatingData(arr(total)), B, 24, False)
    sisa = Sisa(P, 'mf', S, groups)
    torch.manual_seed(42)
    ml = sisa.learn(trd, ted, tot, 0, '')
    # the auto rule: narrow rows (k = 16) take touch_mode 3 for short epochs too; without it touch_mode 2
    assert jobs_seen == ([(True, False, True)] if index_env == '1' else [(True, True, False)]), jobs_seen
    assert len(sisa.log['train_loss']) == S * E and np.isfinite(sisa.log['total_rmse']).all()

    def f32(p):
        return (np.ascontiguousarray(p[0], dtype=np.int32), np.ascontiguousarray(p[1], dtype=np.int32), np.ascontiguousarray(p[2] / 5.0, dtype=np.float32))
    h = O.Hyper(k=k, batch=B, epochs=E)
    torch.manual_seed(42)
    prev = []
    merged = ml[0].user_mat.weight.detach().cpu().numpy()
    for s in (0, 1):
        U, V, log = O.scratch_train(h, P.n_user, P.n_item, f32(parts[s]), f32(parts_te[s]), f32(total), prev_models=prev)
        prev.append((U, V))
        rows = np.asarray(groups[s])
        assert rel(ml[s].item_mat.weight, V) < 1e-4, s
        assert rel(merged[rows], U[rows]) < 1e-4, s
        for key in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr'):
            np.testing.assert_allclose(sisa.log[key][s * E:(s + 1) * E], log[key], rtol=1e-4, err_msg=f'{key} of shard {s}')
