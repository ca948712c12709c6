"""GPU parity of the engine layer (TrainJob / EvalSet through the C ABI) against the
CPU oracle and against goldens produced by the real reference."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
N_USER, N_ITEM = 1508, 2071
RTOL = 1e-4     # BASELINE.json north_star: 1e-4 relative on learned embeddings / metrics


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max())


@pytest.fixture(scope='module')
def toy():
    tr = O.load_csv(os.path.join(G, 'toy', '0_train.csv'))
    te = O.load_csv(os.path.join(G, 'toy', '0_test.csv'))
    full = [list(range(N_USER))]
    return O.partition(*tr, full)[0], O.partition(*te, full)[0]


def _train_gpu(train, k, batch, epochs, seed=42, lr=1e-3, lam=0.1, mu=0.9, n_user=N_USER, n_item=N_ITEM):
    from ultrare_amd import engine, rng
    torch.manual_seed(seed)
    U0, V0 = rng.mf_init(n_user, n_item, k)
    seeds = rng.epoch_seeds(epochs, False)
    perms = rng.epoch_perms(seeds, len(train[0]))
    sh = engine.ShardData(*train, n_user, n_item)
    job = engine.TrainJob([sh], [(U0, V0)], [perms], k, batch, epochs, lr, lam, mu, 0.95)
    job.run()
    torch.cuda.synchronize()
    U, V = job.tables(0)
    return U.cpu().numpy(), V.cpu().numpy(), job, (U0.numpy(), V0.numpy(), perms.numpy())


@pytest.mark.parametrize('E', [1, 3, 50])
def test_full_mf_vs_reference_golden(toy, E):
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    train, test = toy
    U, V, job, _ = _train_gpu(train, 16, 3000, E)
    assert rel(U, g[f'E{E}_U']) < RTOL
    assert rel(V, g[f'E{E}_V']) < RTOL
    loss = np.sqrt(job.epoch_sse(0) / len(train[0]))
    np.testing.assert_allclose(loss, g[f'E{E}_train_loss'], rtol=RTOL)


@pytest.mark.parametrize('k,batch', [(4, 1000), (8, 3000), (16, 3000), (32, 5000), (32, 3000), (64, 3000), (128, 30000),
                                     (20, 3000), (256, 2000), (16, 15000), (16, 9454)])
def test_step_kernel_vs_oracle(toy, k, batch):
    """Every table width (incl. a padded one; the narrow ones cut the toy rows into several work
    units, whose partial sums meet in LDS), batch larger than the shard (1 step per epoch), 2 and 3
    steps per epoch (the boundary between standalone and riding batch-tag preparation); 2 epochs
    against the C oracle on identical init/perms."""
    train, _ = toy
    E = 2
    lr = 1e-3 if k <= 128 else 1e-5          # N(0,1) tables of width 256 diverge at 1e-3 (in the oracle too)
    U, V, job, (U0, V0, perms) = _train_gpu(train, k, batch, E, lr=lr)
    st = O.MFState(U0.copy(), V0.copy())
    losses = []
    for t in range(E):
        losses.append(O.train_epoch(st, train, perms[t], batch, lr, 0.1, 0.9)[0])
    assert np.isfinite(st.U).all()
    assert rel(U, st.U) < 1e-5
    assert rel(V, st.V) < 1e-5
    np.testing.assert_allclose(np.sqrt(job.epoch_sse(0) / len(train[0])), losses, rtol=1e-5)


def test_out_of_order_ticks_are_refused(toy):
    import ctypes
    from ultrare_amd import _native as nv
    train, _ = toy
    _, _, job, _ = _train_gpu(train, 16, 3000, 1)
    rc = nv.lib().ure_job_train(job._job, 3, 5, nv.stream_handle())
    assert rc != 0 and b'in order' in nv.lib().ure_last_error()


@pytest.mark.parametrize('k', [8, 32, 256])
def test_heavy_rows_vs_oracle(k):
    """Rows far longer than a workgroup's 2048 slots (one user who rated all 2600 items, one item
    rated by all 300 users): their units take several passes and all lane groups of a workgroup."""
    rs = np.random.RandomState(5)
    n_user, n_item = 300, 2600
    u = [np.zeros(n_item, dtype=np.int64)]
    i = [np.arange(n_item)]
    for a in range(1, n_user):
        it = np.unique(np.concatenate([[7], rs.choice(n_item, rs.randint(1, 40), replace=False)]))
        u.append(np.full(len(it), a)); i.append(it)
    u, i = np.concatenate(u), np.concatenate(i)
    order = rs.permutation(len(u))
    train = (u[order].astype(np.int32), i[order].astype(np.int32), (rs.randint(1, 6, len(u)) / 5).astype(np.float32))
    lr = 1e-3 if k <= 128 else 1e-5
    U, V, job, (U0, V0, perms) = _train_gpu(train, k, 1500, 2, lr=lr, n_user=n_user, n_item=n_item)
    st = O.MFState(U0.copy(), V0.copy())
    losses = [O.train_epoch(st, train, perms[t], 1500, lr, 0.1, 0.9)[0] for t in range(2)]
    assert rel(U, st.U) < 1e-5 and rel(V, st.V) < 1e-5
    np.testing.assert_allclose(np.sqrt(job.epoch_sse(0) / len(train[0])), losses, rtol=1e-5)


@pytest.mark.parametrize('mode', ['0', '1', '2'])
def test_workgroup_mappings_agree(toy, mode, monkeypatch):
    """URE_SHARD_FAST picks how workgroups are dealt out to shards / XCDs (plain 2-D grid, shard-fast,
    sliced = default): a placement matter only, so three shards side by side must train bit for bit
    the same under each."""
    from ultrare_amd import engine, rng
    raw = O.load_csv(os.path.join(G, 'toy', '0_train.csv'))
    parts = O.partition(*raw, O.uniform_groups(N_USER, 3))
    k, B, E = 16, 3000, 2

    def run():
        torch.manual_seed(11)
        inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
        perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
        job = engine.TrainJob([engine.ShardData(*p, N_USER, N_ITEM) for p in parts], inits, perms, k, B, E, 1e-3, 0.1, 0.9)
        job.run()
        torch.cuda.synchronize()
        return [t.clone() for s_ in range(len(parts)) for t in job.tables(s_)]

    monkeypatch.delenv('URE_SHARD_FAST', raising=False)
    want = run()
    monkeypatch.setenv('URE_SHARD_FAST', mode)
    got = run()
    assert all(torch.equal(a, b) for a, b in zip(want, got))


def test_eval_series_equals_single_evaluations(toy):
    """ure_eval_series (all epochs of a shard in four launches) against one evaluate() per epoch on
    the same model lists: identical results, with and without fixed models, chunked or not."""
    from ultrare_amd import engine, rng
    train, test = toy
    torch.manual_seed(3)
    k, E = 16, 5
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in range(3)]
    perms = rng.epoch_perms(rng.epoch_seeds(E, False), len(train[0]))
    job = engine.TrainJob([engine.ShardData(*train, N_USER, N_ITEM)], [inits[0]], [perms], k, 3000, E, 1e-3, 0.1, 0.9, snapshots=True)
    job.run()
    ev = engine.EvalSet(*test)
    fixed = [tuple(t.to(ev.device).contiguous() for t in init) for init in inits[1:]]
    snapU, snapV = job.snapshots_of(0)
    many = [tuple(torch.randn_like(t) * 0.3 for t in fixed[0]) for _ in range(35)]       # more than one chunk of 32 fixed models
    for before in ([], fixed, many):
        want = torch.zeros(E, 3, dtype=torch.float64, device=ev.device)
        for e in range(E):
            ev.evaluate(before + [job.snapshot(0, e)], job.d, out=want[e])
        for cap in (engine.SERIES_SCRATCH_BYTES, 4 * ev.n * 2):          # second: two members per call
            engine.SERIES_SCRATCH_BYTES, old = cap, engine.SERIES_SCRATCH_BYTES
            ev._series_cap = 0
            try:
                got = ev.evaluate_series(before, snapU, snapV, job.d, torch.zeros(E, 3, dtype=torch.float64, device=ev.device))
            finally:
                engine.SERIES_SCRATCH_BYTES = old
            torch.cuda.synchronize()
            assert torch.equal(got, want)


def test_many_epochs_vs_oracle(toy):
    """120 epochs of two shards side by side (the batch tags of epoch e+1 are prepared by extra workgroups of
    epoch e's launches into the other half of a double buffer, 119 times over) against the C oracle, with the
    StepLR decay crossing its boundaries (epochs 50 and 100)."""
    from ultrare_amd import engine, rng
    raw = O.load_csv(os.path.join(G, 'toy', '0_train.csv'))
    parts = O.partition(*raw, O.uniform_groups(N_USER, 2))
    k, B, E = 16, 1500, 120
    torch.manual_seed(9)
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
    perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
    job = engine.TrainJob([engine.ShardData(*p, N_USER, N_ITEM) for p in parts], inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95)
    job.run()
    torch.cuda.synchronize()
    for s_, p in enumerate(parts):
        st = O.MFState(inits[s_][0].numpy().copy(), inits[s_][1].numpy().copy())
        losses = [O.train_epoch(st, p, perms[s_][t].numpy(), B, 1e-3 * 0.95 ** (t // 50), 0.1, 0.9)[0] for t in range(E)]
        U, V = job.tables(s_)
        assert rel(U.cpu().numpy(), st.U) < 1e-4 and rel(V.cpu().numpy(), st.V) < 1e-4
        np.testing.assert_allclose(np.sqrt(job.epoch_sse(s_) / len(p[0])), losses, rtol=1e-4)


def test_bitwise_reproducible(toy):
    train, _ = toy
    a = _train_gpu(train, 32, 3000, 2)
    b = _train_gpu(train, 32, 3000, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_shards_side_by_side_equal_alone(toy):
    """A job's shards share launches but must train exactly as if alone."""
    from ultrare_amd import engine, rng
    (tu, ti, tr), _ = toy
    idx = O.uniform_groups(N_USER, 3)
    raw = O.load_csv(os.path.join(G, 'toy', '0_train.csv'))
    parts = O.partition(*raw, idx)
    k, B, E = 16, 3000, 3
    torch.manual_seed(7)
    inits, perms = [], []
    for p in parts:
        inits.append(rng.mf_init(N_USER, N_ITEM, k))
        perms.append(rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])))
    shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9)
    job.run()
    for s, p in enumerate(parts):
        solo = engine.TrainJob([engine.ShardData(*p, N_USER, N_ITEM)], [inits[s]], [perms[s]], k, B, E, 1e-3, 0.1, 0.9)
        solo.run()
        torch.cuda.synchronize()
        for a, b in zip(job.tables(s), solo.tables(0)):
            assert torch.equal(a, b)
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        for t in range(E):
            O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3, 0.1, 0.9)
        assert rel(job.tables(s)[0].cpu().numpy(), st.U) < 1e-5
        assert rel(job.tables(s)[1].cpu().numpy(), st.V) < 1e-5


def test_eval_unit_vectors_exact():
    """HR exact, NDCG to 1e-12, on the reference's own baseTest outputs; the scores are
    injected through 1-wide 'tables' so that the ranking kernel sees prescribed values."""
    from ultrare_amd import engine
    g = np.load(os.path.join(G, 'eval_vectors.npz'))
    for c in range(int(g['n_cases'])):
        u, i, r, scores = g[f'c{c}_u'], g[f'c{c}_i'], g[f'c{c}_r'], g[f'c{c}_scores']
        S, n = scores.shape
        ev = engine.EvalSet(u, np.arange(n), r)            # item id = row id -> V row holds the score
        models = []
        for m in range(S):
            U = torch.zeros(int(u.max()) + 1, 4, device='cuda')
            U[:, 0] = 1.0
            V = torch.zeros(n, 4, device='cuda')
            V[:, 0] = torch.from_numpy(scores[m]).cuda()
            models.append((U, V))
        rmse, ndcg, hr = ev.evaluate(models, 4)
        want = g[f'c{c}_expect']
        assert abs(rmse - want[0]) < 1e-6 * want[0]
        assert abs(ndcg - want[1]) < 1e-12, (c, ndcg, want[1])
        assert abs(hr - want[2]) < 1e-12


@pytest.mark.parametrize('E', [1, 50])
def test_eval_on_reference_models(toy, E):
    """Score + rank the reference's own trained tables: metrics vs its baseTest."""
    from ultrare_amd import engine
    g = np.load(os.path.join(G, 'full_mf_toy.npz'))
    _, test = toy
    ev = engine.EvalSet(*test)
    U = torch.from_numpy(g[f'E{E}_U']).cuda()
    V = torch.from_numpy(g[f'E{E}_V']).cuda()
    got = ev.evaluate([(U, V)], 16)
    np.testing.assert_allclose(got, g[f'E{E}_final_stable'], rtol=RTOL)
    pred = ev.predictions()
    assert rel(pred, O.score([(g[f'E{E}_U'], g[f'E{E}_V'])], test[0], test[1])) < 1e-5


def test_large_shard_scatter_fallback_and_many_epochs():
    """A shard beyond the LDS partition's reach (> 1024 ranges of 2048 = 2.1 M interactions)
    takes the plain-scatter tag path; 3 epochs so that epoch parity of the tag buffers and
    the per-epoch standalone launches are exercised.  d = 4 keeps the oracle quick."""
    from ultrare_amd import engine, rng
    rs = np.random.RandomState(3)
    n_user, n_item, n = 4000, 3000, 2_200_000
    key = rs.choice(n_user * n_item, n, replace=False)
    part = ((key // n_item).astype(np.int32), (key % n_item).astype(np.int32), rs.choice([.2, .4, .6, .8, 1.], n).astype(np.float32))
    k, B, E = 4, 300_000, 3
    torch.manual_seed(1)
    U0, V0 = rng.mf_init(n_user, n_item, k)
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), n)
    job = engine.TrainJob([engine.ShardData(*part, n_user, n_item)], [(U0, V0)], [perms], k, B, E, 1e-4, 0.1, 0.9)
    job.run()
    torch.cuda.synchronize()
    st = O.MFState(U0.numpy().copy(), V0.numpy().copy())
    losses = [O.train_epoch(st, part, perms[t].numpy(), B, 1e-4, 0.1, 0.9)[0] for t in range(E)]
    assert np.isfinite(st.U).all()
    U, V = job.tables(0)
    assert rel(U.cpu().numpy(), st.U) < 1e-4 and rel(V.cpu().numpy(), st.V) < 1e-4
    np.testing.assert_allclose(np.sqrt(job.epoch_sse(0) / n), losses, rtol=1e-4)


def test_edge_cases_tiny_shards_ragged_eval_and_large_ensembles():
    """Ragged / degenerate inputs: shards of 1 and 7 interactions, width 1 (padded to 4), a test
    user with more than 512 items (the global-memory top-10 path), an ensemble of 33 models
    (more than one ure_score call), an empty test set."""
    from ultrare_amd import engine, rng
    rs = np.random.RandomState(0)
    for n, k in ((1, 1), (7, 3)):
        part = (rs.randint(0, 5, n).astype(np.int32), rs.randint(0, 6, n).astype(np.int32), rs.rand(n).astype(np.float32))
        torch.manual_seed(n)
        U0, V0 = rng.mf_init(5, 6, k)
        perms = rng.epoch_perms(rng.epoch_seeds(4, True), n)
        job = engine.TrainJob([engine.ShardData(*part, 5, 6)], [(U0, V0)], [perms], k, 4, 4, 1e-2, 0.1, 0.9)
        job.run()
        st = O.MFState(U0.numpy().copy(), V0.numpy().copy())
        for t in range(4):
            O.train_epoch(st, part, perms[t].numpy(), 4, 1e-2, 0.1, 0.9)
        U, V = job.tables(0)
        assert rel(U.cpu().numpy(), st.U) < 1e-5 and rel(V.cpu().numpy(), st.V) < 1e-5
    # evaluation: one user with 700 items, one with 65, many small ones; 33 models
    cnts = [700, 65, 64, 1, 10, 11] + [3] * 40
    uid = np.repeat(np.arange(len(cnts)), cnts).astype(np.int32)
    n = len(uid)
    iid = rs.randint(0, 50, n).astype(np.int32)
    r = rs.choice([0.2, 0.4, 0.6, 0.8, 1.0], n).astype(np.float32)
    models = [(rs.standard_normal((len(cnts), 8)).astype(np.float32) * 0.4, rs.standard_normal((50, 8)).astype(np.float32) * 0.4)
              for _ in range(33)]
    ev = engine.EvalSet(uid, iid, r)
    dev = [(torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()) for U, V in models]
    got = ev.evaluate(dev, 8)
    want = O.eval_metrics((uid, iid, r), models, 3000)
    np.testing.assert_allclose(got, want, rtol=1e-5)
    out = torch.zeros(3, dtype=torch.float64, device='cuda')
    ev.evaluate(dev, 8, out=out)                                     # queued form: same numbers
    np.testing.assert_allclose(out.cpu().numpy(), got, rtol=1e-12)
    empty = engine.EvalSet(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    assert all(np.isnan(x) for x in empty.evaluate(dev[:1], 8))


# ---------------------------------------------------------------- compact end-of-epoch snapshots (round 3)
@pytest.mark.parametrize('touch', [False, True])
def test_compact_snapshots_give_the_series_of_full_snapshots(toy, touch):
    """scratch.py:83-97 after shards were trained side by side: the per-epoch test series from COMPACT snapshots (the rows
    with interactions in the shard; every other row rebuilt as a_e * w0 inside ure_eval_series_compact) are bit for bit those
    from full snapshots.  Four shards of the toy set: three quarters of a shard's user rows are never trained in it."""
    from ultrare_amd import engine, rng
    from oracle import cpu_ref as O
    train, test = toy
    S, k, E, B = 4, 16, 6, 700
    idx = O.uniform_groups(N_USER, S)
    parts = O.partition(*train, idx)
    torch.manual_seed(9)
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
    perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
    shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
    assert all(sh.n_active < 0.6 * (N_USER + N_ITEM) for sh in shards)
    jobs = {m: engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, snapshots=m, touch=touch, lazy_rows=True) for m in ('full', 'compact')}
    assert jobs['compact'].snapshots == 'compact' and jobs['full'].snapshots == 'full' and jobs['compact'].touch == touch
    for j in jobs.values():
        j.run()
    ev = engine.EvalSet(*test)
    fixed = [tuple(torch.randn(n, jobs['full'].d, device=ev.device) * 0.3 for n in (N_USER, N_ITEM)) for _ in range(2)]
    for s in range(S):
        for before in ([], fixed):
            res = {m: j.evaluate_series(s, ev, before, torch.zeros(E, 3, dtype=torch.float64, device=ev.device)) for m, j in jobs.items()}
            torch.cuda.synchronize()
            assert torch.isfinite(res['full']).all() and torch.equal(res['full'], res['compact']), (s, len(before))
        # and member e of the series is what evaluate() gives on the full tables of epoch e
        want = torch.zeros(E, 3, dtype=torch.float64, device=ev.device)
        for e in range(E):
            ev.evaluate(fixed + [jobs['full'].snapshot(s, e)], jobs['full'].d, out=want[e])
        got = jobs['compact'].evaluate_series(s, ev, fixed, torch.zeros(E, 3, dtype=torch.float64, device=ev.device))
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    small = engine.TrainJob.snapshot_bytes(shards, E, k, 'compact')
    assert small < 0.6 * engine.TrainJob.snapshot_bytes(shards, E, k, 'full')
    for j in jobs.values():
        j.close()


def test_compact_snapshots_need_lazy_rows(toy):
    """URE_LAZY_ROWS=0 streams every row like the reference's dense optimizer: there is no closed form to rebuild the
    untrained rows from, so the engine keeps full snapshots."""
    from ultrare_amd import engine, rng
    train, _ = toy
    torch.manual_seed(1)
    init = rng.mf_init(N_USER, N_ITEM, 8)
    perms = rng.epoch_perms(rng.epoch_seeds(2, False), len(train[0]))
    job = engine.TrainJob([engine.ShardData(*train, N_USER, N_ITEM)], [init], [perms], 8, 3000, 2, 1e-3, 0.1, 0.9, lazy_rows=False, snapshots='compact')
    assert job.snapshots == 'full'
    job.run()
    U, V = job.padded_tables(0)
    assert torch.equal(job.snapshot(0, 1)[0], U) and torch.equal(job.snapshot(0, 1)[1], V)
    job.close()


def test_series_from_own_scores_and_from_cached_bases_are_the_same_series(toy):
    """Two other routes to a shard's per-epoch series, against evaluate_series on the same compact snapshots, to the last bit:
    (a) the two-halves form of the C ABI (ure_score_own_compact, then ure_eval_series_own); (b) the fixed models given as a cached base
    (engine.ScoreCache: every model scored once per request, a shard's base = the vectors' sum in the ensemble's order -- what
    Sisa(parallel) queues) -- with 0, 2 and 35 fixed models (more than one ure_score / ure_sum_vectors call holds), in both kernels."""
    from ultrare_amd import engine, rng
    from oracle import cpu_ref as O
    train, test = toy
    S, k, E, B = 3, 16, 11, 900
    parts = O.partition(*train, O.uniform_groups(N_USER, S))
    tests = O.partition(*test, O.uniform_groups(N_USER, S))
    torch.manual_seed(4)
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
    perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
    shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
    total = engine.EvalSet(*test)
    own_sets = [engine.EvalSet(*t) for t in tests]
    many = [tuple(torch.randn(n, engine.pad_dim(k), device=total.device) * 0.3 for n in (N_USER, N_ITEM)) for _ in range(35)]
    zeros = lambda: torch.zeros(E, 3, dtype=torch.float64, device=total.device)
    for touch in (False, True):
        job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, snapshots='compact', touch=touch, lazy_rows=True)
        job.run()
        caches = {id(ev): engine.ScoreCache(ev, job.d) for ev in own_sets + [total]}
        for s in range(S):
            for before in ([], many[:2], many):
                for ev in (own_sets[s], total):
                    want = job.evaluate_series(s, ev, before, zeros())
                    got_own = ev.evaluate_series_own(before, job.own_scores(s, ev), job.d, zeros())
                    base = caches[id(ev)].base([(('m', j), lambda j: many[j], j) for j in range(len(before))])
                    assert len(base) == len(before)
                    got_base = job.evaluate_series(s, ev, base, zeros())
                    torch.cuda.synchronize()
                    assert torch.isfinite(want).all() and torch.equal(got_own, want) and torch.equal(got_base, want), (touch, s, len(before))
        assert len(caches[id(total)].vec) == 35                      # every model scored once per set
        job.close()


def test_upload_many_on_the_device():
    """Small descriptors through one pinned staging buffer and one asynchronous copy: values, shapes, dtypes and alignment."""
    from ultrare_amd import engine
    rs = np.random.RandomState(0)
    arrs = [rs.randint(0, 99, (13, 4)).astype(np.int32), rs.rand(50).astype(np.float32), np.arange(377, dtype=np.int64), np.zeros((0, 4), dtype=np.int32),
            rs.rand(3, 5).astype(np.float64)]
    for _ in range(3):                                        # the staging buffer is reused once its copy is done
        out = engine.upload_many(arrs, engine._device())
        for a, t in zip(arrs, out):
            assert tuple(t.shape) == a.shape and t.is_cuda and t.data_ptr() % 64 == 0 and np.array_equal(t.cpu().numpy(), a)


def test_two_launch_ranking_handles_ties_nans_and_every_segment_class():
    """ure_eval_users with a cached ranking of the ratings runs as two launches (csrc/mf_eval.hip: eval_rank_kernel leaves the
    ten predicted positions packed in the output slots, eval_metrics_kernel turns them into HR / NDCG).  Its fast ranking counts
    with the strict comparison and falls back when two of the best keys are equal: predictions drawn from FEW distinct values
    (ties in nearly every top-10), NaNs, and users of every class -- 1, 3, 10, 16 (four per wavefront), 17, 40, 64 (a lane per
    entry), 65, 128, 300, 512 (several per lane), 513, 700 (from memory: finished by the first launch) -- per user against the
    one-launch kernel (no cached ranking) and the means against the oracle."""
    from ultrare_amd import engine, _native as nv
    rs = np.random.RandomState(11)
    lengths = [1, 3, 10, 16, 17, 18, 31, 32, 33, 40, 64, 65, 128, 300, 512, 513, 700] * 3 + list(rs.randint(1, 90, 400))
    uid = np.repeat(rs.permutation(len(lengths)), lengths).astype(np.int32)
    uid = uid[rs.permutation(len(uid))]                         # interleaved: first-appearance order is not sorted order
    n = len(uid)
    r = rs.choice([0.2, 0.4, 0.6, 0.8, 1.0], n).astype(np.float32)
    for name, pred in (('ties', rs.choice(np.linspace(-1, 1, 7), n).astype(np.float32)),
                       ('distinct', rs.standard_normal(n).astype(np.float32)),
                       ('nan', np.where(rs.rand(n) < 0.2, np.nan, rs.choice(np.linspace(-1, 1, 5), n)).astype(np.float32))):
        ev = engine.EvalSet(uid, np.zeros(n, np.int32), r)
        ev.pred.copy_(torch.from_numpy(pred[ev.order]))
        L, st = nv.lib(), nv.stream_handle()
        out = {}
        for cached in (True, False):
            ev.hits.fill_(-7)
            ev.ndcg.fill_(-7.0)
            nv.check(L.ure_eval_users(nv.ptr(ev.off), ev.n_users, nv.ptr(ev.pred), nv.ptr(ev.rating), nv.ptr(ev.log2), nv.ptr(ev.hits), nv.ptr(ev.ndcg),
                                      nv.ptr(ev.top_rating) if cached else None, ev.n_wide, ev.n_half, st), 'ure_eval_users')
            out[cached] = (ev.hits.cpu().numpy().copy(), ev.ndcg.cpu().numpy().copy())
        assert np.array_equal(out[True][0], out[False][0]), name
        assert np.array_equal(out[True][1], out[False][1]), name          # the same arithmetic in the same order: bit for bit
        want = O.eval_from_pred(uid, r, pred, 3000)
        np.testing.assert_allclose([out[True][1].mean(), (out[True][0] / 10).mean()], want[1:], rtol=1e-12, err_msg=name)


def test_auto_touch_only_for_callers_that_read_at_epoch_ends():
    """a job above the touch threshold that makes no promise about WHEN it reads its tables must stay readable at any tick
    (the default kernel); with epoch_reads / final_only the auto rule may take a touch mode, whose tables exist at epoch boundaries
    (at the end of training in touch_mode 2) only -- reading them inside an epoch raises instead of returning rows that are valid
    for another step."""
    from ultrare_amd import engine, rng, _native as nv
    rs = np.random.RandomState(4)
    n_user, n_item, k, B, E, n = 3000, 2500, 16, 400, 2, 30000
    key = np.unique(rs.randint(0, n_user, n).astype(np.int64) * n_item + rs.randint(0, n_item, n))
    part = ((key // n_item).astype(np.int32), (key % n_item).astype(np.int32), (rs.randint(1, 6, len(key)) / 5).astype(np.float32))
    torch.manual_seed(3)
    init = rng.mf_init(n_user, n_item, k)
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(key))
    old = engine.TOUCH_MIN_TABLE_BYTES
    engine.TOUCH_MIN_TABLE_BYTES = 1 << 10           # the rule fires for this small job
    try:
        sh = engine.ShardData(*part, n_user, n_item)
        free = engine.TrainJob([sh], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95)
        assert not free.touch
        free.run(7)                                  # inside the first epoch (74 steps)
        U_mid, _ = free.tables(0)
        assert torch.isfinite(U_mid).all()
        free.close()
        bound = engine.TrainJob([sh], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, epoch_reads=True)
        assert bound.touch and bound.index           # 74 steps per epoch: the slots sorted by step
        bound.run(7)
        with pytest.raises(nv.NativeError, match='inside an epoch'):
            bound.tables(0)
        bound.run(bound.steps_per_epoch(0) - 7)
        assert torch.isfinite(bound.tables(0)[0]).all()
        bound.close()
    finally:
        engine.TOUCH_MIN_TABLE_BYTES = old


def test_one_series_on_the_total_set_yields_the_shard_sets_numbers_too():
    """scratch.py:83-97 tests every epoch on the shard's own test set and on the total test set, which config.py:144-148 builds from the
    shards' sets: EvalSet.subset_of recognises that, and ure_eval_subset reduces the shard's numbers from what the series on the total set
    left behind.  Against a series of its own on the shard's set: NDCG and HR bit for bit (the same per-user values added in the same
    order), RMSE to rounding; a set that is NOT the total set's rows of its users gets no plan."""
    from ultrare_amd import engine
    rs = np.random.RandomState(5)
    n_user, n_item, d, E = 900, 700, 16, 4
    lengths = rs.randint(1, 60, n_user)
    uid = np.repeat(np.arange(n_user), lengths).astype(np.int32)
    iid = rs.randint(0, n_item, len(uid)).astype(np.int32)
    r = rs.choice([0.2, 0.4, 0.6, 0.8, 1.0], len(uid)).astype(np.float32)
    groups = np.array_split(rs.permutation(n_user), 3)
    parts = [np.flatnonzero(np.isin(uid, g)) for g in groups]
    order = np.concatenate(parts)                                  # the total set = the shards' sets side by side
    total = engine.EvalSet(uid[order], iid[order], r[order])
    dev = total.device
    U = torch.randn(E, n_user, d, device=dev)
    V = torch.randn(E, n_item, d, device=dev)
    fixed = [(torch.randn(n_user, d, device=dev), torch.randn(n_item, d, device=dev))]
    out_total = torch.zeros(E, 3, dtype=torch.float64, device=dev)
    for g, rows in enumerate(parts):
        sub = engine.EvalSet(uid[rows], iid[rows], r[rows])
        plan = sub.subset_of(total)
        assert plan is not None and plan['n'] == len(groups[g])
        assert sub.subset_of(total) is plan                        # kept
        own = torch.zeros(E, 3, dtype=torch.float64, device=dev)
        sub.evaluate_series(fixed, U, V, d, own)
        via = torch.zeros(E, 3, dtype=torch.float64, device=dev)
        total.evaluate_series(fixed, U, V, d, out_total, subset=(plan, via))
        own, via = own.cpu().numpy(), via.cpu().numpy()
        assert np.array_equal(own[:, 1:], via[:, 1:]), g
        np.testing.assert_allclose(via[:, 0], own[:, 0], rtol=1e-6)
    changed = r[parts[0]].copy()
    changed[7] = 1.0 if changed[7] != 1.0 else 0.2
    assert engine.EvalSet(uid[parts[0]], iid[parts[0]], changed).subset_of(total) is None
    other_user = engine.EvalSet(np.full(3, n_user + 5, np.int32), np.zeros(3, np.int32), np.ones(3, np.float32))
    assert other_user.subset_of(total) is None


def test_start_tables_of_all_shards_in_one_launch_and_losses_in_one_launch():
    """ure_copy_rows_batch (utils.py:31-40's tables [rows, k] -> the job's padded [rows, d], columns [k, d) untouched, the closed form's
    copy beside it) and ure_epoch_sse_batch (scratch.py:72-77's per-epoch loss: the sum over the users in double, one fixed order)
    through the C ABI, on more tables than one launch takes (48)."""
    import ctypes
    from ultrare_amd import _native as nv
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    k, d, n = 5, 8, 53
    rows = [int(x) for x in torch.randint(1, 400, (n,), generator=g)]
    rows[7] = 0
    src = [torch.randn(r, k, generator=g).to(dev) for r in rows]
    dst = [torch.full((r, d), 7.0, device=dev) for r in rows]
    dst2 = [torch.full((r, d), 9.0, device=dev) if i % 3 else None for i, r in enumerate(rows)]
    arr = lambda ts: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None and t.numel() else (1 if t is not None else None)) for t in ts])
    # (a table of 0 rows has no storage: any non-NULL address stands for it)
    rows_a = (ctypes.c_int64 * n)(*rows)
    nv.check(nv.lib().ure_copy_rows_batch(n, arr(src), arr(dst), arr(dst2), rows_a, k, d, nv.stream_handle()), 'ure_copy_rows_batch')
    for s, a, b in zip(src, dst, dst2):
        assert torch.equal(a[:, :k], s) and bool((a[:, k:] == 7.0).all())
        if b is not None:
            assert torch.equal(b[:, :k], s) and bool((b[:, k:] == 9.0).all())
    assert nv.lib().ure_copy_rows_batch(1, arr(src[:1] + [None] * (n - 1)), None, None, rows_a, k, d, None) != 0        # dst missing: refused
    E = 7
    users = [int(x) for x in torch.randint(1, 3000, (n,), generator=g)]
    sse = [(torch.rand(E, u, generator=g) * 3).to(dev) for u in users]
    out = torch.zeros(n, E, dtype=torch.float64, device=dev)
    nv.check(nv.lib().ure_epoch_sse_batch(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in sse]), (ctypes.c_int64 * n)(*users), E, out.data_ptr(),
                                          nv.stream_handle()), 'ure_epoch_sse_batch')
    want = torch.stack([t.double().sum(dim=1) for t in sse])
    assert torch.allclose(out, want, rtol=1e-13, atol=0)
    again = torch.zeros_like(out)
    nv.check(nv.lib().ure_epoch_sse_batch(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in sse]), (ctypes.c_int64 * n)(*users), E, again.data_ptr(),
                                          nv.stream_handle()), 'ure_epoch_sse_batch')
    assert torch.equal(out, again)


@pytest.mark.parametrize('groups', [1, 7])
def test_batch_tags_made_on_the_device_equal_the_hosts(groups):
    """ure_device_randperm_tags (read.py:127-133: the RandomSampler's permutation of an epoch, as batch tags; MT19937 by its parallel
    phases, the shuffle with deterministic reservations) against ure_host_randperm_tags -- itself pinned to torch.randperm in
    tests/test_cpu_host.py -- bit for bit: shards of different sizes in ONE table (1 row, 2 rows, around the 624-output block of the
    generator, 65,536 + 1, the largest the path takes), seeds of 62 bits, several batch sizes; no workgroup gave up."""
    import ctypes
    from ultrare_amd import _native as nv, rng
    L = nv.lib()
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(7)
    cases = [(1, 1), (2, 1), (3, 2), (623, 100), (624, 7), (625, 624), (626, 1), (5000, 30000), (65537, 4097), (56321, 30000), ((1 << 18) + 5, 30000), (1 << 20, 30000)]
    if groups == 1:
        cases.append((70001, 999))          # 130 more shuffles on the one workgroup: its round counter (4,096 rounds between wipes of the reservations) wraps
    table, want, outs = [], [], []
    for n, batch in cases:
        reps = 130 if n == 70001 else 1 if n > 100000 else 3
        seeds = rs.randint(0, 2 ** 62, size=reps).astype(np.int64)
        host = torch.empty(reps, n, dtype=torch.int16)
        nv.check(L.ure_host_randperm_tags(seeds.ctypes.data, reps, n, batch, host.data_ptr(), 4), 'ure_host_randperm_tags')
        out = torch.full((reps, n), -1, dtype=torch.int16, device=dev)
        for r in range(reps):
            table.append((int(seeds[r]), out.data_ptr() + 2 * n * r, n, batch))
        want.append(host)
        outs.append(out)
    tab = np.array(table, dtype=rng.PERM_DTYPE)
    assert tab.itemsize == 24
    tab_d = torch.from_numpy(tab.view(np.uint8)).to(dev)
    n_max = max(n for n, _ in cases)
    words = int(L.ure_device_randperm_tags_scratch(n_max, groups))
    scratch = torch.zeros(words, dtype=torch.int32, device=dev)
    nv.check(L.ure_device_randperm_tags(tab_d.data_ptr(), len(tab), n_max, scratch.data_ptr(), words, groups, nv.stream_handle()), 'ure_device_randperm_tags')
    torch.cuda.synchronize()
    for (n, batch), host, out in zip(cases, want, outs):
        assert torch.equal(out.cpu(), host), (n, batch)
    flags = scratch[2 * ((n_max + 63) // 64 * 64) * groups:][:groups].cpu()
    assert int(flags.abs().sum()) == 0
    # refusals: more than 2^20 rows, scratch too small
    assert L.ure_device_randperm_tags(tab_d.data_ptr(), 1, (1 << 20) + 1, scratch.data_ptr(), words, 1, None) != 0
    assert L.ure_device_randperm_tags(tab_d.data_ptr(), len(tab), n_max, scratch.data_ptr(), 16, groups, None) != 0
