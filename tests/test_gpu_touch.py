"""Touch mode of the step kernel (csrc/mf_touch.h): a step visits only the rows it trains; the rows in
between are advanced by the optimizer's closed form when they are next trained.  Same results as the
reference's dense optimizer (scratch.py:64-69) within the 1e-4 bar of BASELINE.json -- checked against
the C oracle, against the default kernel, across epoch and StepLR boundaries, with snapshots, and at
configs[3] shard size."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TRAIN = os.path.join(G, 'toy', '0_train.csv')
N_USER, N_ITEM = 1508, 2071


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / np.abs(b).max())


def _setup(S, k, B, E, seed=42):
    from ultrare_amd import engine, rng
    raw = O.load_csv(TRAIN)
    parts = O.partition(*raw, O.uniform_groups(N_USER, S))
    torch.manual_seed(seed)
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
    perms = [rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])) for p in parts]
    shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
    return parts, inits, perms, shards


@pytest.mark.parametrize('S,k,B,E', [(1, 16, 3000, 3), (3, 32, 700, 5), (2, 64, 1500, 53), (2, 128, 500, 2), (3, 8, 450, 2)])
def test_touch_mode_vs_oracle(S, k, B, E):
    """Shards side by side in touch mode against the C oracle's dense optimizer: users with few ratings and
    most items are trained in a fraction of the steps only (B = 450: 21 steps per epoch), so nearly every
    update is followed by a closed-form advance; E = 53 crosses the StepLR boundary at epoch 50."""
    from ultrare_amd import engine
    parts, inits, perms, shards = _setup(S, k, B, E)
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
    assert job.touch
    job.run()
    for s, p in enumerate(parts):
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        losses = [O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3 * 0.95 ** (t // 50), 0.1, 0.9)[0] for t in range(E)]
        U, V = job.tables(s)
        assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, (s, rel(U, st.U), rel(V, st.V))
        np.testing.assert_allclose(np.sqrt(job.epoch_sse(s) / len(p[0])), losses, rtol=2e-5)
    job.close()


def test_touch_mode_vs_default_kernel_epoch_by_epoch():
    """The same job in both modes, tables read at every epoch boundary (materialize + continue): the two
    kernels stay within float32 rounding of each other; rows trained in every step differ least."""
    from ultrare_amd import engine
    k, B, E = 32, 1200, 4
    parts, inits, perms, shards = _setup(1, k, B, E, seed=3)
    jobs = [engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=t) for t in (False, True)]
    assert [j.touch for j in jobs] == [False, True]
    for e in range(E):
        for j in jobs:
            j.run_epochs(1)
        (U0, V0), (U1, V1) = jobs[0].tables(0), jobs[1].tables(0)
        assert rel(U1, U0) < 5e-6 and rel(V1, V0) < 5e-6, e
    for j in jobs:
        j.close()


def test_touch_mode_reads_only_at_epoch_boundaries():
    from ultrare_amd import _native as nv
    from ultrare_amd import engine
    parts, inits, perms, shards = _setup(1, 16, 3000, 2)
    job = engine.TrainJob(shards, inits, perms, 16, 3000, 2, 1e-3, 0.1, 0.9, 0.95, touch=True)
    job.run(3)                                   # 10 steps per epoch: inside epoch 0
    with pytest.raises(nv.NativeError, match='epoch boundaries'):
        job.tables(0)
    job.run(7)
    job.tables(0)                                # the boundary: fine, and training goes on from it
    job.run()
    job.tables(0)
    job.close()


def test_touch_mode_snapshots_equal_tables_at_epoch_ends():
    from ultrare_amd import engine
    k, B, E = 16, 2000, 3
    parts, inits, perms, shards = _setup(2, k, B, E, seed=8)
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True, snapshots=True)
    job.run()
    # the two shards have different step counts, so their epoch boundaries fall on different ticks: the last
    # snapshot of each equals its final tables ...
    for s in range(2):
        snapU, snapV = job.snapshots_of(s)
        U, V = job.padded_tables(s)
        assert torch.equal(snapU[E - 1], U) and torch.equal(snapV[E - 1], V)
    # ... and every snapshot equals the tables of the same shard trained alone, read epoch by epoch
    for s in range(2):
        one = engine.TrainJob([shards[s]], [inits[s]], [perms[s]], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
        for e in range(E):
            one.run_epochs(1)
            U, V = one.padded_tables(0)
            assert torch.equal(job.snapshots_of(s)[0][e], U) and torch.equal(job.snapshots_of(s)[1][e], V), (s, e)
        one.close()
    job.close()


def test_touch_mode_is_bitwise_reproducible():
    from ultrare_amd import engine
    out = []
    for _ in range(2):
        parts, inits, perms, shards = _setup(2, 32, 900, 3, seed=1)
        job = engine.TrainJob(shards, inits, perms, 32, 900, 3, 1e-3, 0.1, 0.9, 0.95, touch=True)
        job.run()
        out.append([tuple(t.clone() for t in job.tables(s)) for s in range(2)])
        job.close()
    for s in range(2):
        assert torch.equal(out[0][s][0], out[1][s][0]) and torch.equal(out[0][s][1], out[1][s][1])


def test_touch_mode_refusals():
    """Without lazy rows the engine falls back to the default kernel (the C ABI refuses a descriptor that asks for touch
    mode there); 95 steps per epoch -- refused in round 2 -- now run in two windows."""
    from ultrare_amd import engine
    parts, inits, perms, shards = _setup(1, 16, 300, 1)           # 95 steps per epoch
    job = engine.TrainJob(shards, inits, perms, 16, 300, 1, 1e-3, 0.1, 0.9, 0.95, touch=True)
    assert job.touch
    job.run()
    job.close()
    job = engine.TrainJob(shards, inits, perms, 16, 300, 1, 1e-3, 0.1, 0.9, 0.95, touch=True, lazy_rows=False)
    assert not job.touch
    job.close()


def test_touch_mode_d128_shards_at_configs3_size_vs_oracle():
    """configs[3] per-shard size (5,063 users x 60,000 items, ~703 k ratings, d = 128, 24 steps per epoch): two
    shards side by side in touch mode -- the mode bench.py's ml25m workload selects by itself -- against the C
    oracle after two epochs (the second starts from rows in both buffers)."""
    from ultrare_amd import engine, rng, synth
    n_user, n_item, k, B, S, E = 162000, 60000, 128, 30000, 2, 2
    parts = []
    for s in range(S):
        d = synth.make_dataset(5063, n_item, 703125, 78125, seed=31 + s)
        ids = np.sort(np.random.RandomState(60 + s).choice(n_user, 5063, replace=False))
        u, i, r = d['train']
        parts.append((ids[u].astype(np.int32), i.astype(np.int32), (r / 5).astype(np.float32)))
    torch.manual_seed(42)
    inits, perms = [], []
    for p in parts:
        inits.append(rng.mf_init(n_user, n_item, k))
        perms.append(rng.epoch_perms(rng.epoch_seeds(E, True), len(p[0])))
    shards = [engine.ShardData(*p, n_user, n_item) for p in parts]
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
    assert job.touch
    job.run()
    for s in range(S):
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        losses = [O.train_epoch(st, parts[s], perms[s][t].numpy(), B, 1e-3, 0.1, 0.9)[0] for t in range(E)]
        U, V = job.tables(s)
        assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, s
        np.testing.assert_allclose(np.sqrt(job.epoch_sse(s) / len(parts[s][0])), losses, rtol=2e-5)
    job.close()


def test_touch_rows_accounting_matches_the_permutations():
    """ure_job_touch_rows (what bench.py's roofline counts as streamed rows in touch mode): the number of (row, step)
    pairs of the last epoch in which a row has an interaction, recomputed here from the epoch's permutation."""
    from ultrare_amd import engine
    k, B, E = 16, 900, 3
    parts, inits, perms, shards = _setup(2, k, B, E, seed=12)
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
    job.run()
    got = job.touch_rows_per_step()
    for s, p in enumerate(parts):
        n = len(p[0])
        steps = (n + B - 1) // B
        perm = perms[s][E - 1].numpy()
        step_of = np.empty(n, dtype=np.int64)
        step_of[perm] = np.arange(n) // B                                  # read.py:133: batch b = perm[b*B:(b+1)*B]
        pairs = len(np.unique(p[0].astype(np.int64) * steps + step_of)) + len(np.unique((N_USER + p[1].astype(np.int64)) * steps + step_of))
        assert abs(got[s] * steps - pairs) < 0.5, (s, got[s] * steps, pairs)
    job.close()
    dense = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=False)
    assert dense.touch_rows_per_step() is None
    dense.close()


# ---------------------------------------------------------------- epochs longer than one 64-step window (round 3)
def _truncate(part, n):
    return tuple(x[:n] for x in part)


@pytest.fixture(params=['windows', 'index'])
def long_epochs(request, monkeypatch):
    """Epochs of more than 63 steps run in touch_mode 3 (csrc/mf_index.h: the epoch's slots sorted by step) by default and in 64-step
    windows (touch_mode 1) with URE_TOUCH_INDEX=0: the long-epoch tests run both ways."""
    monkeypatch.setenv('URE_TOUCH_INDEX', '1' if request.param == 'index' else '0')
    return request.param


@pytest.mark.parametrize('n_rows,B,k,E', [(None, 437, 16, 3), (None, 219, 32, 2), (27714, 37, 16, 2), (None, 219, 128, 2)])   # d = 128: two float4 per lane
def test_touch_mode_windows_vs_oracle(n_rows, B, k, E, long_epochs):
    """65, 130 and 750 optimizer steps per epoch (full MF at 25 M rows has 750: config.py:182-188 with batch 30,000): an
    epoch is worked off in windows of 64 steps, each with its own row masks; a row that is not trained in a window is
    carried to the window's end in one closed-form step and on from there.  Against the C oracle's dense optimizer."""
    from ultrare_amd import engine, rng
    raw = O.load_csv(TRAIN)
    part = O.partition(*raw, O.uniform_groups(N_USER, 1))[0]
    if n_rows:
        part = _truncate(part, n_rows)
    steps = (len(part[0]) + B - 1) // B
    assert steps in (65, 130, 750), steps
    torch.manual_seed(11)
    init = rng.mf_init(N_USER, N_ITEM, k)
    if k > 64:      # N(0, 1) rows of width 128 start with predictions of +-11 and diverge in the reference's arithmetic (oracle: NaN)
        init = tuple(t * 0.3 for t in init)
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(part[0]))
    job = engine.TrainJob([engine.ShardData(*part, N_USER, N_ITEM)], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
    assert job.touch and job.index == (long_epochs == 'index')
    job.run()
    st = O.MFState(init[0].numpy().copy(), init[1].numpy().copy())
    losses = [O.train_epoch(st, part, perms[t].numpy(), B, 1e-3, 0.1, 0.9)[0] for t in range(E)]
    U, V = job.tables(0)
    assert np.isfinite(st.U).all() and np.isfinite(st.V).all()
    assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, (rel(U, st.U), rel(V, st.V))
    np.testing.assert_allclose(np.sqrt(job.epoch_sse(0) / len(part[0])), losses, rtol=2e-5)
    job.close()


def test_touch_mode_windows_with_shards_of_different_length_and_the_steplr_boundary(long_epochs):
    """Three shards whose epochs have 93-96 steps (one full window and a short one, ending on different ticks), 52 epochs:
    window starts of different shards fall on different ticks, and epochs 51-52 run at the decayed learning rate."""
    from ultrare_amd import engine
    S, k, B, E = 3, 16, 100, 52
    parts, inits, perms, shards = _setup(S, k, B, E, seed=5)
    assert len({(len(p[0]) + B - 1) // B for p in parts}) > 1 and all((len(p[0]) + B - 1) // B > 64 for p in parts)
    job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True, snapshots='compact')
    job.run()
    for s, p in enumerate(parts):
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        for t in range(E):
            O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3 * 0.95 ** (t // 50), 0.1, 0.9)
        U, V = job.tables(s)
        assert rel(U, st.U) < 3e-5 and rel(V, st.V) < 3e-5, (s, rel(U, st.U), rel(V, st.V))
        # the last compact snapshot holds the active rows of the final tables
        sh = shards[s]
        rows = torch.as_tensor(sh._sched_host[:sh.n_active, 0].astype(np.int64)).to(U.device)
        full = torch.cat([job.padded_tables(s)[0], job.padded_tables(s)[1]])
        assert torch.equal(job.state[s]['snap'][E - 1], full[rows])
    job.close()


def test_touch_mode_windows_match_the_default_kernel_epoch_by_epoch(long_epochs):
    from ultrare_amd import engine
    k, B, E = 32, 150, 3
    parts, inits, perms, shards = _setup(1, k, B, E, seed=3)
    assert (len(parts[0][0]) + B - 1) // B > 128
    jobs = [engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=t) for t in (False, True)]
    for e in range(E):
        for j in jobs:
            j.run_epochs(1)
        (U0, V0), (U1, V1) = jobs[0].tables(0), jobs[1].tables(0)
        assert rel(U1, U0) < 1e-5 and rel(V1, V0) < 1e-5, e
    for j in jobs:
        j.close()


# ---------------------------------------------------------------- touch_mode 2: masks one epoch ahead (round 3)
@pytest.mark.parametrize('S,k,B,E', [(1, 16, 3000, 3), (3, 32, 700, 5), (2, 64, 1500, 53), (2, 128, 600, 2), (3, 8, 450, 4), (2, 16, 9000, 3)])
def test_touch_ahead_mode_equals_mode_1_bit_for_bit(S, k, B, E):
    """touch_mode 2 (no dense pass at the epoch starts: a row's owner carries it across the epoch boundary at its last step, with
    the next epoch's masks built one epoch ahead from tags prepared two ahead) applies the same table entries to the same values as
    mode 1, so the final tables, the training losses and every compact end-of-epoch snapshot agree to the last bit -- across the
    StepLR boundary (E = 53), for shards whose epochs end on different ticks, with standalone tag preparation (B = 9000: two
    steps per epoch, no riders) and with rows that have no step in an epoch (orphans).  And both agree with the C oracle."""
    from ultrare_amd import engine
    parts, inits, perms, shards = _setup(S, k, B, E)
    if k > 64:
        inits = [tuple(t * 0.3 for t in init) for init in inits]
    jobs = {}
    for ahead in (False, True):
        job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True, snapshots='compact', final_only=ahead)
        assert job.touch and job.ahead == ahead
        job.run()
        jobs[ahead] = job
    for s, p in enumerate(parts):
        (U1, V1), (U2, V2) = jobs[False].tables(s), jobs[True].tables(s)
        assert torch.equal(U1, U2) and torch.equal(V1, V2), s
        assert torch.equal(jobs[False].state[s]['snap'], jobs[True].state[s]['snap']), s
        assert np.array_equal(jobs[False].epoch_sse(s), jobs[True].epoch_sse(s))
        if E <= 5:
            st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
            for t in range(E):
                O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3, 0.1, 0.9)
            assert rel(U2, st.U) < 2e-5 and rel(V2, st.V) < 2e-5
    for j in jobs.values():
        j.close()


def test_touch_ahead_mode_orphans_and_table_reads():
    """Rows without a step in an epoch: with small batches out of a shard most light rows skip whole epochs (orphans), and are
    carried over by launch B.  Tables cannot be read before the end in this mode; the snapshots can."""
    from ultrare_amd import _native as nv
    from ultrare_amd import engine
    k, B, E = 16, 400, 4
    parts, inits, perms, shards = _setup(2, k, B, E, seed=9)
    # keep the first 1,500 interactions of each shard only: 4 steps per epoch, most rows idle in most epochs
    from ultrare_amd import rng
    parts = [tuple(x[:1500] for x in p) for p in parts]
    shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
    torch.manual_seed(2)
    perms = [rng.epoch_perms(rng.epoch_seeds(E, True), 1500) for _ in parts]
    jobs = {a: engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True, snapshots='compact', final_only=a) for a in (False, True)}
    jobs[True].run(4)                                  # the end of epoch 0 of both shards (4 steps per epoch)
    with pytest.raises(nv.NativeError, match='touch_mode 2'):
        jobs[True].tables(0)
    jobs[True].run()
    jobs[False].run()
    for s, p in enumerate(parts):
        assert torch.equal(jobs[False].tables(s)[0], jobs[True].tables(s)[0]) and torch.equal(jobs[False].tables(s)[1], jobs[True].tables(s)[1])
        assert torch.equal(jobs[False].state[s]['snap'], jobs[True].state[s]['snap'])
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        for t in range(E):
            O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3, 0.1, 0.9)
        assert rel(jobs[True].tables(s)[0], st.U) < 2e-5 and rel(jobs[True].tables(s)[1], st.V) < 2e-5
    for j in jobs.values():
        j.close()


def test_touch_mode_windows_heavy_row_pass_skipping(long_epochs):
    """A row far heavier than a workgroup's lane groups can cover one pass each (one item with 5,000 of the shard's 5,600 interactions)
    is cut into long work units of many scan passes; with epochs of several windows every pass carries the mask of the steps it holds
    and the lane groups skip the passes without a slot of the step.  94 steps per epoch (two windows), against the C oracle."""
    from ultrare_amd import engine, rng
    rs = np.random.RandomState(3)
    n_user, n_item, k, B, E = 5200, 40, 16, 60, 3
    heavy_users = rs.permutation(n_user)[:5000]
    u = np.concatenate([heavy_users, rs.randint(0, n_user, 600)])
    i = np.concatenate([np.zeros(5000, dtype=np.int64), rs.randint(1, n_item, 600)])
    key = np.unique(u.astype(np.int64) * n_item + i)                      # no duplicate (user, item)
    u, i = (key // n_item).astype(np.int32), (key % n_item).astype(np.int32)
    r = (rs.randint(1, 6, len(u)) / 5).astype(np.float32)
    part = (u, i, r)
    steps = (len(u) + B - 1) // B
    assert steps > 64
    torch.manual_seed(21)
    init = tuple(t * 0.3 for t in rng.mf_init(n_user, n_item, k))
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(u))
    sh = engine.ShardData(*part, n_user, n_item)
    assert sh.max_row >= 5000
    job = engine.TrainJob([sh], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
    assert job.touch and not job.ahead and job.index == (long_epochs == 'index')
    job.run()
    st = O.MFState(init[0].numpy().copy(), init[1].numpy().copy())
    losses = [O.train_epoch(st, part, perms[t].numpy(), B, 1e-3, 0.1, 0.9)[0] for t in range(E)]
    U, V = job.tables(0)
    assert np.isfinite(st.V).all()
    assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, (rel(U, st.U), rel(V, st.V))
    np.testing.assert_allclose(np.sqrt(job.epoch_sse(0) / len(u)), losses, rtol=2e-5)
    job.close()


@pytest.mark.parametrize('mode', ['default', 'touch', 'ahead'])
@pytest.mark.parametrize('S,k,B,E', [(3, 32, 700, 5), (2, 16, 4000, 3), (1, 8, 90000, 2)])
def test_host_batch_tags_train_like_permutations(mode, S, k, B, E):
    """struct ure_shard: file_tags (ABI 5).  The same job driven by the epoch permutations (the device derives the batch tags:
    csrc/tag_prep.h) and by the tags the host makes of the same permutations (rng.epoch_tags): identical tables and losses,
    bit for bit, in the default kernel and in both touch modes -- riders (3 or more steps per epoch), standalone tag launches
    (B = 4000 and 90000: 2 steps / 1 step per epoch) and the first epoch alike."""
    from ultrare_amd import engine, rng
    raw = O.load_csv(TRAIN)
    parts = O.partition(*raw, O.uniform_groups(N_USER, S))
    torch.manual_seed(7)
    inits = [rng.mf_init(N_USER, N_ITEM, k) for _ in parts]
    seeds = [rng.epoch_seeds(E, True) for _ in parts]
    perms = [rng.epoch_perms(sd, len(p[0])) for sd, p in zip(seeds, parts)]
    tags = [rng.epoch_tags(sd, len(p[0]), B) for sd, p in zip(seeds, parts)]
    got = []
    for feed in (perms, tags):
        shards = [engine.ShardData(*p, N_USER, N_ITEM) for p in parts]
        job = engine.TrainJob(shards, inits, feed, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=mode != 'default', final_only=mode == 'ahead')
        assert job.touch == (mode != 'default') and job.ahead == (mode == 'ahead' and max(job.steps_per_epoch(s) for s in range(S)) <= 63)
        job.run()
        got.append([(job.tables(s)[0].cpu().numpy().copy(), job.tables(s)[1].cpu().numpy().copy(), job.epoch_sse(s)) for s in range(S)])
        job.close()
    for a, b in zip(*got):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


# ---------------------------------------------------------------- touch_mode 3: the epoch's slots sorted by step (round 4)
@pytest.mark.parametrize('S,k,B,E', [(1, 16, 3000, 3), (3, 32, 700, 5), (2, 64, 1500, 53), (2, 128, 500, 2), (3, 8, 450, 2)])
def test_touch_index_mode_vs_oracle_at_any_epoch_length(S, k, B, E):
    """touch='index' forces touch_mode 3 for short epochs too (one mask word): shards side by side, the StepLR boundary, every table
    width -- against the C oracle's dense optimizer, and bitwise reproducible."""
    from ultrare_amd import engine
    parts, inits, perms, shards = _setup(S, k, B, E)
    got = []
    for rep in range(2):
        job = engine.TrainJob(shards, inits, perms, k, B, E, 1e-3, 0.1, 0.9, 0.95, touch='index')
        assert job.touch and job.index and not job.ahead
        job.run()
        got.append([(job.tables(s)[0].clone(), job.tables(s)[1].clone(), job.epoch_sse(s)) for s in range(S)])
        job.close()
    for s, p in enumerate(parts):
        st = O.MFState(inits[s][0].numpy().copy(), inits[s][1].numpy().copy())
        losses = [O.train_epoch(st, p, perms[s][t].numpy(), B, 1e-3 * 0.95 ** (t // 50), 0.1, 0.9)[0] for t in range(E)]
        U, V, sse = got[0][s]
        assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, (s, rel(U, st.U), rel(V, st.V))
        np.testing.assert_allclose(np.sqrt(sse / len(p[0])), losses, rtol=2e-5)
        assert torch.equal(U, got[1][s][0]) and torch.equal(V, got[1][s][1]) and np.array_equal(sse, got[1][s][2])


@pytest.mark.parametrize('B,heavy,split', [(60, 16, 384), (60, 16, 8), (600, 16, 8), (600, 4, 2), (2900, 1, 1)])
def test_touch_index_mode_heavy_and_split_rows(B, heavy, split, monkeypatch):
    """One item with 5,000 of the shard's 5,600 interactions.  Its runs take a whole workgroup per step (heavy), or -- split -- one
    workgroup per 256 slots of the step whose partial sums a second launch adds in part order: B = 600 gives it ~535 slots per step
    (three parts), B = 2,900 about 2,600 (eleven parts: the top item of the 25 M set); thresholds of 1 put every row of the schedule's
    first 256 into the workgroup classes.  Against the C oracle, with snapshots at every epoch end."""
    from ultrare_amd import engine, rng
    monkeypatch.setattr(engine, 'INDEX_HEAVY_SLOTS', heavy)
    monkeypatch.setattr(engine, 'INDEX_SPLIT_SLOTS', split)
    rs = np.random.RandomState(3)
    n_user, n_item, k, E = 5200, 40, 16, 3
    heavy_users = rs.permutation(n_user)[:5000]
    u = np.concatenate([heavy_users, rs.randint(0, n_user, 600)])
    i = np.concatenate([np.zeros(5000, dtype=np.int64), rs.randint(1, n_item, 600)])
    key = np.unique(u.astype(np.int64) * n_item + i)
    u, i = (key // n_item).astype(np.int32), (key % n_item).astype(np.int32)
    r = (rs.randint(1, 6, len(u)) / 5).astype(np.float32)
    part = (u, i, r)
    torch.manual_seed(21)
    init = tuple(t * 0.3 for t in rng.mf_init(n_user, n_item, k))
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(u))
    sh = engine.ShardData(*part, n_user, n_item)
    job = engine.TrainJob([sh], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch='index', snapshots='compact')
    assert job.index
    st = O.MFState(init[0].numpy().copy(), init[1].numpy().copy())
    rows = torch.as_tensor(sh._sched_host[:sh.n_active, 0].astype(np.int64)).to(sh.device)
    for t in range(E):
        job.run_epochs(1)
        loss = O.train_epoch(st, part, perms[t].numpy(), B, 1e-3, 0.1, 0.9)[0]
        U, V = job.tables(0)
        assert np.isfinite(st.V).all()
        assert rel(U, st.U) < 2e-5 and rel(V, st.V) < 2e-5, (t, rel(U, st.U), rel(V, st.V))
        np.testing.assert_allclose(np.sqrt(job.epoch_sse(0)[t] / len(u)), loss, rtol=2e-5)
        full = torch.cat([job.padded_tables(0)[0], job.padded_tables(0)[1]])
        assert torch.equal(job.state[0]['snap'][t], full[rows])
    per_step = job.touch_rows_per_step()[0]
    steps = (len(u) + B - 1) // B
    pairs = sum(len(np.unique(np.concatenate([u[perms[E - 1].numpy()[s0:s0 + B]], n_user + i[perms[E - 1].numpy()[s0:s0 + B]]]))) for s0 in range(0, len(u), B))
    assert abs(per_step * steps - pairs) < 0.5, (per_step * steps, pairs)          # the index holds exactly the (row, step) pairs of the last epoch
    job.close()


@pytest.mark.parametrize('B,k', [(437, 16), (37, 16), (219, 128)])
def test_touch_index_staged_scatter_equals_the_direct_one(B, k, monkeypatch):
    """Epochs of 65 / 130 / 750 steps in touch_mode 3: the scatter that sorts a chunk of 4,096 slots in LDS and stores a step's stretch at a
    time (idx_scatter_staged_kernel, the default since round 5) puts every slot exactly where the record-by-record scatter
    (idx_scatter_kernel, URE_INDEX_STAGED=0) puts it: the same index (ure_job_index_read), and tables, losses bit for bit."""
    from ultrare_amd import engine, rng
    monkeypatch.setenv('URE_TOUCH_INDEX', '1')
    raw = O.load_csv(TRAIN)
    part = O.partition(*raw, O.uniform_groups(N_USER, 1))[0]
    if B == 37:
        part = _truncate(part, 27714)
    E = 2
    torch.manual_seed(11)
    init = tuple(t * (0.3 if k > 64 else 1.0) for t in rng.mf_init(N_USER, N_ITEM, k))
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(part[0]))
    got = []
    for staged in ('1', '0'):
        monkeypatch.setenv('URE_INDEX_STAGED', staged)
        job = engine.TrainJob([engine.ShardData(*part, N_USER, N_ITEM)], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch=True)
        assert job.index and job.steps_per_epoch(0) > 63
        job.run_epochs(1)
        sb = job.index_array(0, 'step_begin')
        index = (sb, job.index_array(0, 'sslot')[:int(sb[job.steps_per_epoch(0)])], job.index_array(0, 'items'), job.index_array(0, 'step_item'))
        job.run()
        U, V = job.tables(0)
        got.append((U.clone(), V.clone(), job.epoch_sse(0).copy(), index))
        job.close()
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    for a, b in zip(got[0][3], got[1][3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('B', [437, 3000, 37])
def test_touch_index_arrays_equal_the_numpy_restatement(B):
    """The device-built index of every epoch -- sorted slots (steps, opposite ids, ratings, rows, the opposite rows' buffers), runs, items with
    their buffers and gaps (carried by the sorted slots since round 5: idx_own_bits), step_item -- against tools/check_index.py's numpy sort:
    65 steps per epoch (two mask words, the staged scatter), 10 (one word, chunks of 1,024 slots, the compact short scatter), 750 (twelve words)."""
    import importlib.util
    from ultrare_amd import engine, rng
    spec = importlib.util.spec_from_file_location('check_index', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_index.py'))
    CI = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(CI)
    raw = O.load_csv(TRAIN)
    part = O.partition(*raw, O.uniform_groups(N_USER, 1))[0]
    k, E = 16, 3
    torch.manual_seed(11)
    init = rng.mf_init(N_USER, N_ITEM, k)
    perms = rng.epoch_perms(rng.epoch_seeds(E, True), len(part[0]))
    job = engine.TrainJob([engine.ShardData(*part, N_USER, N_ITEM)], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch='index')
    assert job.index
    steps = job.steps_per_epoch(0)
    for e in range(E):
        job.run(1)
        assert CI.check(job, 0, part, perms[e].numpy(), B) > 0
        job.run(steps - 1)
    job.close()


def test_touch_index_mode_refusals():
    from ultrare_amd import engine, _native as nv
    parts, inits, perms, shards = _setup(1, 16, 20, 1)
    assert (len(parts[0][0]) + 19) // 20 > engine.INDEX_MAX_STEPS
    job = engine.TrainJob(shards, inits, perms, 16, 20, 1, 1e-3, 0.1, 0.9, 0.95, touch='index')
    assert job.touch and not job.index                 # more than 1,008 steps per epoch: windows
    job.close()
