"""Model inits made on the device (csrc/mf_init.hip, csrc/normal_math.h) against torch's own CPU `tensor.normal_()` -- what
MF.init_weight draws (utils.py:31-40) -- bit for bit, and the operator surface with and without them."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_fills(state, skip, nu, nv):
    g = torch.Generator()
    g.set_state(state)
    for n in skip:
        torch.empty(n).normal_(0, 1, generator=g)
    U = torch.empty(nu).normal_(0, 1, generator=g) if nu else torch.empty(0)
    V = torch.empty(nv).normal_(0, 1, generator=g) if nv else torch.empty(0)
    return U, V, g.get_state()


def _states():
    """Generator states at the start of a block (freshly seeded: the first draw regenerates), in the middle of one, on its last
    output and right behind it."""
    out = []
    g = torch.Generator()
    g.manual_seed(42)
    out.append(g.get_state().clone())
    for pre in (333, 623, 624, 625, 624 * 7 + 1):
        g.manual_seed(1000 + pre)
        torch.empty(pre, dtype=torch.int32).random_(generator=g)
        out.append(g.get_state().clone())
    return out


@pytest.mark.parametrize('nu,nv', [(16, 16), (1616, 41), (6040 * 32, 3416 * 32), (1508 * 16 + 3, 2071 * 5), (640_000, 17), (0, 4099), (1_300_001, 650_000)])
def test_device_fill_is_torchs_normal_bit_for_bit(nu, nv):
    """Every state class x table lengths with and without the re-drawn tail, inside one segment and over several (1,300,001 + 650,000
    values: four segments, a jump tree of two levels), the discarded fills skipped by the native call: tables and end states."""
    from ultrare_amd import rng
    dev = torch.device('cuda', 0)
    states = _states()
    skip = [n for n in (nu, nv) if n]                        # the constructors' two discarded fills
    skip_draws = sum(rng.fill_draws(n) for n in skip)
    mine = [s.clone() for s in states]
    got = rng.mf_init_device(mine, nu, nv, skip_draws, dev, keep_states=True)
    torch.cuda.synchronize()
    for s, (Ud, Vd, end) in zip(states, got):
        U, V, want_end = _torch_fills(s, skip, nu, nv)
        assert torch.equal(Ud.cpu().view(torch.int32), U.view(torch.int32))
        assert torch.equal(Vd.cpu().view(torch.int32), V.view(torch.int32))
        assert torch.equal(end, want_end)


def test_device_fill_many_shards_and_a_deep_tree():
    """32 shards of one request in one call, and one table of 9 segments (four tree levels, the last one partial)."""
    from ultrare_amd import rng
    dev = torch.device('cuda', 0)
    g = torch.Generator()
    g.manual_seed(7)
    torch.empty(100).normal_(generator=g)
    nu, nv = 30_000 * 16, 11_000 * 16 + 7
    states, want = [], []
    for _ in range(32):
        states.append(g.get_state().clone())
        want.append((torch.empty(nu).normal_(0, 1, generator=g), torch.empty(nv).normal_(0, 1, generator=g)))
    got = rng.mf_init_device([s.clone() for s in states], nu, nv, 0, dev)
    torch.cuda.synchronize()
    for (U, V), (Ud, Vd) in zip(want, got):
        assert torch.equal(Ud.cpu().view(torch.int32), U.view(torch.int32)) and torch.equal(Vd.cpu().view(torch.int32), V.view(torch.int32))
    nu = 8 * 1024 * 624 + 12_345
    g.manual_seed(99)
    st = g.get_state().clone()
    U = torch.empty(nu).normal_(0, 1, generator=g)
    (Ud, Vd), = rng.mf_init_device([st], nu, 0, 0, dev)
    torch.cuda.synchronize()
    assert Vd.numel() == 0 and torch.equal(Ud.cpu().view(torch.int32), U.view(torch.int32))
    assert torch.equal(st, g.get_state())


def test_device_fill_check_and_refusals():
    from ultrare_amd import _native as nv, rng
    dev = torch.device('cuda', 0)
    rng._DEVICE_FILL.clear()
    assert rng.device_fill_ok(dev)                       # this build, this torch, this device: the product's gate
    L = nv.lib()
    st = torch.get_rng_state().clone()
    buf = torch.empty(64, device=dev)
    scratch = torch.empty(int(L.ure_device_mf_init_scratch(1, 16, 16)), dtype=torch.int32, device=dev)
    args = lambda nu, nv_, words: (1, (ctypes.c_void_p * 1)(st.data_ptr()), st.numel(), (ctypes.c_int64 * 1)(0), (ctypes.c_void_p * 1)(buf.data_ptr()), nu,
                                   (ctypes.c_void_p * 1)(buf.data_ptr()), nv_, scratch.data_ptr(), words, 1, None)
    assert L.ure_device_mf_init(*args(15, 16, scratch.numel())) != 0          # ATen's scalar path is not restated
    assert L.ure_device_mf_init(*args(16, 16, 10)) != 0                       # scratch too small
    assert torch.equal(st, torch.get_rng_state())
    torch.cuda.synchronize()


def test_sisa_request_with_device_inits_equals_host_inits_bitwise(tmp_path, monkeypatch):
    """Sisa(parallel).learn with the shards' inits made on the device (rng.DEVICE_INIT_MIN_NORMALS = 0) against the host's batch fill: models,
    merged matrix and every log series equal to the last bit; and Scratch.train likewise."""
    from test_gpu_surface import Param, _loaders, _sisa_inputs
    from ultrare_amd import rng
    from ultrare_amd.method.scratch import Scratch
    from ultrare_amd.method.sisa import Sisa
    from ultrare_amd.read import readRating
    from test_gpu_surface import TRAIN, TEST, N_USER
    idx, trd, ted, tot = _sisa_inputs(3)
    res = []
    for floor in (0, 1 << 40):
        monkeypatch.setattr(rng, 'DEVICE_INIT_MIN_NORMALS', floor)
        before = rng.STATS.get('device_normals', 0)
        s = Sisa(Param(3, parallel=True), 'mf', 3, idx)
        torch.manual_seed(42)
        ml = s.learn(trd, ted, tot, 0, '')
        assert (rng.STATS.get('device_normals', 0) > before) == (floor == 0)
        res.append(([m.item_mat.weight.detach().cpu().clone() for m in ml], ml[0].user_mat.weight.detach().cpu().clone(), dict(s.log), dict(s.log0)))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[1][1]) and res[0][3] == res[1][3]
    for key in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr'):
        assert res[0][2][key] == res[1][2][key], key
    tr, idx1 = readRating(TRAIN, N_USER, 5, [], [], 1, [])
    te, _ = readRating(TEST, N_USER, 5, [], [], 1, idx1)
    out = []
    for floor in (0, 1 << 40):
        monkeypatch.setattr(rng, 'DEVICE_INIT_MIN_NORMALS', floor)
        train, test = _loaders(tr[0], te[0], 3000)
        sc = Scratch(Param(2), 'mf')
        torch.manual_seed(42)
        m = sc.train(train, test, [], 0, '')
        out.append((m.user_mat.weight.detach().cpu().clone(), m.item_mat.weight.detach().cpu().clone(), list(sc.log['train_loss'])))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
