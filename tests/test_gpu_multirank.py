"""The N > 1 path end to end on hardware: two ranks (gloo process group, both on the one
visible GPU) run Sisa.learn / Sisa.unlearn shard-parallel -- shards placed by LPT, every rank
replaying the RNG stream, isolated training, one broadcast of the trained tables, per-epoch logs
gathered -- and must reproduce the single-process results and the reference goldens.
(The production backend is RCCL; only the transport differs here.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')

_WORKER = r'''
import copy, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ultrare_amd.method.sisa import Sisa, assign_shards
from ultrare_amd.read import RatingData, loadData, readRating
out, S, E = sys.argv[2], 4, 3
G = os.path.join(sys.argv[1], 'tests', 'golden')
TRAIN, TEST = os.path.join(G, 'toy', '0_train.csv'), os.path.join(G, 'toy', '0_test.csv')
torch.cuda.set_device(0)
dist.init_process_group('gloo')
rank = dist.get_rank()

class P:
    k, lam, seed, batch, lr, lr_decay, momentum, epochs, parallel = 16, 0.1, 42, 3000, 0.001, 0.95, 0.9, E, True
    n_user, n_item = 1508, 2071

def inputs(dels):
    tr, idx = readRating(TRAIN, 1508, 5, dels, [], S, [])
    te, _ = readRating(TEST, 1508, 5, [], [], S, idx)
    return (idx, [loadData(RatingData(a), 3000, 24) for a in tr], [loadData(RatingData(a), 3000, 24, False) for a in te],
            loadData(RatingData(np.hstack(te)), 3000, 24, False))

idx, trd, ted, tot = inputs([])
save = os.path.join(out, f'rank{rank}')
os.makedirs(save, exist_ok=True)
s = Sisa(P, 'mf', S, idx)
torch.manual_seed(42)
ml = s.learn(trd, ted, tot, 0, save)
g = np.load(os.path.join(G, 'sisa_toy.npz'))
dels = g['S4_unB_del_user'].tolist()
idx2, trd2, ted2, tot2 = inputs(dels)
s2 = Sisa(P, 'mf', S, idx2)
torch.manual_seed(42)
ml2 = s2.unlearn([copy.deepcopy(m) for m in ml], trd2, ted2, tot2, dels, 0, save)
np.savez(os.path.join(out, f'res{rank}.npz'),
         owner=assign_shards([len(d.dataset) for d in trd], 2),
         merged=ml[0].user_mat.weight.detach().cpu().numpy(), un_merged=ml2[0].user_mat.weight.detach().cpu().numpy(),
         log0=[s.log0['total_rmse'], s.log0['total_ndcg'], s.log0['total_hr']],
         un_log0=[s2.log0['total_rmse'], s2.log0['total_ndcg'], s2.log0['total_hr']],
         **{f'V{i}': ml[i].item_mat.weight.detach().cpu().numpy() for i in range(S)},
         **{f'unV{i}': ml2[i].item_mat.weight.detach().cpu().numpy() for i in range(S)},
         **{'log_' + k: np.asarray(v, dtype=np.float64) for k, v in s.log.items() if k != 'time'},
         **{'unlog_' + k: np.asarray(v, dtype=np.float64) for k, v in s2.log.items() if k != 'time'})
dist.destroy_process_group()
'''


def test_two_ranks_share_one_gpu(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=500) == 0
    r0, r1 = np.load(tmp_path / 'res0.npz'), np.load(tmp_path / 'res1.npz')
    g = np.load(os.path.join(G, 'sisa_toy.npz'))
    assert sorted(set(r0['owner'].tolist())) == [0, 1]                     # both ranks trained something
    for key in r0.files:                                                   # every rank ends with the same state
        assert np.array_equal(r0[key], r1[key]), key

    def rel(a, b):
        return float(np.abs(a - b).max() / np.abs(b).max())
    for i in range(4):
        assert rel(r0[f'V{i}'], g[f'S4_learn_V{i}']) < 1e-4
        assert rel(r0[f'unV{i}'], g[f'S4_unB_V{i}']) < 1e-4
    assert rel(r0['merged'], g['S4_learn_Umerged']) < 1e-4 and rel(r0['un_merged'], g['S4_unB_Umerged']) < 1e-4
    np.testing.assert_allclose(r0['log0'], g['S4_learn_log0'], rtol=1e-4)
    np.testing.assert_allclose(r0['un_log0'], g['S4_unB_log0'], rtol=1e-4)
    for key in ('train_loss', 'test_rmse', 'test_ndcg', 'test_hr', 'total_rmse', 'total_ndcg', 'total_hr'):
        np.testing.assert_allclose(r0['log_' + key], g['S4_learn_log_' + key], rtol=1e-4, err_msg=key)
    for key in ('train_loss', 'test_rmse', 'total_rmse', 'total_ndcg', 'total_hr'):
        np.testing.assert_allclose(r0['unlog_' + key], g['S4_unB_log_' + key], rtol=1e-4, err_msg=key)
    # only rank 0 writes artifacts (scratch.py:131-144)
    assert os.path.exists(tmp_path / 'rank0' / 'user_mat4.npy') and os.path.exists(tmp_path / 'rank0' / 'log0.npy')
    assert not os.path.exists(tmp_path / 'rank1' / 'user_mat4.npy')


def _bench(args, timeout=900):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines                 # the driver contract: ONE JSON line
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher: the script starts both ranks (here over gloo, both on
    the one visible GPU), rank 0 prints the one JSON line with n_gpus == 2, the exchange leg ran the
    all-gather, and the configs[2] split placed ONE job's 8 shards over both ranks."""
    out = _bench(['--gpus', '2', '--backend', 'gloo', '--force-device', '0', '--steps', '2', '--warmup', '1',
                  '--roofline-steps', '1', '--splits', 'config2'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['interactions_timed'] > 2 * 2 * 5 * 100000          # both ranks' shards were counted
    ex = out['exchange']
    assert ex['received_ok'] and ex['shards'] == 10 and ex['bytes_gathered'] == 2 * ex['bytes_per_rank']
    c2 = out['north_star_splits']['config2']
    assert c2['scaling'] == 'strong' and c2['shards_per_rank'] == [4, 4] and c2['value'] > 0
    assert out['roofline']['frac'] > 0 and out['roofline']['source_hash']


def test_bench_rccl_calls_on_one_rank():
    """The RCCL half of the N > 1 path on a one-GPU box: a one-rank nccl process group runs the same
    init / barrier / all_reduce / all_gather_into_tensor calls the 8-GPU run makes."""
    out = _bench(['--gpus', '1', '--force-dist', '--steps', '2', '--warmup', '1', '--roofline-steps', '1', '--splits', 'config2'])
    assert out['n_gpus'] == 1 and out['exchange']['backend'] == 'nccl' and out['exchange']['received_ok']
    assert out['north_star_splits']['config2']['shards_per_rank'] == [8]


def test_bench_split_mode_headline():
    out = _bench(['--gpus', '2', '--backend', 'gloo', '--force-device', '0', '--steps', '2', '--warmup', '1', '--roofline-steps', '1',
                  '--split', '1', '--shards', '8', '--d', '64', '--splits', 'none'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['shards_per_gpu'] == 4


def test_cli_under_two_ranks_writes_one_set_of_artifacts(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 main.py --group 3` (here: two processes with the launcher's
    environment, gloo transport, one GPU): rank 0 writes param.pkl / deletion.npy / the OT label cache atomically, the
    other rank reads the cache after a barrier, --parallel 1 is implied, the shards are spread over the ranks and ONE
    set of model files appears -- with the final test of a single-process run of the same command."""
    import shutil
    import pickle
    data = tmp_path / 'data'
    (data / 'toy').mkdir(parents=True)
    shutil.copy(os.path.join(G, 'toy', '0_train.csv'), data / 'toy' / '0_train.csv')
    shutil.copy(os.path.join(G, 'toy', '0_test.csv'), data / 'toy' / '0_test.csv')
    common = ['--dataset', 'toy', '--epoch', '2', '--verbose', '0', '--data-dir', str(data)]
    clean = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    cli = [sys.executable, os.path.join(ROOT, 'main.py')]
    # the full-MF stage (its user matrix is what the OT grouping clusters), then --group 3 alone as the reference of the test
    for save in ('one', 'two'):
        subprocess.run(cli + common + ['--group', '0', '--save-dir', str(tmp_path / save)], env=clean, check=True, stdout=subprocess.DEVNULL, timeout=600)
    subprocess.run(cli + common + ['--group', '3', '--parallel', '1', '--save-dir', str(tmp_path / 'one')], env=clean, check=True,
                   stdout=subprocess.DEVNULL, timeout=600)
    os.remove(data / 'toy' / 'val' / 'emb-ot3.npy')          # the two-rank run must build (and share) the label cache itself
    env = dict(clean, MASTER_ADDR='127.0.0.1', MASTER_PORT='29647', WORLD_SIZE='2', URE_DIST_BACKEND='gloo')
    procs = [subprocess.Popen(cli + common + ['--group', '3', '--save-dir', str(tmp_path / 'two')], env=dict(env, RANK=str(r), LOCAL_RANK='0'),
                              stdout=subprocess.DEVNULL) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    one, two = tmp_path / 'one' / '2' / 'rand' / 'toy_g3', tmp_path / 'two' / '2' / 'rand' / 'toy_g3'
    assert (two / 'param.pkl').exists() and pickle.load(open(two / 'param.pkl', 'rb')).n_group == 3
    assert not [f for f in os.listdir(two) if '.tmp' in f] and not [f for f in os.listdir(data / 'toy' / 'val') if '.tmp' in f]
    for stage in ('MF_emb-ot_sisa_learn', 'MF_emb-ot_sisa_unlearn'):
        a = np.load(one / stage / 'log0.npy', allow_pickle=True).item()
        b = np.load(two / stage / 'log0.npy', allow_pickle=True).item()
        for key in ('total_rmse', 'total_ndcg', 'total_hr'):
            assert abs(a[key] - b[key]) <= 1e-6 * abs(a[key]), (stage, key)
        for i in (1, 2, 3):
            np.testing.assert_allclose(np.load(two / stage / f'user_mat{i}.npy'), np.load(one / stage / f'user_mat{i}.npy'), rtol=2e-5, atol=1e-6)
            assert np.array_equal(np.load(two / stage / f'item_mat{i}.npy'), np.load(one / stage / f'item_mat{i}.npy'))
