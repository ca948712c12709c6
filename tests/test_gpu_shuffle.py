"""csrc/perm_chain.hip (ure_device_shuffle_tags): an epoch's batch tags -- read.py:127-133, the RandomSampler's torch.randperm of the
epoch -- made on the device by many workgroups per permutation and for any number of rows, against ure_host_randperm_tags (itself
pinned to torch.randperm in tests/test_cpu_host.py), bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cases, reps_of, seed=7, scratch_fill=0, range_log2=0):
    from ultrare_amd import _native as nv, rng
    L = nv.lib()
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(seed)
    table, want, outs = [], [], []
    for n, batch in cases:
        reps = reps_of(n)
        seeds = rs.randint(0, 2 ** 62, size=reps).astype(np.int64)
        host = torch.empty(reps, n, dtype=torch.int16)
        nv.check(L.ure_host_randperm_tags(seeds.ctypes.data, reps, n, batch, host.data_ptr(), 0), 'ure_host_randperm_tags')
        out = torch.full((reps, n), -1, dtype=torch.int16, device=dev)
        for r in range(reps):
            table.append((int(seeds[r]), out.data_ptr() + 2 * n * r, n, batch))
        want.append(host)
        outs.append(out)
    tab = np.array(table, dtype=rng.PERM_DTYPE)
    tab_d = torch.from_numpy(tab.view(np.uint8)).to(dev)
    n_max = max(n for n, _ in cases)
    words = int(L.ure_device_shuffle_tags_scratch(n_max, len(tab)))
    # (whatever an earlier call left in the scratch must not matter: the link values of a previous, longer permutation above all)
    scratch = torch.full((words,), scratch_fill, dtype=torch.int32, device=dev)
    flag_at = int(L.ure_device_shuffle_tags_flag(n_max, len(tab)))
    scratch[flag_at] = 0
    for _ in range(2):                                       # the second call runs over the first one's leavings
        for o in outs:
            o.fill_(-1)
        nv.check(L.ure_device_shuffle_tags(tab_d.data_ptr(), len(tab), n_max, scratch.data_ptr(), words, range_log2, nv.stream_handle()), 'ure_device_shuffle_tags')
        torch.cuda.synchronize()
        for (n, batch), host, out in zip(cases, want, outs):
            got = out.cpu()
            assert torch.equal(got, host), (n, batch, int((got != host).sum()))
    assert int(scratch[flag_at]) == 0
    return L, tab_d, len(tab), n_max, scratch, words


@pytest.mark.parametrize('fill,range_log2', [(0, 0), (-1, 11), (0x12345678, 14), (7, 12)])
def test_shuffle_tags_equal_the_hosts(fill, range_log2):
    """Shards of different sizes in ONE table: 1, 2, 3 rows, around the generator's 624-output block and the eight-block pass (4,992
    outputs), around the 32,768 targets of a link workgroup and the 4,096 rows of a resolve workgroup, the sizes of BASELINE.json's
    shards, 2^20 and 2^20 + 1 (where perm_tags.hip stops); seeds of 62 bits; several batch sizes; more permutations than XCDs."""
    cases = [(1, 1), (2, 1), (3, 2), (4, 1), (5, 3), (623, 100), (624, 7), (625, 624), (626, 1), (4992, 30000), (4993, 17), (4994, 30000), (5000, 30000),
             (32767, 30000), (32768, 1000), (32769, 30000), (32770, 555), (65537, 4097), (56321, 30000), (179718, 30000), ((1 << 18) + 5, 30000),
             (1 << 20, 30000), ((1 << 20) + 1, 30000)]
    L, tab_d, n_tab, n_max, scratch, words = _run(cases, lambda n: 1 if n > 100000 else 3, scratch_fill=fill, range_log2=range_log2)
    # refusals: scratch too small, 2^31 rows
    assert L.ure_device_shuffle_tags(tab_d.data_ptr(), n_tab, n_max, scratch.data_ptr(), 16, 0, None) != 0
    assert L.ure_device_shuffle_tags(tab_d.data_ptr(), 1, (1 << 28) + 1, scratch.data_ptr(), 1 << 40, 0, None) != 0
    assert L.ure_device_shuffle_tags(tab_d.data_ptr(), 1, (1 << 26) + 1, scratch.data_ptr(), 1 << 40, 12, None) != 0
    assert L.ure_device_shuffle_tags(tab_d.data_ptr(), n_tab, n_max, scratch.data_ptr(), words, 13, None) != 0


def test_shuffle_tags_many_permutations_of_a_request():
    """A request's shape: 5 shards x 12 epochs, every shard its own size, in (shard, epoch) order -- 60 permutations over the eight XCD slots."""
    sizes = [179718, 184837, 171049, 170284, 191026]
    cases = [(n, 30000) for n in sizes]
    _run(cases, lambda n: 12, seed=11)


@pytest.mark.parametrize('sizes', [[4 * 1024 * 1024 + 3], [22_500_000], [3_000_001, 2_200_000, 700_000, 638_977, 5]])
def test_shuffle_tags_beyond_2_to_20_rows(sizes):
    """config.py:182-188's full-MF run at the 25 M shape shuffles 22.5 M rows per epoch (750 steps of 30,000): the path perm_tags.hip
    refuses and the host made with one sequential Fisher-Yates per epoch.  Beyond 2^21 rows the generator's stream is cut into segments of
    1,024 blocks whose start blocks come from the device-side jump tree (csrc/mt_jump_dev.h): permutations of several lengths in one
    launch, one of them a word longer than a segment, one of a few rows."""
    n = max(sizes)
    _run([(x, 30000) for x in sizes], lambda n: 1, seed=3, range_log2=14 if n < 5_000_000 and len(sizes) == 1 else 0)
