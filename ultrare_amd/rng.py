"""RNG stream contract of the reference, host logic.

Everything random on the hot path comes from torch's process-global CPU generator
(model init, DataLoader base seeds, RandomSampler seeds) plus numpy's global state
(grouping, deletion set).  To match the reference for a fixed seed the engine must
consume those streams draw for draw; the draws are data independent, so they can
be taken up front and the permutations expanded later (or on other threads).

  method/utils.py:31-40   two nn.Embedding constructors + init_weight  -> mf_init
  torch DataLoader        _base_seed per iterator                      -> draw_seed
  torch RandomSampler     seed -> Generator -> randperm(N)             -> epoch_perm
"""
import ctypes
import threading

import numpy as np
import torch


def host_cpus():
    """CPUs this rank may use: the process's affinity mask (not the machine's CPU count) shared among the ranks the
    launcher started on this host (LOCAL_WORLD_SIZE)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # a container's CPU quota (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us): the 1-GPU boxes of the pool this was
    # measured on show 256 CPUs and grant 16; threads beyond the quota only get throttled (tools/probe_host.py)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)))


def limit_torch_threads():
    """torch sizes its intra-op thread pool by the machine's CPU count and knows nothing of a container's CPU quota: on a
    256-CPU host that grants this container 16, every CPU-side torch op (a normal_() fill, a copy) woke 256 OpenMP threads,
    whose spinning used the cgroup's quota up and got the whole process throttled for the rest of the 100 ms period --
    50-80 ms stalls in the middle of a 15 ms Sisa.learn (cpu.stat: 28 of 285 periods throttled; profiles/r03/NOTES.md).
    The pool is capped at 4 threads (and at the CPUs this rank may really use, host_cpus()): the CPU-side torch ops of the path are
    small (fills of ~200 k normals, copies) and run on several worker threads at once, where a wide intra-op pool only adds hand-offs
    -- medians of alternating runs (tools/ab_host.py), learn / unlearn at 5 shards: 16.4 / 14.3 ms with 16 threads, 14.6 / 13.1 with 4,
    15.2 / 14.0 with 1; at 16 shards 24.3 / 27.0, 27.6 / 24.8, 22.5 / 20.2.  URE_TORCH_THREADS=0 leaves torch alone, =n sets n."""
    import os
    env = os.environ.get('URE_TORCH_THREADS', '')
    if env == '0':
        return torch.get_num_threads()
    want = int(env) if env else min(torch.get_num_threads(), host_cpus(), 4)
    if want >= 1 and want != torch.get_num_threads():
        torch.set_num_threads(want)
    return torch.get_num_threads()


_THREADS_LOCK = threading.Lock()
_THREADS_DEPTH = [0, None]        # nesting depth of torch_threads(), torch's thread count before the outermost one


class torch_threads:
    """Context of one request (Scratch.train, Sisa.learn / unlearn): torch's intra-op pool capped as limit_torch_threads() describes,
    and set back to what the host program had when the outermost request ends -- importing this package changes nothing
    process-global.  Nested requests (Sisa -> Scratch) share one cap."""

    def __enter__(self):
        with _THREADS_LOCK:
            if _THREADS_DEPTH[0] == 0:
                _THREADS_DEPTH[1] = torch.get_num_threads()
                limit_torch_threads()
            _THREADS_DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        with _THREADS_LOCK:
            _THREADS_DEPTH[0] -= 1
            if _THREADS_DEPTH[0] == 0 and _THREADS_DEPTH[1] is not None and torch.get_num_threads() != _THREADS_DEPTH[1]:
                torch.set_num_threads(_THREADS_DEPTH[1])
        return False


def perm_threads():
    """Host threads of the permutation expander per SISA call: URE_PERM_THREADS, else every CPU of the rank up to 64
    (250 Fisher-Yates permutations of 0.5 ms each are half of a 5-shard, 50-epoch call's wall time on 16 threads)."""
    import os
    env = os.environ.get('URE_PERM_THREADS')
    # (the rank's CPUs, not more: with 32 threads on a 16-CPU grant the isolated expansion is faster -- 2.3 against 3.0 ms per 250
    # permutations -- and the whole call slower, medians of 5 alternating runs: learn 17.3 against 15.2 ms; tools/ab_host.py)
    return max(1, int(env)) if env else min(64, host_cpus())


def fill_draws(n):
    """32-bit MT19937 outputs one `tensor.normal_()` of n >= 16 float32 elements consumes on the CPU: ATen fills
    the tensor with n uniforms and turns them into normals 16 at a time; when 16 does not divide n the last 16
    values are drawn again (checked against the generator state in tests/test_cpu_host.py)."""
    assert n >= 16
    return n + (16 if n % 16 else 0)


STATS = {'normals': 0, 'skipped_draws': 0}      # host work of this process: N(0, 1) values computed / generator outputs skipped
_STATS_LOCK = threading.Lock()                  # (the model inits of a request run on a worker each)


def _count(**kw):
    with _STATS_LOCK:
        for k, v in kw.items():
            STATS[k] = STATS.get(k, 0) + int(v)


def advance_state(state, n_draws, count=True):
    """A copy of a torch CPU generator state moved past n_draws 32-bit outputs (ure_host_mt_advance).  count=False: a second walk
    over draws that are accounted for elsewhere (STATS counts every output of the stream once)."""
    from . import _native as nv
    if count:
        _count(skipped_draws=n_draws)
    out = state.clone()
    nv.check(nv.lib().ure_host_mt_advance(out.data_ptr(), out.numel(), int(n_draws)), 'ure_host_mt_advance')
    return out


def model_draws(n_user, n_item, k, epochs, with_total_test):
    """32-bit outputs of one Scratch.train call: (the two discarded constructor fills, the two kept fills of
    init_weight, the per-epoch int64 seeds: 2 outputs each).  None when a table has fewer than 16 elements
    (ATen then takes its scalar path, which caches a second normal: such calls are replayed, not skipped)."""
    nu, ni = n_user * k, n_item * k
    if min(nu, ni) < 16:
        return None
    fills = fill_draws(nu) + fill_draws(ni)
    return fills, fills, 2 * epochs * (4 if with_total_test else 3)


_NATIVE_FILL = [None]


def native_fill_ok():
    """Whether ure_host_mf_init reproduces THIS torch build's `tensor.normal_()` bit for bit (csrc/host_rng.cpp, host_normal_avx2.cpp:
    ATen's AVX2 fill through the installed PyTorch's own avx_mathfun kernels), established once per process on three pairs of fills
    from a state in the middle of a generator block -- lengths with and without the redrawn tail, several blocks --: tables and end
    state compared.  URE_NATIVE_FILL=0 keeps torch's fill."""
    if _NATIVE_FILL[0] is None:
        import os
        ok = os.environ.get('URE_NATIVE_FILL', '1') != '0'
        if ok:
            from . import _native as nv
            g = torch.Generator()
            g.manual_seed(20240607)
            torch.empty(5, dtype=torch.int64).random_(generator=g)
            for nu, nv_ in ((1616, 41), (16, 2000), (4099, 4112)):
                st = g.get_state().clone()
                want = [torch.empty(n).normal_(0, 1, generator=g) for n in (nu, nv_)]
                got = [torch.empty(n) for n in (nu, nv_)]
                rc = nv.lib().ure_host_mf_init(st.data_ptr(), st.numel(), 0, got[0].data_ptr(), nu, got[1].data_ptr(), nv_, 2)
                ok = ok and rc == 0 and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(want, got)) and torch.equal(st, g.get_state())
        _NATIVE_FILL[0] = bool(ok)
    return _NATIVE_FILL[0]


_DEVICE_FILL = {}
DEVICE_INIT_MIN_NORMALS = 1 << 20      # requests with fewer N(0, 1) values keep the host's batch fill (an ml-1m shard: 0.3 M)


def device_fill_ok(device):
    """Whether ure_device_mf_init (csrc/mf_init.hip: MT19937 and the Box-Muller arithmetic of csrc/normal_math.h on the device)
    reproduces THIS torch build's CPU `tensor.normal_()` bit for bit on `device`, established once per process: two shards from states in
    the middle of a generator block, a table with the re-drawn tail and one of several segments (a jump tree of two levels) -- tables
    and end states compared.  URE_DEVICE_INIT=0 keeps the host fill."""
    import os
    key = str(device)
    if key not in _DEVICE_FILL:
        ok = os.environ.get('URE_DEVICE_INIT', '1') != '0' and torch.device(device).type == 'cuda'
        if ok:
            try:
                g = torch.Generator()
                g.manual_seed(20240609)
                torch.empty(7, dtype=torch.int64).random_(generator=g)
                nu, nv_ = 1616 * 1024 + 5, 4099
                states, want = [], []
                for _ in range(2):
                    states.append(g.get_state().clone())
                    want.append([torch.empty(n).normal_(0, 1, generator=g) for n in (nu, nv_)] + [g.get_state().clone()])
                got = mf_init_device([st.clone() for st in states], nu, nv_, 0, device, keep_states=True)
                torch.cuda.synchronize(device)
                for (U, V, end), (Ud, Vd, st) in zip(want, got):
                    ok = ok and torch.equal(U.view(torch.int32), Ud.cpu().view(torch.int32)) and torch.equal(V.view(torch.int32), Vd.cpu().view(torch.int32)) \
                        and torch.equal(end, st)
            except Exception:
                ok = False
        _DEVICE_FILL[key] = bool(ok)
    return _DEVICE_FILL[key]


def mf_init_device(states, nu, nv_, skip_draws, device, stream=None, keep_states=False):
    """The kept fills of len(states) models made on `device` (ure_device_mf_init), queued on `stream` (default: the current one):
    states[s] = the torch CPU generator state of shard s (moved, in place, past skip_draws outputs and the two fills).
    -> [(U0 [nu], V0 [nv_]) device views (+ the state with keep_states)] of ONE allocation; the caller orders its reads behind `stream`."""
    from . import _native as nv
    L = nv.lib()
    S = len(states)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            block = torch.empty((S, nu + nv_), dtype=torch.float32, device=dev)
            words = int(L.ure_device_mf_init_scratch(S, nu, nv_))
            scratch = torch.empty(words, dtype=torch.int32, device=dev)
            base, row = block.data_ptr(), 4 * (nu + nv_)
            st_a = (ctypes.c_void_p * S)(*[x.data_ptr() for x in states])
            u_a = (ctypes.c_void_p * S)(*[base + s * row for s in range(S)])
            v_a = (ctypes.c_void_p * S)(*[base + s * row + 4 * nu for s in range(S)])
            skip = (ctypes.c_int64 * S)(*([int(skip_draws)] * S))
            nv.check(L.ure_device_mf_init(S, st_a, states[0].numel(), skip, u_a, nu, v_a, nv_, scratch.data_ptr(), words, host_cpus(), st.cuda_stream),
                     'ure_device_mf_init')
            del scratch            # (back to this stream's pool: reused only behind the kernels that read it)
    _count(skipped_draws=S * int(skip_draws), device_normals=S * (nu + nv_))
    return [((block[s, :nu], block[s, nu:]) + ((states[s],) if keep_states else ())) for s in range(S)]


def fill_threads(n_normals, sharers=1):
    """Host threads for the 16-blocks of a model init of n_normals values when `sharers` inits run side by side: one for small tables
    (an ml-1m shard: 0.3 M values, 0.5 ms), up to eight for big ones (the 25 M shape at d = 128: 28 M values)."""
    return 1 if n_normals < (2 << 20) else max(1, min(8, host_cpus() // max(1, int(sharers))))


def mf_init(n_user, n_item, k, generator=None, threads=None, device=None):
    """The four N(0,1) fills of `MF(n_user, n_item, k)` (utils.py:31-40).  The first two (the nn.Embedding
    constructors') are overwritten by init_weight: the stream is moved past them without computing them.  The two kept ones come
    from ONE native call (ure_host_mf_init: the uniforms in bulk, the 16-blocks through the installed PyTorch's own kernels on
    `threads` threads) where that reproduces torch's fill bit for bit (native_fill_ok), else from torch itself.
    device: tables of DEVICE_INIT_MIN_NORMALS values or more are made there instead (mf_init_device, queued on the current stream) and
    returned as device tensors."""
    g = generator
    draws = model_draws(n_user, n_item, k, 0, False)
    if device is not None and draws is not None and _device_init_wanted((n_user + n_item) * k, device):
        state = (torch.get_rng_state() if g is None else g.get_state()).clone()
        U0, V0 = mf_init_device([state], n_user * k, n_item * k, draws[0], device)[0]
        (torch.set_rng_state if g is None else g.set_state)(state)
        return U0.view(n_user, k), V0.view(n_item, k)
    if draws is not None and native_fill_ok():
        from . import _native as nv
        state = (torch.get_rng_state() if g is None else g.get_state()).clone()
        U0, V0 = torch.empty(n_user, k), torch.empty(n_item, k)
        threads = fill_threads((n_user + n_item) * k) if threads is None else max(1, int(threads))
        nv.check(nv.lib().ure_host_mf_init(state.data_ptr(), state.numel(), int(draws[0]), U0.data_ptr(), n_user * k, V0.data_ptr(), n_item * k,
                                           threads), 'ure_host_mf_init')
        (torch.set_rng_state if g is None else g.set_state)(state)
        _count(skipped_draws=draws[0], normals=(n_user + n_item) * k)
        return U0, V0
    if draws is None:
        torch.empty(n_user, k).normal_(0, 1, generator=g)
        torch.empty(n_item, k).normal_(0, 1, generator=g)
    elif g is None:
        torch.set_rng_state(advance_state(torch.get_rng_state(), draws[0]))
    else:
        g.set_state(advance_state(g.get_state(), draws[0]))
    U0 = torch.empty(n_user, k).normal_(0, 1, generator=g)
    V0 = torch.empty(n_item, k).normal_(0, 1, generator=g)
    _count(normals=(n_user + n_item) * k * (1 if draws is not None else 2))
    return U0, V0


def skip_model(n_user, n_item, k, epochs, with_total_test):
    """Move the global generator past ALL draws of one Scratch.train call (a shard another rank owns)."""
    draws = model_draws(n_user, n_item, k, epochs, with_total_test)
    if draws is None:
        mf_init(n_user, n_item, k)
        epoch_seeds(epochs, with_total_test)
        return
    torch.set_rng_state(advance_state(torch.get_rng_state(), sum(draws)))


def draw_seed():
    return int(torch.empty((), dtype=torch.int64).random_().item())


def epoch_seeds(epochs, with_total_test, generator=None):
    """Per epoch the reference draws: train loader base seed, sampler seed, group-test
    loader base seed and (SISA only) total-test loader base seed (scratch.py:78-97).
    Returns the sampler seeds; the others only advance the stream."""
    per = 4 if with_total_test else 3
    if epochs <= 0:
        return []
    # one vectorised draw: random_() fills serially from the same generator, so the values are those of
    # `per * epochs` scalar draws (checked in tests/test_cpu_host.py)
    draws = torch.empty(epochs * per, dtype=torch.int64).random_(generator=generator)
    return [int(v) for v in draws[1::per].tolist()]


_STREAM_MEMO = {}


def shard_streams(n_shards, n_user, n_item, k, epochs, with_total_test, want_seeds=False):
    """The generator states at which each of n_shards consecutive Scratch.train calls starts, computed by
    skip-ahead (the draws are data independent), and the state after the last one.  With them every shard's
    draws can be taken on its own thread from its own torch.Generator.  None when skipping is not possible.
    want_seeds: -> (starts, end, seeds) with seeds[s] = the shard's per-epoch sampler seeds (epoch_seeds), read off the stream where
    the walk passes them -- they follow the shard's four fills -- instead of by a second walk per shard later."""
    draws = model_draws(n_user, n_item, k, epochs, with_total_test)
    if draws is None:
        return None
    s = torch.get_rng_state()
    # The states are a pure function of (generator state, the call's fills and seed draws, count).  Long distances are jumped
    # (csrc/mt_jump.cpp: 56.8 M outputs per shard at BASELINE.json configs[3]'s shape in 0.2 ms instead of 19 ms of walking -- 7-9 ms for
    # its 32 shards in a cold process, 0.2-0.6 s in round 4); short ones are walked: ~1 ms for a 5-shard ml-1m request.  Requests
    # that start from a state seen before (the harness seeds the generator before each top-level call: SURVEY D7) take states and
    # seeds from a small memo instead (32 x 5 KB); URE_STREAM_MEMO=0 computes every time.
    import hashlib
    import os
    per = 4 if with_total_test else 3
    key = None
    if os.environ.get('URE_STREAM_MEMO', '1') != '0':
        # (everything the seeds depend on, not just the total distance: where the four fills end, how many draws an epoch makes, how many epochs)
        key = (hashlib.blake2b(s.numpy().tobytes(), digest_size=16).digest(), int(n_shards), int(draws[0]), int(draws[1]), int(draws[2]), per, int(epochs))
        hit = _STREAM_MEMO.get(key)
        if hit is not None:
            _count(memo_hits=1)
            out = [t.clone() for t in hit[0]], hit[1].clone()
            return out + ([list(x) for x in hit[2]],) if want_seeds else out
    from . import _native as nv
    starts, seeds = [], []
    for _ in range(n_shards):
        starts.append(s)
        s = advance_state(s, draws[0] + draws[1])                      # past the four fills ...
        vals = np.empty(max(epochs, 0) * per, dtype=np.int64)          # ... the epochs' int64 draws as `random_()` makes them ...
        nv.check(nv.lib().ure_host_draw_int64(s.data_ptr(), s.numel(), 0, len(vals), vals.ctypes.data), 'ure_host_draw_int64')
        seeds.append(vals[1::per].tolist())
        s = advance_state(s, draws[2])                                  # ... and past them
    if key is not None:
        if len(_STREAM_MEMO) >= 64:
            _STREAM_MEMO.clear()
        _STREAM_MEMO[key] = ([t.clone() for t in starts], s.clone(), [list(x) for x in seeds])
    return (starts, s, seeds) if want_seeds else (starts, s)


def epoch_perm(seed, n):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).to(torch.int32)


class _HostPool:
    """Reusable (pinned when a HIP device is present) host buffers for the epoch permutations.
    A 50-epoch, 5-shard ml-1m job needs 180 MB of them; allocating, first-touching and
    unmapping that much pageable memory per call costs more host time than the whole
    training costs device time, so buffers are kept and handed out again."""

    def __init__(self, min_step=1 << 20):
        self.min_step = min_step
        self.free = []          # [(buffer, event or None)]: a buffer whose last copy is still in flight is not handed out
        self.lent = {}
        self.lock = threading.Lock()

    def take(self, shape, dtype):
        need = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        with self.lock:
            best = None
            for i, (b, ev) in enumerate(self.free):
                if b.numel() >= need and (best is None or b.numel() < self.free[best][0].numel()) and (ev is None or ev.query()):
                    best = i
            buf = self.free.pop(best)[0] if best is not None else None
        if buf is None:
            # size classes (steps of a quarter of the power of two above the size, at least min_step): the shards of consecutive
            # requests differ by a few rows -- another deletion set -- and must find the buffers of the previous request large
            # enough; allocating 36 MB of pinned memory costs more host time than expanding the permutations that go into it
            size = max(need, 1)
            step = max(self.min_step, 1 << max(size.bit_length() - 2, 0))
            size = (size + step - 1) // step * step
            buf = torch.empty(size, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        view = buf[:need].view(dtype).view(shape)
        with self.lock:
            self.lent[view.data_ptr()] = buf
        return view

    def give(self, view, event=None):
        """Hand a buffer back; with `event` (recorded after the last copy that reads it) it is reused only once that is done."""
        with self.lock:
            buf = self.lent.pop(view.data_ptr(), None)
            if buf is not None:
                self.free.append((buf, event))


POOL = _HostPool()           # the epoch permutations
SMALL = _HostPool(1 << 14)   # small descriptors on their way to the device (engine.to_device_async)
STAGING = _HostPool()        # layout staging (engine.build_shards): its own pool, so that a 23 MB request never takes a 36 MB permutation buffer


def release(perms):
    """Hand a permutation buffer from epoch_perms(pooled=True) back (after it was uploaded)."""
    if torch.is_tensor(perms):
        host = getattr(perms, '_ure_host', None)
        if host is not None:                    # uploaded by a background worker: wait for the (last) copy
            chunks = getattr(perms, '_ure_chunks', None)
            if chunks is not None:
                chunks[-1][1].wait()
                if chunks[-1][2][0] is not None:
                    chunks[-1][2][0].synchronize()
            else:
                perms._ure_event.synchronize()
            shared = getattr(perms, '_ure_shared', None)
            if shared is not None:
                shared.drop()                   # (a view of a block the shards of a call share)
            else:
                POOL.give(host)
            perms._ure_host = None
        else:
            POOL.give(perms)


def epoch_perms(seeds, n, threads=0, pooled=False):
    """[len(seeds), n] int32 permutations, perms[t] == torch.randperm(n, generator seeded with
    seeds[t]).  Each epoch has its own generator, so the epochs are expanded concurrently by
    the library's host threads (ure_host_randperm, a restatement of ATen's MT19937
    Fisher-Yates loop checked against torch.randperm in tests/test_cpu_host.py)."""
    from . import _native as nv
    out = POOL.take((len(seeds), n), torch.int32) if pooled else torch.empty(len(seeds), n, dtype=torch.int32)
    if len(seeds) == 0 or n == 0:
        return out
    if n >= (2 ** 32 - 1) // 20:          # ATen switches algorithm for huge n: use torch itself
        for t, s in enumerate(seeds):
            out[t] = epoch_perm(s, n)
        return out
    sd = np.asarray(seeds, dtype=np.uint64).astype(np.int64)
    nv.check(nv.lib().ure_host_randperm(sd.ctypes.data, len(sd), n, out.data_ptr(), int(threads or 0)), 'ure_host_randperm')
    return out


_EXPANDER = None


_UPLOAD_STREAMS = {}


def epoch_tags(seeds, n, batch, threads=0):
    """[len(seeds), n] int16 (uint16 bits): tags[t][perm_t[b]] = b // batch for perm_t = the permutation of epoch_perms -- the
    step of the epoch in which every interaction trains (ure_host_randperm_tags; struct ure_shard: file_tags)."""
    from . import _native as nv
    nv.lib()
    sd = np.asarray(seeds, dtype=np.uint64).astype(np.int64)
    out = torch.empty((len(sd), int(n)), dtype=torch.int16)
    nv.check(nv.lib().ure_host_randperm_tags(sd.ctypes.data, len(sd), int(n), int(batch), out.data_ptr(), int(threads or 0)), 'ure_host_randperm_tags')
    return out


def epoch_tags_device(seeds, n, batch, device, bounds=None, method=None):
    """epoch_tags made on the device (device_tags: no host shuffle, no upload) for ONE shard -- Scratch.train's path (config.py:182-188's
    full-MF run: 896,914 rows per epoch at ml-1m, 22.5 M at the 25 M shape).  -> the [len(seeds), n] int16 tensor on `device`, arriving in
    chunks of epochs TrainJob.run waits for; None when the device path does not apply (URE_DEVICE_TAGS=0, no device, too many rows or steps)."""
    n, batch = int(n), int(batch)
    if (device is None or torch.device(device).type != 'cuda' or not device_tags_wanted() or n < 1 or n > DEVICE_TAGS_MAX_ROWS
            or batch < 1 or -(-n // batch) > 65535 or len(seeds) == 0):
        return None
    t = _task_of(dict(start_state=None, n_user=0, n_item=0, k=0, epochs=len(seeds), with_total_test=True, n_rows=n, shuffle=True,
                      device=torch.device(device), tags_batch=batch, seeds=list(seeds)), buffers=False)
    if not device_tags([t], bounds=bounds, method=method):
        return None
    return t.perms_value


def epoch_perms_async(seeds, n, threads=0, pooled=False, device=None, tags_batch=0):
    """epoch_perms on a background thread (tags_batch = B > 0: the batch tags of those permutations instead, int16: epoch_tags): returns a future whose result() is the tensor.  The seeds
    are already drawn, so expanding them needs nothing from torch's generator and overlaps with the
    caller's next draws (the next shard's model init); the native call runs without the GIL.
    With `device` the worker also uploads the permutations on a side stream as soon as they exist;
    result() is then the DEVICE tensor, carrying `_ure_event` (recorded after the copy: consumers
    make their stream wait for it, engine.TrainJob does) and `_ure_host` (the pinned source,
    handed back by release() once the copy is done)."""
    global _EXPANDER
    if _EXPANDER is None:
        from concurrent.futures import ThreadPoolExecutor
        _EXPANDER = ThreadPoolExecutor(max_workers=1, thread_name_prefix='ure-perms')
    from . import _native as nv
    nv.lib()                                            # load the library on the calling thread
    if len(seeds) == 0 or n == 0 or n >= (2 ** 32 - 1) // 20:
        out = epoch_perms(seeds, n, threads, pooled)    # nothing to expand, or ATen's huge-n algorithm (torch itself): done here

        class _Done:
            def result(self_inner):
                return out
        return _Done()
    tags_batch = int(tags_batch) if 0 < -(-n // max(int(tags_batch), 1)) <= 65535 else 0
    word = torch.int16 if tags_batch else torch.int32
    out = POOL.take((len(seeds), n), word) if pooled else torch.empty(len(seeds), n, dtype=word)
    sd = np.asarray(seeds, dtype=np.uint64).astype(np.int64)

    on_dev = ready = None
    if device is not None and torch.device(device).type == 'cuda':
        # destination allocated here, on the caller's stream (the allocator's pools are per stream);
        # the side stream starts its copy only after whatever the caller's stream was doing with that memory
        dev = torch.device(device)
        on_dev = torch.empty((len(seeds), n), dtype=word, device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))

    def work():
        if tags_batch:
            nv.check(nv.lib().ure_host_randperm_tags(sd.ctypes.data, len(sd), n, tags_batch, out.data_ptr(), int(threads or 0)), 'ure_host_randperm_tags')
        else:
            nv.check(nv.lib().ure_host_randperm(sd.ctypes.data, len(sd), n, out.data_ptr(), int(threads or 0)), 'ure_host_randperm')
        if on_dev is None:
            return out
        dev = on_dev.device
        with torch.cuda.device(dev):
            st = _UPLOAD_STREAMS.get(dev)
            if st is None:
                st = _UPLOAD_STREAMS[dev] = torch.cuda.Stream(dev)
            st.wait_event(ready)
            with torch.cuda.stream(st):
                on_dev.copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
        on_dev._ure_event, on_dev._ure_host = ev, out
        return on_dev
    return _EXPANDER.submit(work)


_SHARD_POOL = None
_TEST_CHUNK_DELAY_S = 0.0      # tests only: the worker sleeps this long before every chunk (late arrivals must not change results)


def worker_pool():
    """The host workers of a SISA call (per-shard draws, layout builds): they spend their time inside native calls and
    copies that release the GIL.  Two per shard of a call up to 64 (8 workers -- round 2 -- left half of a 16-shard call's
    draws waiting behind the other half)."""
    global _SHARD_POOL
    if _SHARD_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _SHARD_POOL = ThreadPoolExecutor(max_workers=max(4, min(64, 2 * host_cpus())), thread_name_prefix='ure-shard')
    return _SHARD_POOL


class _DrawsTask:
    """Everything random of ONE Scratch.train call as two pieces of work for a worker thread: init() -- the model init from a
    generator positioned at the shard's start state, uploaded -- and chunks() -- the expanded permutations, one chunk of epochs
    per step of the generator.  Buffers and events are created on the calling thread (the pool and the allocator see the
    caller's current stream)."""

    def __init__(self, start_state, n_user, n_item, k, epochs, with_total_test, n_rows, shuffle, threads, device, want_perms, chunk_epochs, tags_batch=0,
                 buffers=True):
        from . import _native as nv
        nv.lib()
        self.args = (start_state, n_user, n_item, k, epochs, with_total_test, n_rows, shuffle, int(threads or 0), want_perms)
        self.device = torch.device(device) if device is not None and torch.device(device).type == 'cuda' else None
        self.host = self.on_dev = self.ready = self.stream = None
        self.init_value = self.perms_value = None
        self.init_done, self.error = threading.Event(), None
        self.sharers = 1                            # inits of the same call running beside this one (start_inits)
        self.seeds = None                           # (a caller that knows them -- shard_streams(want_seeds=True) -- sets them: _task_of)
        self._buffer_args = (chunk_epochs, tags_batch)
        if buffers:
            self.make_buffers()

    def _buffer_plan(self):
        """-> (word type, epochs, rows) of the buffers chunks() fills and uploads, or None (no permutations, or not on a device)."""
        _, tags_batch = self._buffer_args
        _, _, _, _, epochs, _, n_rows, shuffle, _, want_perms = self.args
        big = n_rows >= (2 ** 32 - 1) // 20
        # tags_batch = B > 0: the permutations leave the host as BATCH TAGS (uint16 [epochs, n_rows]: the step of the epoch in which
        # every interaction trains; struct ure_shard: file_tags; engine.TrainJob tells them from permutations by their dtype) -- half
        # the bytes on PCIe, and no partition phases on the device.  Only on the chunked device path.
        self.tags_batch = int(tags_batch) if (self.device is not None and 0 < -(-n_rows // max(int(tags_batch), 1)) <= 65535) else 0
        if not (want_perms and shuffle and n_rows > 0 and epochs > 0 and not big):
            return None
        return (torch.int16 if self.tags_batch else torch.int32), epochs, n_rows

    def _adopt(self, host, on_dev, ready, shared=None):
        """Take the buffers (views of a block all shards of a call share, or this shard's own)."""
        chunk_epochs, _ = self._buffer_args
        _, _, _, _, epochs, _, n_rows = self.args[:7]
        self.host, self.on_dev, self.ready = host, on_dev, ready
        if on_dev is None:
            return
        on_dev._ure_host, on_dev._ure_shared = host, shared
        # chunks of at least chunk_epochs epochs and ~4 MB: every chunk costs its worker ~0.1 ms of Python (slices, a copy,
        # an event) under the GIL, and a request of 16 small shards had 112 of them competing with the calling thread
        chunk_epochs = max(int(chunk_epochs), -(-(4 << 20) // ((2 if self.tags_batch else 4) * n_rows)))
        on_dev._ure_chunks = [(min(epochs, c0 + chunk_epochs), threading.Event(), [None]) for c0 in range(0, epochs, chunk_epochs)]
        if self.error is not None:                  # the init failed before the buffers were there: never leave a consumer waiting
            for _, flag, _ in on_dev._ure_chunks:
                flag.set()

    def make_buffers(self):
        """The permutations' host and device buffers and events, on the CALLING thread (the pool and the allocator see the caller's
        current stream).  init() does not need them (start_inits), chunks() does."""
        plan = self._buffer_plan()
        if plan is None:
            return
        word, epochs, n_rows = plan
        host = POOL.take((epochs, n_rows), word)
        on_dev = ready = None
        if self.device is not None:
            on_dev = torch.empty((epochs, n_rows), dtype=word, device=self.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        self._adopt(host, on_dev, ready)

    def fail(self, exc):
        """Never leave a consumer waiting: init() / the chunk flags are released, the exception is kept for result()."""
        self.error = exc
        self.init_done.set()
        if self.on_dev is not None:
            for _, flag, _ in self.on_dev._ure_chunks:
                flag.set()

    def _upload_stream(self):
        """The side stream of the calling worker thread."""
        key = (self.device, threading.get_ident())
        st = _UPLOAD_STREAMS.get(key)
        if st is None:
            st = _UPLOAD_STREAMS[key] = torch.cuda.Stream(self.device)
        return st

    def seeds_first(self):
        """The per-epoch seeds WITHOUT the model init: they follow the four fills in the stream, whose lengths are known, so a generator
        moved past them (ure_host_mt_advance) draws them at once -- and the permutations, the bulk of a request's host work, are expanded
        beside the inits instead of behind them.  False: the tables are too small to skip (the seeds come with init())."""
        if self.seeds is not None:
            return True
        start_state, n_user, n_item, k, epochs, with_total_test = self.args[:6]
        draws = model_draws(n_user, n_item, k, 0, False)
        if draws is None:
            return False
        # (one native call: the state is copied, moved past the four fills and asked for the epochs' int64 draws -- as `random_()` makes them)
        from . import _native as nv
        per = 4 if with_total_test else 3
        vals = np.empty(max(epochs, 0) * per, dtype=np.int64)
        nv.check(nv.lib().ure_host_draw_int64(start_state.data_ptr(), start_state.numel(), int(draws[0] + draws[1]), len(vals), vals.ctypes.data),
                 'ure_host_draw_int64')
        self.seeds = vals[1::per].tolist()
        return True

    def init(self):
        start_state, n_user, n_item, k, epochs, with_total_test = self.args[:6]
        from .engine import mark
        mark('w: init start')
        g = torch.Generator()
        g.set_state(start_state)
        init = mf_init(n_user, n_item, k, generator=g, threads=fill_threads((n_user + n_item) * k, self.sharers))
        seeds = epoch_seeds(epochs, with_total_test, generator=g)
        if self.seeds is None:
            self.seeds = seeds
        mark('w: init drawn')
        if self.device is not None:
            # the init tables go up first, on a side stream of this worker
            dev = self.device
            with torch.cuda.device(dev):
                st = self._upload_stream()
                with torch.cuda.stream(st):
                    up = tuple(t.to(dev) for t in init)
                    ev0 = torch.cuda.Event()
                    ev0.record(st)
            for t in up:
                t._ure_event = ev0
            init = up
        self.init_value = init
        self.init_done.set()

    def chunks(self):
        """Generator: every step expands (and uploads) one chunk of epochs; the permutations are complete when it ends."""
        from . import _native as nv
        _, _, _, _, epochs, _, n_rows, shuffle, threads, want_perms = self.args
        if not want_perms:
            return
        if not shuffle:
            self.perms_value = torch.arange(n_rows, dtype=torch.int32).repeat(epochs, 1)
            return
        if self.host is None:
            self.perms_value = epoch_perms(self.seeds, n_rows, threads)
            return
        sd = np.asarray(self.seeds, dtype=np.uint64).astype(np.int64)
        L = nv.lib()
        if self.on_dev is None:
            nv.check(L.ure_host_randperm(sd.ctypes.data, len(sd), n_rows, self.host.data_ptr(), threads), 'ure_host_randperm')
            self.perms_value = self.host
            return
        from .engine import mark
        dev, on_dev, host = self.device, self.on_dev, self.host
        with torch.cuda.device(dev):
            st = self._upload_stream()
        sd_ptr, host_ptr, row_bytes = sd.ctypes.data, host.data_ptr(), (2 if self.tags_batch else 4) * n_rows
        self.perms_value = on_dev
        first = True
        c0 = 0
        for c1, flag, slot in on_dev._ure_chunks:
            if _TEST_CHUNK_DELAY_S:
                import time
                time.sleep(_TEST_CHUNK_DELAY_S)
            if self.tags_batch:
                nv.check(L.ure_host_randperm_tags(sd_ptr + 8 * c0, c1 - c0, n_rows, self.tags_batch, host_ptr + row_bytes * c0, threads), 'ure_host_randperm_tags')
            else:
                nv.check(L.ure_host_randperm(sd_ptr + 8 * c0, c1 - c0, n_rows, host_ptr + row_bytes * c0, threads), 'ure_host_randperm')
            with torch.cuda.device(dev), torch.cuda.stream(st):
                if first:
                    st.wait_event(self.ready)
                    first = False
                on_dev[c0:c1].copy_(host[c0:c1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
            slot[0] = ev
            flag.set()
            mark(f'w: chunk to {c1}')
            c0 = c1
            yield


class ShardDraws:
    """Handle of rng.shard_draws_async / draws_batch_async: init() blocks until the model init is there (on the device when a
    device was given); perms() returns the permutations -- at once when they arrive in chunks (a device tensor whose
    `_ure_chunks` = [(first epoch after the chunk, threading.Event set once the chunk's upload is queued, [its HIP event])]
    tell a consumer when each part may be read: engine.TrainJob.run does), otherwise after the worker is done."""

    def __init__(self, future, task):
        self._future, self._task = future, task

    def init(self):
        self._task.init_done.wait()
        if self._task.error is not None:
            raise self._task.error
        return self._task.init_value

    def perms(self):
        if self._task.on_dev is not None:
            return self._task.on_dev
        self._future.result()
        if self._task.error is not None:
            raise self._task.error
        return self._task.perms_value

    def result(self):
        return self.init(), self.perms()


class _SharedBlock:
    """One pooled host block that the shards of a call share: it goes back to the pool when the last of them lets go."""

    def __init__(self, block, users):
        self.block, self.left, self.lock = block, users, threading.Lock()

    def drop(self):
        with self.lock:
            self.left -= 1
            last = self.left == 0
        if last:
            POOL.give(self.block)


def make_buffers_together(tasks):
    """make_buffers for the shards of one call from ONE host block, ONE device allocation and one event (a buffer, an allocation and an
    event per shard were 16 x 0.1-0.2 ms of the calling thread at 16 shards, beside 16 busy init workers).  Shards that cannot share
    (no device, another word type) get their own."""
    plans = [(t, t._buffer_plan()) for t in tasks]
    share = [(t, p) for t, p in plans if p is not None and t.device is not None]
    if len(share) < 2 or len({(p[0], str(t.device)) for t, p in share}) != 1:
        share = []
    if share:
        word, dev = share[0][1][0], share[0][0].device
        al = lambda x: (x + 63) // 64 * 64
        offs, at = [], 0
        for _, (_, epochs, n_rows) in share:
            offs.append(at)
            at += al(epochs * n_rows)
        host_all = POOL.take((at,), word)
        dev_all = torch.empty(at, dtype=word, device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        block = _SharedBlock(host_all, len(share))
        for (t, (_, epochs, n_rows)), o in zip(share, offs):
            t._adopt(host_all[o:o + epochs * n_rows].view(epochs, n_rows), dev_all[o:o + epochs * n_rows].view(epochs, n_rows), ready, block)
    shared = {id(t) for t, _ in share}
    for t, p in plans:
        if id(t) not in shared:
            t.make_buffers()


PERM_DTYPE = np.dtype([('seed', '<i8'), ('tags', '<u8'), ('n', '<i4'), ('batch', '<i4')])        # struct ure_perm
DEVICE_TAGS_MAX_ROWS = 1 << 27      # csrc/perm_chain.hip takes 2^28 (16,384 ranges of 16,384 targets); a row index keeps a bit free here (csrc/perm_tags.hip alone stopped at 2^20)
RESERVATIONS_MAX_ROWS = 1 << 20     # csrc/perm_tags.hip: a swap's index shares a 32-bit reservation word with the round counter
DEVICE_TAGS_GROUPS = 256            # workgroups (= permutations at a time) per launch: 32 / 64 / 128 / 256 -> 13.4 / 11.2 / 9.8 / 9.7 ms per 5-shard request, 18.4 / 14.8 / 12.8 / 12.3 at 16 shards
# Which of the two device shuffles makes a chunk (profiles/r05/exp_shuffle.json): csrc/perm_chain.hip -- many workgroups per permutation, six
# launches -- has the latency (5 x 180 k rows 0.23 ms against 0.59, one 897 k-row epoch 0.74 against 3.42), the big shards (50 x 897 k rows 1.89 ms
# against 4.37; beyond 2^20 rows it is the only one) and the throughput (250 x 180 k rows 1.22 ms against 1.61; 800 x 56 k rows 1.16 against 1.27):
# it makes every chunk.  csrc/perm_tags.hip -- one workgroup per permutation, rounds of reservations: round 4's -- stays behind
# URE_SHUFFLE=reservations (shards of up to 2^20 rows) as the second implementation the first is tested against.
_PERM_STREAMS = {}


def device_tags_wanted():
    """URE_DEVICE_TAGS=0 keeps the host's expansion threads."""
    import os
    return os.environ.get('URE_DEVICE_TAGS', '1') != '0'


def shuffle_method(n_max, n_perms, method=None):
    """-> 'chain' (csrc/perm_chain.hip) or 'reservations' (csrc/perm_tags.hip) for a launch of n_perms permutations of up to n_max rows."""
    import os
    method = method or os.environ.get('URE_SHUFFLE', 'auto')
    return 'reservations' if method == 'reservations' and n_max <= RESERVATIONS_MAX_ROWS else 'chain'


def default_tag_bounds(epochs, n_shards, n_max):
    """The chunks of epochs a call's shuffles are launched in (the same for every shard; TrainJob.run waits for a chunk right before the
    launches that read it).  Small shards: the first epochs in one small chunk -- the chain shuffle makes it in ~0.4 ms, training can
    start --, the rest about DEVICE_TAGS_GROUPS permutations per launch (two small chunks first cost a request 0.35 ms more device time for 0.1 ms
    less waiting).  Big shards (chain throughout): about eight
    permutations per launch, a launch per epoch beyond 4 M rows -- a chunk of those keeps up with the epochs that consume it."""
    S = max(1, n_shards)
    if n_max > (1 << 18):
        per = 1 if n_max > (4 << 20) else max(1, (8 if n_max >= (1 << 19) else 32) // S)
        return list(range(0, epochs, per)) + [epochs]
    bounds, at = [0], 0
    if epochs > 1:                                       # (one small chunk first: ~64 permutations, 0.4-0.5 ms by the chain shuffle)
        at = min(epochs, max(1, 64 // S))
        bounds.append(at)
    per = max(2, DEVICE_TAGS_GROUPS // S)
    while at < epochs:
        at = min(epochs, at + per)
        bounds.append(at)
    return bounds


def device_tags(tasks, bounds=None, defer=False, method=None):
    """The batch tags of a call's shards made on the DEVICE instead of by host threads: the seeds come by skip-ahead
    (_DrawsTask.seeds_first), every (shard, epoch) is one entry of ONE descriptor table, and the epochs go out in a few launches on a
    side stream -- chunk c of all shards together, by ure_device_shuffle_tags (csrc/perm_chain.hip) or ure_device_randperm_tags
    (csrc/perm_tags.hip), whichever suits the chunk (shuffle_method) --, each with the event TrainJob.run waits for before the launches
    that read it.  No host buffer, no upload, no expansion threads.  -> False when a shard cannot take this path (no tags, more than 2^27
    rows, tables too small to skip ahead): the caller falls back to the host path for the whole call.
    bounds: the chunks' epoch boundaries [0, ..., epochs] instead of default_tag_bounds.  defer: nothing is launched here; -> fire(c),
    which launches chunk c (in order) -- bench.py puts the shuffles of the epochs it times inside its clock."""
    from . import _native as nv
    from .engine import upload_many
    plans = [t._buffer_plan() for t in tasks]
    if not tasks or any(p is None or p[0] != torch.int16 or p[2] > DEVICE_TAGS_MAX_ROWS or t.device is None for t, p in zip(tasks, plans)):
        return False
    if len({str(t.device) for t in tasks}) != 1 or not all(t.seeds_first() for t in tasks):
        return False
    dev = tasks[0].device
    epochs = plans[0][1]
    if any(p[1] != epochs for p in plans):
        return False
    al = lambda x: (x + 63) // 64 * 64
    offs, at = [], 0
    for _, e, n_rows in plans:
        offs.append(at)
        at += al(e * n_rows)
    from .engine import mark
    mark('tags: start')
    main = torch.cuda.current_stream(dev)
    dev_all = torch.empty(at, dtype=torch.int16, device=dev)
    S = len(tasks)
    n_max = max(p[2] for p in plans)
    if bounds is None:
        bounds = default_tag_bounds(epochs, S, n_max)
    bounds = [int(b) for b in bounds]
    assert bounds[0] == 0 and bounds[-1] == epochs and all(a < b for a, b in zip(bounds[:-1], bounds[1:]))
    # the table in launch order -- (chunk, shard, epoch) --, built shard by epoch and reordered once
    n_of = np.array([p[2] for p in plans], dtype=np.int64)
    full = np.zeros((S, epochs), dtype=PERM_DTYPE)
    full['seed'] = np.array([t.seeds for t in tasks], dtype=np.uint64).astype(np.int64)
    full['tags'] = dev_all.data_ptr() + 2 * (np.array(offs, dtype=np.int64)[:, None] + np.arange(epochs, dtype=np.int64)[None, :] * n_of[:, None])
    full['n'] = n_of[:, None]
    full['batch'] = np.array([t.tags_batch for t in tasks], dtype=np.int64)[:, None]
    table = np.concatenate([full[:, c0:c1].reshape(-1) for c0, c1 in zip(bounds[:-1], bounds[1:])])
    launches, at_row = [], 0
    for c, (c0, c1) in enumerate(zip(bounds[:-1], bounds[1:])):
        n_p = S * (c1 - c0)
        launches.append((c1, at_row, at_row + n_p, shuffle_method(n_max, n_p, method), 0))          # (one side stream: chunks alternating between two ran beside each other and took
                                                                                        # longer each -- 12.8 against 9.2 ms of shuffle kernels per six requests, the 16-shard request 11.7 against 9.3 ms)
        at_row += n_p
    mark('tags: table')
    L = nv.lib()
    sides = _PERM_STREAMS.get(str(dev))
    if sides is None:
        sides = _PERM_STREAMS[str(dev)] = (torch.cuda.Stream(dev),)      # (stream priorities change nothing here: measured, profiles/r05/NOTES.md)
    # scratch per (side stream, method): the launches of a stream follow each other, so they share it
    scratch, flags, keep = {}, [], [dev_all]
    for _, lo, hi, how, side in launches:
        if how == 'chain':
            want = (int(L.ure_device_shuffle_tags_scratch(n_max, hi - lo)), hi - lo)
        else:
            groups = min(DEVICE_TAGS_GROUPS, hi - lo)
            want = (int(L.ure_device_randperm_tags_scratch(n_max, groups)), groups)
        have = scratch.get((side, how))
        if have is None or want[1] > have[1]:
            scratch[(side, how)] = want
    n_al = al(n_max)
    for (side, how), (words, n_p) in list(scratch.items()):
        block = torch.empty(words, dtype=torch.int32, device=dev)
        if how == 'chain':
            flag_at = int(L.ure_device_shuffle_tags_flag(n_max, n_p))
            f = block[flag_at:flag_at + 1]
        else:
            f = block[2 * n_al * n_p:2 * n_al * n_p + n_p]           # a word per workgroup: 0xdead if it ever gave up (device_tags_check)
        f.zero_()
        flags.append(f)
        keep.append(block)
        scratch[(side, how)] = (block, words, n_p)
    mark('tags: scratch')
    table_dev = upload_many([table.view(np.uint8)], dev)[0]
    keep.append(table_dev)
    # (the blocks are made on the caller's stream and worked on by the side streams: the allocator must not hand them to anybody
    # else before the side streams are through with them, whatever becomes of the request -- ADVICE r4)
    for block in keep:
        for side in sides:
            block.record_stream(side)
    ready = torch.cuda.Event()
    ready.record(main)
    for side in sides:
        side.wait_event(ready)
    chunks = [(c1, threading.Event(), [None]) for c1, *_ in launches]    # (the same for every shard: a launch holds chunk c of all of them)

    def fire(c):
        c1, lo, hi, how, side = launches[c]
        block, words, n_p = scratch[(side, how)]
        ptr = table_dev.data_ptr() + lo * PERM_DTYPE.itemsize
        if how == 'chain':
            nv.check(L.ure_device_shuffle_tags(ptr, hi - lo, n_max, block.data_ptr(), words, 0, sides[side].cuda_stream), 'ure_device_shuffle_tags')
        else:
            nv.check(L.ure_device_randperm_tags(ptr, hi - lo, n_max, block.data_ptr(), words, n_p, sides[side].cuda_stream), 'ure_device_randperm_tags')
        ev = torch.cuda.Event()
        ev.record(sides[side])
        chunks[c][2][0] = ev
        chunks[c][1].set()
    mark('tags: uploaded')
    if not defer:
        for c in range(len(launches)):
            fire(c)
    mark('tags: launched')
    for t, (_, e, n_rows), o in zip(tasks, plans, offs):
        on_dev = dev_all[o:o + e * n_rows].view(e, n_rows)
        t.host, t.on_dev, t.ready = None, on_dev, ready
        on_dev._ure_host, on_dev._ure_shared = None, None
        on_dev._ure_keep = tuple(keep)                            # (alive as long as the tags are: the side streams work on them)
        on_dev._ure_flags = flags
        on_dev._ure_chunks = list(chunks)
        on_dev._ure_methods = [how for *_, how, _ in launches]
        t.perms_value = on_dev
    return fire if defer else True


def device_tags_check(perms):
    """After the request's device work is done (the caller has synchronised): did a workgroup of perm_tags_kernel give up, did the resolve
    pass of perm_chain.hip meet a link it cannot follow?  Neither can happen -- and their tags match no batch, so nothing trained on them --,
    but tags that were not made must not go unnoticed.  perms: the tag tensors of the call's shards (shared flag words; read once)."""
    seen = set()
    for p in perms:
        for f in getattr(p, '_ure_flags', None) or ():
            if f.data_ptr() not in seen:
                seen.add(f.data_ptr())
                if bool(f.ne(0).any()):
                    from ._native import NativeError
                    raise NativeError('device shuffle: a permutation was given up (URE_DEVICE_TAGS=0 takes the host path)')


def _task_of(sp, buffers=True):
    t = _DrawsTask(sp['start_state'], sp['n_user'], sp['n_item'], sp['k'], sp['epochs'], sp['with_total_test'], sp.get('n_rows', 0),
                      sp.get('shuffle', False), sp.get('threads', 0), sp.get('device'), sp.get('want_perms', True), sp.get('chunk_epochs', 8),
                      sp.get('tags_batch', 0), buffers)
    if sp.get('seeds') is not None:
        t.seeds = list(sp['seeds'])
    return t


def _guarded_init(t):
    try:
        t.init()
    except BaseException as e:
        t.fail(e)
        raise


def start_inits(specs):
    """The model inits of a call's shards, each on a worker of its own, started at once: all they need is the shard's start state.
    -> the tasks, to be handed to draws_batch_async(tasks=...) once the caller has done what is more urgent than the permutations'
    buffers (a request: getting its layouts under way)."""
    tasks = [_task_of(sp, buffers=False) for sp in specs]
    _submit_inits(tasks)
    return tasks


def _submit_inits(tasks):
    """The model inits of a call's shards: on the device where the tables are big enough to be worth it (_device_init), else ONE native
    call for all shards' fills and one upload (_batch_init), else a worker per shard."""
    pool = worker_pool()
    if tasks and _device_init_possible(tasks):
        pool.submit(_device_init, tasks)
    elif len(tasks) > 1 and _batch_init_possible(tasks):
        pool.submit(_batch_init, tasks)
    else:
        for t in tasks:
            t.sharers = len(tasks)
            pool.submit(_guarded_init, t)


def _device_init_wanted(n_normals, device):
    return device is not None and torch.device(device).type == 'cuda' and n_normals >= DEVICE_INIT_MIN_NORMALS and device_fill_ok(device)


def _device_init_possible(tasks):
    a = tasks[0].args
    if tasks[0].device is None or not all(t.args[1:4] == a[1:4] and t.args[4:6] == a[4:6] and t.device == tasks[0].device for t in tasks):
        return False
    return model_draws(a[1], a[2], a[3], 0, False) is not None and _device_init_wanted(len(tasks) * (a[1] + a[2]) * a[3], tasks[0].device)


def _device_init(tasks):
    """The model inits of all shards of a call made on the device (mf_init_device): the host positions the generators (microseconds per
    shard: csrc/mt_jump.cpp), the kernels run on this worker's side stream, every task gets its views and the event consumers wait for.
    No host normals, no upload -- 909 M values and 3.6 GB at BASELINE.json configs[3]."""
    from .engine import mark
    try:
        mark('w: init start')
        _, n_user, n_item, k, epochs, with_total_test = tasks[0].args[:6]
        draws = model_draws(n_user, n_item, k, 0, False)
        dev = tasks[0].device
        states = [t.args[0].clone() for t in tasks]
        with torch.cuda.device(dev):
            up = tasks[0]._upload_stream()
            out = mf_init_device(states, n_user * k, n_item * k, draws[0], dev, stream=up)
            ev = torch.cuda.Event()
            ev.record(up)
        for t, st, (U0, V0) in zip(tasks, states, out):
            if t.seeds is None:                          # (the epochs' seeds follow the fills in the stream)
                g = torch.Generator()
                g.set_state(st)
                t.seeds = epoch_seeds(epochs, with_total_test, generator=g)
            U0, V0 = U0.view(n_user, k), V0.view(n_item, k)
            U0._ure_event = V0._ure_event = ev
            t.init_value = (U0, V0)
        mark('w: init queued on the device')
        for t in tasks:
            t.init_done.set()
    except BaseException as e:
        for t in tasks:
            t.fail(e)
        raise


def _batch_init_possible(tasks):
    a = tasks[0].args
    return (native_fill_ok() and all(t.args[1:4] == a[1:4] and t.args[4:6] == a[4:6] and t.device == tasks[0].device for t in tasks)
            and model_draws(a[1], a[2], a[3], 0, False) is not None
            and len(tasks) * (a[1] + a[2]) * a[3] * 4 <= (256 << 20))          # (one pinned block for all of them: small tables only)


def _batch_init(tasks):
    """The model inits of all shards of a call (same table sizes) by ONE native call (ure_host_mf_init_batch: the shards side by side on
    host threads) into one pinned block, uploaded in one copy; every task gets its views.  (A worker per shard -- 16 Python threads at
    configs[4] -- kept each other and the calling thread waiting for the interpreter lock: the layouts' worker started 2 ms late.)"""
    from . import _native as nv
    from .engine import mark
    try:
        mark('w: init start')
        _, n_user, n_item, k, epochs, with_total_test = tasks[0].args[:6]
        S, nu, nv_ = len(tasks), n_user * k, n_item * k
        draws = model_draws(n_user, n_item, k, 0, False)
        block = POOL.take((S, nu + nv_), torch.float32)
        states = [t.args[0].clone() for t in tasks]
        st_a = (ctypes.c_void_p * S)(*[x.data_ptr() for x in states])
        base, row = block.data_ptr(), 4 * (nu + nv_)
        u_a = (ctypes.c_void_p * S)(*[base + s * row for s in range(S)])
        v_a = (ctypes.c_void_p * S)(*[base + s * row + 4 * nu for s in range(S)])
        skip = (ctypes.c_int64 * S)(*([int(draws[0])] * S))
        nv.check(nv.lib().ure_host_mf_init_batch(S, st_a, states[0].numel(), skip, u_a, nu, v_a, nv_, host_cpus()), 'ure_host_mf_init_batch')
        _count(skipped_draws=S * draws[0], normals=S * (n_user + n_item) * k)
        per = 4 if with_total_test else 3
        for t, st in zip(tasks, states):
            if t.seeds is None:                          # (the epochs' seeds follow the fills in the stream)
                g = torch.Generator()
                g.set_state(st)
                t.seeds = epoch_seeds(epochs, with_total_test, generator=g)
        mark('w: init drawn')
        dev = tasks[0].device
        if dev is not None:
            with torch.cuda.device(dev):
                up = tasks[0]._upload_stream()
                with torch.cuda.stream(up):
                    on_dev = block.to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(up)
            POOL.give(block, ev)
            for s, t in enumerate(tasks):
                U0, V0 = on_dev[s, :nu].view(n_user, k), on_dev[s, nu:].view(n_item, k)
                U0._ure_event = V0._ure_event = ev
                t.init_value = (U0, V0)
        else:
            host = block.clone()
            POOL.give(block)
            for s, t in enumerate(tasks):
                t.init_value = (host[s, :nu].view(n_user, k), host[s, nu:].view(n_item, k))
        for t in tasks:
            t.init_done.set()
    except BaseException as e:
        for t in tasks:
            t.fail(e)
        raise


def draws_batch_async(specs, n_workers=0, tasks=None):
    """shard_draws_async for the shards of one call: every model init on a worker of its own, and beside them FEW chunk workers --
    worker w takes the shards w, w + W, ...: their seeds first (_DrawsTask.seeds_first: no init needed), then
    their permutation chunks round robin, so that the first chunk of every shard arrives before anybody's second.  One thread per shard -- round 2 -- meant 16 Python threads taking turns on the GIL
    with the calling thread for a 16-shard call (10 ms between two of its marks).  specs: list of dicts of shard_draws_async's
    arguments.  -> [ShardDraws]."""
    pool = worker_pool()
    started = tasks is not None                      # (start_inits: the inits are running already)
    if started:
        for t, sp in zip(tasks, specs):
            t.args = t.args[:8] + (int(sp.get('threads', 0) or 0),) + t.args[9:]
    else:
        tasks = [_task_of(sp, buffers=False) for sp in specs]
    on_device = device_tags_wanted() and device_tags(tasks)
    if not on_device:
        make_buffers_together(tasks)
    W = max(1, min(len(tasks), int(n_workers) if n_workers else max(2, host_cpus() // 2)))

    def work(mine):
        todo = list(mine)
        try:
            for t in mine:
                if not t.seeds_first():
                    t.init_done.wait()                  # (tables too small to skip: the seeds come with the init)
                    if t.error is not None:
                        raise t.error
            gens = [(t, t.chunks()) for t in mine]
            while gens:
                for t, g in list(gens):
                    try:
                        next(g)
                    except StopIteration:
                        gens.remove((t, g))
                        todo.remove(t)
        except BaseException as e:
            for t in todo:
                t.fail(e)
            raise

    if not started:
        _submit_inits(tasks)                            # the model inits ...
    if on_device:                                       # (the tags are being made on the device: nothing left for chunk workers)
        from concurrent.futures import Future
        done = Future()
        done.set_result(None)
        return [ShardDraws(done, t) for t in tasks]
    futures = [pool.submit(work, tasks[w::W]) for w in range(W)]      # ... and beside them the permutation chunks
    return [ShardDraws(futures[i % W], t) for i, t in enumerate(tasks)]


def shard_draws_async(start_state, n_user, n_item, k, epochs, with_total_test, n_rows, shuffle, threads=0, device=None, want_perms=True,
                      chunk_epochs=8):
    """Everything random of ONE Scratch.train call, taken on a worker thread from its own generator positioned at
    `start_state` (shard_streams): the model init (utils.py:31-40), the per-epoch seeds and the expanded
    permutations.  The shards of a SISA call are independent streams once their start states are known, so their
    draws run side by side instead of one after the other (draws_batch_async).  With a HIP `device` the init tables and
    the permutations are uploaded on a side stream, the permutations in chunks of epochs so that training starts on the first
    epochs while the later ones are still being expanded.  -> ShardDraws."""
    return draws_batch_async([dict(start_state=start_state, n_user=n_user, n_item=n_item, k=k, epochs=epochs, with_total_test=with_total_test,
                                   n_rows=n_rows, shuffle=shuffle, threads=threads, device=device, want_perms=want_perms,
                                   chunk_epochs=chunk_epochs)], 1)[0]


def seed_all(seed):
    """method/utils.py:21-25 as written (numpy + the device generator; NOT the torch
    CPU generator -- SURVEY D7: the harness seeds that one)."""
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
