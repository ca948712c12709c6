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
import numpy as np
import torch


def mf_init(n_user, n_item, k):
    """The four N(0,1) fills of `MF(n_user, n_item, k)`; the first two (Embedding
    constructors) are overwritten by init_weight but still advance the stream."""
    torch.empty(n_user, k).normal_(0, 1)
    torch.empty(n_item, k).normal_(0, 1)
    U0 = torch.empty(n_user, k).normal_(0, 1)
    V0 = torch.empty(n_item, k).normal_(0, 1)
    return U0, V0


def draw_seed():
    return int(torch.empty((), dtype=torch.int64).random_().item())


def epoch_seeds(epochs, with_total_test):
    """Per epoch the reference draws: train loader base seed, sampler seed, group-test
    loader base seed and (SISA only) total-test loader base seed (scratch.py:78-97).
    Returns the sampler seeds; the others only advance the stream."""
    per = 4 if with_total_test else 3
    if epochs <= 0:
        return []
    # one vectorised draw: random_() fills serially from the same generator, so the values are those of
    # `per * epochs` scalar draws (checked in tests/test_cpu_host.py)
    draws = torch.empty(epochs * per, dtype=torch.int64).random_()
    return [int(v) for v in draws[1::per].tolist()]


def epoch_perm(seed, n):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).to(torch.int32)


class _HostPool:
    """Reusable (pinned when a HIP device is present) host buffers for the epoch permutations.
    A 50-epoch, 5-shard ml-1m job needs 180 MB of them; allocating, first-touching and
    unmapping that much pageable memory per call costs more host time than the whole
    training costs device time, so buffers are kept and handed out again."""

    def __init__(self):
        self.free = []
        self.lent = {}

    def take(self, shape, dtype):
        need = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        best = None
        for i, b in enumerate(self.free):
            if b.numel() >= need and (best is None or b.numel() < self.free[best].numel()):
                best = i
        if best is not None:
            buf = self.free.pop(best)
        else:
            size = max(need, 1)
            size = (size + (1 << 20) - 1) >> 20 << 20
            buf = torch.empty(size, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        view = buf[:need].view(dtype).view(shape)
        self.lent[view.data_ptr()] = buf
        return view

    def give(self, view):
        buf = self.lent.pop(view.data_ptr(), None)
        if buf is not None:
            self.free.append(buf)


POOL = _HostPool()


def release(perms):
    """Hand a permutation buffer from epoch_perms(pooled=True) back (after it was uploaded)."""
    if torch.is_tensor(perms):
        host = getattr(perms, '_ure_host', None)
        if host is not None:                    # uploaded by the background worker: wait for that copy
            perms._ure_event.synchronize()
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


def epoch_perms_async(seeds, n, threads=0, pooled=False, device=None):
    """epoch_perms on a background thread: returns a future whose result() is the tensor.  The seeds
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
    out = POOL.take((len(seeds), n), torch.int32) if pooled else torch.empty(len(seeds), n, dtype=torch.int32)
    sd = np.asarray(seeds, dtype=np.uint64).astype(np.int64)

    on_dev = ready = None
    if device is not None and torch.device(device).type == 'cuda':
        # destination allocated here, on the caller's stream (the allocator's pools are per stream);
        # the side stream starts its copy only after whatever the caller's stream was doing with that memory
        dev = torch.device(device)
        on_dev = torch.empty((len(seeds), n), dtype=torch.int32, device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))

    def work():
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


def seed_all(seed):
    """method/utils.py:21-25 as written (numpy + the device generator; NOT the torch
    CPU generator -- SURVEY D7: the harness seeds that one)."""
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
