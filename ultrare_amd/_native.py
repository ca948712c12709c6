"""ctypes binding of libultrare_hip.so (include/ultrare_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module
raises.  Device memory, streams and process groups come from torch (plumbing);
all arithmetic of the hot path happens inside the library's HIP kernels.
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libultrare_hip.so')
ABI_VERSION = 1
MAX_MODELS_PER_CALL = 32

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64


class UreShard(ctypes.Structure):
    """struct ure_shard (include/ultrare_hip.h)."""
    _fields_ = [
        ('ent_oid', _vp), ('ent_r', _vp), ('ent_tag', _vp), ('ent_src', _vp), ('file_tag', _vp), ('inv_stage', _vp), ('inv_off', _vp),
        ('sched', _vp), ('n_block', _i32), ('n_wave', _i32), ('n_active', _i32), ('n_slots', _i64),
        ('U', _vp * 2), ('V', _vp * 2), ('mU', _vp), ('mV', _vp),
        ('U0', _vp), ('V0', _vp), ('lr_host', _vp), ('lazy_rows', _i32),
        ('perm', _vp), ('lr', _vp), ('sse', _vp),
        ('N', _i32), ('n_user', _i32), ('n_item', _i32), ('d', _i32),
        ('batch', _i32), ('epochs', _i32),
        ('lam', ctypes.c_float), ('mu', ctypes.c_float),
    ]


class NativeError(RuntimeError):
    pass


_PROTOTYPES = {
    'ure_abi_version': (ctypes.c_int, []),
    'ure_last_error': (ctypes.c_char_p, []),
    'ure_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_char_p, ctypes.c_int]),
    'ure_job_create': (ctypes.c_int, [ctypes.POINTER(UreShard), ctypes.c_int, ctypes.POINTER(_vp)]),
    'ure_job_destroy': (ctypes.c_int, [_vp]),
    'ure_job_shard_steps': (_i64, [_vp, ctypes.c_int]),
    'ure_job_ticks': (_i64, [_vp]),
    'ure_job_train': (ctypes.c_int, [_vp, _i64, _i64, _vp]),
    'ure_job_materialize': (ctypes.c_int, [_vp, _i64, _vp]),
    'ure_job_train_profiled': (ctypes.c_int, [_vp, _i64, _i64, _vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64)]),
    'ure_host_randperm': (ctypes.c_int, [_vp, ctypes.c_int, _i64, _vp, ctypes.c_int]),
    'ure_score': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, _vp, _vp, _vp, _i64, ctypes.c_int, _vp, _vp, _vp]),
    'ure_eval_users': (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ure_merge_rows': (ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_int, _vp]),
    'ure_ot_cost': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'ure_ot_centroids': (ctypes.c_int, [_vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    'ure_ot_assign': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp, ctypes.POINTER(ctypes.c_double)]),
}
EXPORTS = tuple(_PROTOTYPES)

_lib = None


def lib():
    """Load the library once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f'{LIB_PATH} is missing: run `python -m ultrare_amd.build` '
                              '(there is no CPU fallback for the SISA hot path)')
        # torch first: it bundles its own libamdhip64 / libhsa-runtime64, and the process must
        # end up with ONE HIP runtime -- the one that owns the device pointers torch hands us.
        # Loading this library first would pull /opt/rocm's copy and split the process in two.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.ure_abi_version() != ABI_VERSION:
            raise NativeError(f'ABI mismatch: library {L.ure_abi_version()} != binding {ABI_VERSION}')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ure_last_error().decode(errors='replace')
        raise NativeError(f'{what} failed (code {rc}): {msg}')


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags['C_CONTIGUOUS']
        return t.ctypes.data
    assert t.is_contiguous()
    return t.data_ptr()


def stream_handle(stream=None):
    """hipStream_t of a torch stream (default: the current stream of the current device)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def ot_assign(dist_kn):
    """Exact balanced OT on the host (ure_ot_assign): dist [k][n] fp32 ->
    (label int32 [n], plan int32 [n][k] in units of 1/(n k), objective)."""
    dist_kn = np.ascontiguousarray(dist_kn, dtype=np.float32)
    k, n = dist_kn.shape
    label = np.empty(n, dtype=np.int32)
    plan = np.empty((n, k), dtype=np.int32)
    obj = ctypes.c_double()
    check(lib().ure_ot_assign(dist_kn.ctypes.data, n, k, label.ctypes.data, plan.ctypes.data, ctypes.byref(obj)),
          'ure_ot_assign')
    return label, plan, obj.value
