"""The stateless C-ABI entry points as PyTorch custom ops (`torch.ops.ultrare.*`).

BASELINE.json's north_star words the boundary as "HIP kernels exposed to Python as PyTorch-ROCm custom ops"; the
boundary of record is the C ABI (include/ultrare_hip.h, bound with ctypes in _native.py) and these ops are thin
calls into the same library: they take and return torch tensors, run on the current HIP stream, and have no CPU
implementation (a CPU tensor is an error, not a fallback).  The training job (ure_job_*) is stateful -- a handle
to resident shards -- and stays behind engine.TrainJob.

    torch.ops.ultrare.mf_score(U, V, uid, iid)          utils.py:42-43    row-wise dot of gathered rows
    torch.ops.ultrare.ot_cost(X, C)                     utils.py:637      [k, n] squared distances, numpy's fp32 order
    torch.ops.ultrare.ot_cost_mfma(X, C)                (same, |x|^2 - 2 x.c + |c|^2 on the matrix cores; cross-check only)
    torch.ops.ultrare.ot_centroids(X, label, k)         utils.py:648      cluster means, numpy's fp32 order
    torch.ops.ultrare.merge_rows(dst, src, rows)        sisa.py:55-56     dst[rows] = src[rows]  (in place)
"""
import ctypes

import torch

from . import _native as nv
from . import engine


def _dev(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise nv.NativeError('ultrare ops run on the HIP device only (no CPU fallback)')


@torch.library.custom_op('ultrare::mf_score', mutates_args=())
def mf_score(U: torch.Tensor, V: torch.Tensor, uid: torch.Tensor, iid: torch.Tensor) -> torch.Tensor:
    _dev(U, V, uid, iid)
    d = engine.pad_dim(U.shape[1])
    assert U.shape[1] == d and V.shape[1] == d and U.is_contiguous() and V.is_contiguous(), 'tables must have the padded width'
    uid, iid = uid.to(torch.int32).contiguous(), iid.to(torch.int32).contiguous()
    pred = torch.empty(uid.numel(), dtype=torch.float32, device=U.device)
    Up, Vp = (ctypes.c_void_p * 1)(U.data_ptr()), (ctypes.c_void_p * 1)(V.data_ptr())
    nv.check(nv.lib().ure_score(Up, Vp, 1, 1, 1, 1, nv.ptr(uid), nv.ptr(iid), None, uid.numel(), d, nv.ptr(pred), None, nv.stream_handle()),
             'ure_score')
    return pred


@mf_score.register_fake
def _(U, V, uid, iid):
    return U.new_empty(uid.numel())


def _cost(fn, X, C):
    _dev(X, C)
    X, C = X.float().contiguous(), C.float().contiguous()
    dist = torch.empty(C.shape[0], X.shape[0], dtype=torch.float32, device=X.device)
    nv.check(fn(nv.ptr(X), nv.ptr(C), X.shape[0], C.shape[0], X.shape[1], nv.ptr(dist), nv.stream_handle()), 'ure_ot_cost')
    return dist


@torch.library.custom_op('ultrare::ot_cost', mutates_args=())
def ot_cost(X: torch.Tensor, C: torch.Tensor) -> torch.Tensor:
    return _cost(nv.lib().ure_ot_cost, X, C)


@torch.library.custom_op('ultrare::ot_cost_mfma', mutates_args=())
def ot_cost_mfma(X: torch.Tensor, C: torch.Tensor) -> torch.Tensor:
    return _cost(nv.lib().ure_ot_cost_mfma, X, C)


@ot_cost.register_fake
def _(X, C):
    return X.new_empty(C.shape[0], X.shape[0])


@ot_cost_mfma.register_fake
def _(X, C):
    return X.new_empty(C.shape[0], X.shape[0])


@torch.library.custom_op('ultrare::ot_centroids', mutates_args=())
def ot_centroids(X: torch.Tensor, label: torch.Tensor, k: int) -> torch.Tensor:
    _dev(X, label)
    X, label = X.float().contiguous(), label.to(torch.int32).contiguous()
    C = torch.empty(k, X.shape[1], dtype=torch.float32, device=X.device)
    counts = torch.empty(k, dtype=torch.int32, device=X.device)
    nv.check(nv.lib().ure_ot_centroids(nv.ptr(X), nv.ptr(label), X.shape[0], k, X.shape[1], nv.ptr(C), nv.ptr(counts), nv.stream_handle()),
             'ure_ot_centroids')
    return C


@ot_centroids.register_fake
def _(X, label, k):
    return X.new_empty(k, X.shape[1])


@torch.library.custom_op('ultrare::merge_rows', mutates_args=('dst',))
def merge_rows(dst: torch.Tensor, src: torch.Tensor, rows: torch.Tensor) -> None:
    _dev(dst, src, rows)
    assert dst.is_contiguous() and src.is_contiguous() and dst.shape == src.shape
    rows = rows.to(torch.int64).contiguous()
    nv.check(nv.lib().ure_merge_rows(nv.ptr(dst), nv.ptr(src), nv.ptr(rows), rows.numel(), dst.shape[1], nv.stream_handle()), 'ure_merge_rows')
