#!/usr/bin/env python3
"""Experiment (profiles/r05/NOTES.md 6): the chip cut in two by CU masks -- the shuffles' side stream on a fraction of the compute units, the
training launches on the rest -- for the bench's inclusive region.  Streams from hipExtStreamCreateWithCUMask wrapped as torch ExternalStreams;
bench.train_leg is run with rng's side stream and the current stream replaced.   python tools/exp_cumask.py [--frac 0.25] [--mode split|side|none]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def masked_stream(hip, bits):
    words = (len(bits) + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b:
            arr[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frac', type=float, default=0.25)
    ap.add_argument('--mode', default='split')
    ap.add_argument('--layout', default='interleaved', help='interleaved: every 1/frac-th bit; low: the lowest bits')
    a = ap.parse_args()
    import bench
    from ultrare_amd import rng
    torch.cuda.init()
    hip = ctypes.CDLL('libamdhip64.so')
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    k = max(1, int(round(1 / a.frac)))
    if a.layout == 'interleaved':
        side_bits = [1 if (i // 8) % k == 0 else 0 for i in range(n_cu)]       # (groups of 8 bits: one per XCD if the bits go round the XCDs)
    else:
        side_bits = [1 if i < n_cu * a.frac else 0 for i in range(n_cu)]
    main_bits = [1 - b for b in side_bits]
    args = argparse.Namespace(gpus=1, backend='nccl', force_dist=False, force_device=None)
    D = bench.Dist.__new__(bench.Dist)
    D.world, D.rank, D.local, D.pg = 1, 0, 0, None
    D.barrier = lambda: None
    D.max = lambda x: float(x)
    D.sum = lambda x: int(x)
    out = {'n_cu': n_cu, 'frac': a.frac, 'mode': a.mode, 'layout': a.layout, 'side_cus': sum(side_bits)}
    dev = str(torch.device('cuda', 0))
    if a.mode in ('split', 'side'):
        rng._PERM_STREAMS[dev] = (masked_stream(hip, side_bits),)
    bench.NO_RESIDENT = True
    if a.mode == 'split':
        main = masked_stream(hip, main_bits)
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            leg = bench.train_leg(D, 'ml1m', 5, 32, 30000, 50, 3, False, 0)
    else:
        leg = bench.train_leg(D, 'ml1m', 5, 32, 30000, 50, 3, False, 0)
    out.update(value=round(leg['value'] / 1e9, 3), device_ms=round(leg['dev_ms'], 3), waited_ms=round(leg['waited_ms'], 3),
               avg_launch_us=round((leg['dev_ms'] - leg['waited_ms']) / leg['n_launch'] * 1e3, 2))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
