#!/usr/bin/env python3
"""End-to-end wall time of the SISA path on synthetic ml-1m: Sisa.learn then Sisa.unlearn
after a 2 % random user deletion (BASELINE.json metric, second half), through the
reference's operator surface (ultrare_amd.measure.sisa_request).  Prints one JSON object.

    python tools/e2e_sisa.py [--shards 5] [--k 32] [--epochs 50] [--parallel 1]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd.measure import sisa_request as measure  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shards', type=int, default=5)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=50)
    ap.add_argument('--parallel', type=int, default=1)
    ap.add_argument('--delper', type=float, default=2.0)
    ap.add_argument('--workload', choices=['ml1m', 'ml25m'], default='ml1m')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(measure(a.shards, a.k, a.epochs, a.parallel, a.delper, reps=a.reps, workload=a.workload)))


if __name__ == '__main__':
    main()
