#!/usr/bin/env python3
"""Cold end-to-end unlearning request at ml-1m size, from CSV files on disk to the final
ensemble test (the reference's Instance.__group path, config.py:123-174, uniform groups):
read + partition with the deletion set, build layouts, retrain the affected shards, merge, test
(ultrare_amd.measure.cold_request).  Prints one JSON object with the wall time of each phase."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultrare_amd.measure import cold_request as measure  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shards', type=int, default=5)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--epochs', type=int, default=50)
    a = ap.parse_args()
    print(json.dumps(measure(a.shards, a.k, a.epochs)))


if __name__ == '__main__':
    main()
