#!/usr/bin/env python3
"""One shape of bench.py's train leg on its own (for rocprofv3 --stats A/B runs): prints the bench_train record."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402

if __name__ == "__main__":
    import torch
    from gnode import _lib
    n, m, B, reps = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (75000, 500000, 4, 3)))
    dev = torch.device("cuda:0")
    print(json.dumps(bench.bench_train(_lib.load(), dev, n, m, B, 64, 30, 0.5, reps)))
