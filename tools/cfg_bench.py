#!/usr/bin/env python3
"""Slab timings of chosen BASELINE configs (bench.other_configs) without the rest of bench.py:
     python tools/cfg_bench.py 3 5m [--prec f16x,bf16x3] [--iters 5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    argv = sys.argv[1:]
    prec, iters = ("f16x", "bf16x3"), 5
    if "--prec" in argv:
        i = argv.index("--prec"); prec = tuple(argv[i + 1].split(",")); del argv[i:i + 2]
    if "--iters" in argv:
        i = argv.index("--iters"); iters = int(argv[i + 1]); del argv[i:i + 2]
    rows, secs, outside = bench.other_configs(torch.device("cuda", 0), precisions=prec, iters=iters, only=set(argv) or None)
    for r in rows + outside:
        r.pop("kernels", None)
        print(json.dumps(r))


if __name__ == "__main__":
    main()
