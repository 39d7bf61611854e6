#!/usr/bin/env python3
"""Fused backward (csrc/train_bwd.hip) against the two launches it replaces, N = 262 144, in a cold loop (different tensors per
call: the 256-MB memory-side cache must not hold the operands).    python tools/bwd_bench.py [out=256] [act=leaky_relu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops

out = int(sys.argv[1]) if len(sys.argv) > 1 else 256
act = sys.argv[2] if len(sys.argv) > 2 else "leaky_relu"
N, dev, R = 262144, "cuda", 6
torch.manual_seed(0)
xs = [torch.randn(N, 256, device=dev) for _ in range(R)]
gs = [torch.randn(N, out, device=dev) for _ in range(R)]
W = torch.randn(out, 256, device=dev) / 16
(pt,) = ops.train_pack_many([(W, True)])


def timeit(fn, iters=5):
    for i in range(R): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for i in range(R): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (iters * R) * 1e6


def two(i):
    ops.linear_dgrad(gs[i], W, xs[i], act, packed_t=pt)
    ops.linear_wgrad(xs[i], gs[i], act, split_bf16=True)


t_f = timeit(lambda i: ops.linear_bwd_fused(gs[i], xs[i], act, pt))
t_2 = timeit(two)
mb = N * (out + 512) * 4 / 1e6
print(f"out={out} act={act}: fused {t_f:.1f} us ({mb / t_f:.2f} TB/s of {mb:.0f} MB)   two launches {t_2:.1f} us")
