#!/usr/bin/env python3
"""Training forward of a 256 wide Linear at N = 262 144, cold loop: the packed entry point (csrc/train_fwd.hip unless
NA_TRAIN_FUSED_FWD=0) against the unpacked one (lsnt::kernel<0>).    python tools/fwd_bench.py [in1=0] [act=leaky_relu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import ops

in1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
act = sys.argv[2] if len(sys.argv) > 2 else "leaky_relu"
N, dev, R = 262144, "cuda", 6
torch.manual_seed(0)
xs = [torch.randn(N, 256, device=dev) for _ in range(R)]
x1 = [torch.randn(N, in1, device=dev) for _ in range(R)] if in1 else [None] * R
W = torch.randn(256, 256 + in1, device=dev) / 16
b = torch.randn(256, device=dev)
(pf,) = ops.train_pack_many([(W, False)])


def timeit(fn, iters=5):
    for i in range(R): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for i in range(R): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (iters * R) * 1e6


t_p = timeit(lambda i: ops.linear_f32(xs[i], W, b, pre_act=act, x1=x1[i], split_bf16=True, packed=pf))
t_u = timeit(lambda i: ops.linear_f32(xs[i], W, b, pre_act=act, x1=x1[i], split_bf16=True))
mb = N * (512 + in1) * 4 / 1e6
print(f"in1={in1} act={act}: packed entry {t_p:.1f} us ({mb / t_p:.2f} TB/s of {mb:.0f} MB)   unpacked (streaming kernel + pack) {t_u:.1f} us")
