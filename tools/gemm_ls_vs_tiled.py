#!/usr/bin/env python3
"""Forward and input gradient of a few Linear shapes through whichever training GEMMs the environment selects (default: the
layer-synchronous kernels; NA_TRAIN_GEMM=tiled: round 1's K-staged ones -- the choice is read once per process), saved to a file:
    python tools/gemm_ls_vs_tiled.py OUT.pt
tests/test_gpu_train_gemm.py runs it twice and compares the two files (same three products per k, another order of additions)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from nerf_atlas_amd import ops

SHAPES = [(4096, 256, 0, 256, "leaky_relu"), (4096 + 37, 256, 38, 256, "sin"), (8192, 38, 0, 256, "none"), (4096, 256, 0, 65, "leaky_relu"),
          (4096, 64, 38, 64, "sin")]


def main():
    out = {}
    for (N, in0, in1, o, act) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        x0 = torch.randn(N, in0, device="cuda", generator=g)
        x1 = torch.randn(N, in1, device="cuda", generator=g) if in1 else None
        W = torch.randn(o, in0 + in1, device="cuda", generator=g) * 0.06
        b = torch.randn(o, device="cuda", generator=g)
        gy = torch.randn(N, o, device="cuda", generator=g)
        y = ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=True)
        g0, g1 = ops.linear_dgrad(gy, W, x0, act, x1, True, in1 > 0)
        out[(N, in0, in1, o, act)] = (y.cpu(), g0.cpu(), None if g1 is None else g1.cpu())
    torch.save(out, sys.argv[1])


if __name__ == "__main__":
    main()
