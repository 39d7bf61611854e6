#!/usr/bin/env python3
"""One generic fused-MLP shape in a loop, for profiling (rocprofv3 / tools/pmc_collect.py --kernel mlp_forward_kernel):
    python tools/mlp_case.py tiny|hash|fourier|delta|hash96 [bf16|bf16x3] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_atlas_amd import config, neural_blocks as nb

case = sys.argv[1] if len(sys.argv) > 1 else "fourier"
config.set_precision(sys.argv[2] if len(sys.argv) > 2 else "bf16")
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
kw = {"tiny": dict(in_size=3, out=4, num_layers=6, hidden_size=256, init="xavier"),
      "hash": dict(in_size=3, out=65, num_layers=4, hidden_size=256, enc=nb.HashEncoder()),
      "fourier": dict(in_size=3, out=65, num_layers=6, hidden_size=256, enc=nb.FourierEncoder(input_dims=3, sigma=1 << 4)),
      "delta": dict(in_size=3, out=19, num_layers=5, hidden_size=256, enc=nb.HashEncoder()),
      # the mip `first` shape with its 96-wide latent read from HBM instead of generated in the prologue
      "hash96": dict(in_size=3, out=65, num_layers=4, hidden_size=256, enc=nb.HashEncoder(), latent_size=96)}[case]
m = nb.SkipConnMLP(**kw).cuda()
n = 4 * 1024 * 1024
x = torch.rand(n, 3, device="cuda") * 2 - 1
lat = torch.rand(n, 96, device="cuda") if case == "hash96" else None
with torch.no_grad():
    y = m(x, lat)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        y = m(x, lat)
    ev[1].record()
torch.cuda.synchronize()
print(case, f"{ev[0].elapsed_time(ev[1]) / iters:.3f} ms per {n} samples", float(y.double().abs().mean()))
