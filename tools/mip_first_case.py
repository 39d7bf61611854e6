#!/usr/bin/env python3
"""PlainNeRF.first of the mip config (hash + 96-wide IPE latent) in its three forms: latent generated in the kernel prologue,
latent read from HBM, the latter on random points (tools/mip_first_case.py; used for the prologue ablations of DESIGN 3)."""
import os, sys, math, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_atlas_amd import config, ops, nerf
from nerf_atlas_amd.utils import load_mip
config.set_precision("bf16")
torch.manual_seed(0)
m = nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted",
                   mip=load_mip(types.SimpleNamespace(mip="cylinder"))).cuda().eval()
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device="cuda")
focal = 0.5 * 800 / math.tan(0.5 * 0.6911)
rays = ops.raygen(c2w, focal, 800, (300, 300, 200, 200))
pts, ts, r_o, r_d, _ = nerf.compute_pts_ts(rays, 2.0, 6.0, 128)
lazy = m.mip_latent(rays, ts)
mat = lazy.tensor()
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): y = f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
with torch.no_grad():
    print("first, IPE generated in the prologue (GEN kernel):", round(timed(lambda: m.first(pts, lazy)), 3), "ms")
    print("first, latent read from HBM (same shape, non-GEN) :", round(timed(lambda: m.first(pts, mat)), 3), "ms")
    x = torch.rand_like(pts.reshape(-1, 3)) * 2 - 1
    print("same non-GEN kernel on random points               :", round(timed(lambda: m.first(x, mat.reshape(-1, 96))), 3), "ms")
