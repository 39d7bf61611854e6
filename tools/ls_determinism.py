#!/usr/bin/env python3
"""Run-to-run determinism of the layer-synchronous renderer: the same call N times (ts mode and explicit-position mode,
the parity modes f16x and bf16x3 and the bf16 fast mode, alpha / weights requested), every output compared bit for bit with the first run.
    python tools/ls_determinism.py [N]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_atlas_amd import nerf, config, cameras, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
bad = 0
c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
cam = cameras.NeRFCamera(cam_to_world=c2w, focal=0.5 * 800 / math.tan(0.5 * 0.6911)).cuda()
for prec in ("f16x", "bf16x3", "bf16"):
    config.set_precision(prec)
    m = nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
    with torch.no_grad():
        for mode, crop in (("ts", (380, 390, 37, 41)), ("ts", (300, 0, 24, 800)), ("pts", (380, 390, 40, 40))):
            rays = cam.sample_positions(crop, size=800)
            ts, _ = ops.compute_ts(2.0, 6.0, 128, "cuda")
            pts = None
            if mode == "pts":
                pts = ops.compute_pts(rays, ts) + 0.01 * torch.randn(128, *rays.shape[:-1], 3, device="cuda")
            ref = None
            for i in range(n):
                torch.empty(1 + (i * 7919) % 100000, device="cuda")  # perturb the allocator
                out = m._render_fused(rays, ts, True, pts=pts)
                cur = [t.clone() for t in out]
                if ref is None:
                    ref = cur
                elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
                    bad += 1
                    d = [float((a - b).abs().max()) for a, b in zip(cur, ref)]
                    nz = [int(((a - b) != 0).sum()) for a, b in zip(cur, ref)]
                    print(prec, mode, crop, "run", i, "differs: max", d, "count", nz)
# whole models through the generic fused MLP kernels (register engine), the warp and the standalone compositing
from nerf_atlas_amd import refl, sdf as nsdf
from nerf_atlas_amd.utils import load_mip
import types
rays = cam.sample_positions((380, 390, 40, 40), size=800)
times = torch.tensor([0.5], device="cuda")
for prec in ("f16x", "bf16x3", "bf16"):
    config.set_precision(prec)
    canon = nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted")
    models = {
        "dnerf": (nerf.DynamicNeRF(canonical=canon, spline=6), lambda m: m((rays, times))),
        "tiny": (nerf.TinyNeRF(steps=128, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted"), lambda m: m(rays)),
        "volsdf": (nerf.VolSDF(sdf=nsdf.SDF(nsdf.MLP(intermediate_size=64), refl.View(latent_size=64, act="upshifted", out_features=3),
                                            t_near=2.0, t_far=6.0), steps=128, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted"),
                   lambda m: m(rays)),
        "mip": (nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted",
                               mip=load_mip(types.SimpleNamespace(mip="cone"))), lambda m: m(rays)),
    }
    with torch.no_grad():
        for name, (m, run) in models.items():
            m = m.cuda().eval()
            if name == "dnerf":
                torch.nn.init.normal_(m.delta_estim.out.weight, std=0.05)
            ref = None
            for i in range(n):
                torch.empty(1 + (i * 7919) % 100000, device="cuda")
                out = run(m).clone()
                if ref is None:
                    ref = out
                elif not torch.equal(out, ref):
                    bad += 1
                    print(prec, name, "run", i, "differs: max", float((out - ref).abs().max()), "count", int((out != ref).sum()))
print(f"{bad} nondeterministic runs")
sys.exit(1 if bad else 0)
