#!/usr/bin/env python3
"""Run-to-run reproducibility of the one-launch PlainNeRF + Positional / PosLinearView renderers (MODEL 7 / 8, round 6): a slab that
fills all 256 workgroups (2 sample groups each), rendered N times per head -- plv with explicit points and three refl_latent columns,
the D-NeRF form -- every output element against the first run; prints a checksum per head so that two builds (NA_LIB_PATH: the
timing-stress library) can be compared.      python tools/head_repeat.py [N=60]"""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_atlas_amd.nerf as nerf  # noqa: E402
import nerf_atlas_amd.refl as refl  # noqa: E402
from nerf_atlas_amd import config, ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda", 0)
    size, T = 800, 128
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, size, (300, 0, 48, size))   # 38 400 rays = 150 rays per sample group of the 512
    config.set_precision("f16x")
    bad = 0
    with torch.no_grad():
        for kind, n_rl in (("pos", 0), ("pos-linear-view", 0), ("pos-linear-view", 3)):
            torch.manual_seed(2)
            m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", bg="black")
            m.set_refl(refl.refl_kinds[kind](latent_size=64 + n_rl, act="upshifted", out_features=3))
            m = m.to(dev).eval()
            if n_rl:
                pts, ts, r_o, r_d, _ = nerf.compute_pts_ts(rays, 2.0, 6.0, T)
                g = torch.Generator(device=dev).manual_seed(5)
                rl = torch.randn((T,) + tuple(rays.shape[:-1]) + (n_rl,), device=dev, generator=g) * 0.3
                fn = lambda: (m.from_pts(pts, ts, r_o, r_d, refl_latent=rl, rays=rays), m.weights)  # noqa: E731
            else:
                fn = lambda: (m(rays, want_weights=True), m.weights)  # noqa: E731
            first, fw = fn()
            first, fw = first.clone(), fw.clone()
            assert torch.isfinite(first).all()
            diff = 0
            for _ in range(n):
                out, w = fn()
                if not (torch.equal(out, first) and torch.equal(w, fw)):
                    diff += 1
            bad += diff
            print(f"{kind} n_rl={n_rl}: {diff} of {n} runs differ; checksum {float(first.double().sum()):.9f} {float(fw.double().sum()):.6f}", flush=True)
        # the deformation network as one bf16x3 launch (MODEL 4 outside f16x, round 6): rows [T, R, 38] of `make dnerf`'s network
        torch.manual_seed(4)
        dyn = nerf.DynamicNeRF(canonical=nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64), spline=6, refl_latent=3).to(dev).eval()
        with torch.no_grad():
            for l in dyn.delta_estim._linears():
                l.weight.mul_(1.0).add_(torch.randn_like(l.weight) * 0.02)   # (the out layer is zero-initialised: give it rows)
        ts, _ = ops.compute_ts(2.0, 6.0, T, dev)
        packed = dyn.packed_deformation_ls("bf16x3")
        fn = lambda: ops.mlp_hash_ls(rays, ts, dyn.delta_estim.enc.tables(), packed, "bf16x3", 38)  # noqa: E731
        first = fn().clone()
        assert torch.isfinite(first).all()
        diff = sum(0 if torch.equal(fn(), first) else 1 for _ in range(n))
        bad += diff
        print(f"deformation rows bf16x3: {diff} of {n} runs differ; checksum {float(first.double().sum()):.9f} {float(first.double().abs().sum()):.6f}", flush=True)
    print(f"\n{bad} irreproducible runs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
