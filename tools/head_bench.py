"""Throughput of the one-launch PlainNeRF renderers with the three colour heads of the reference's recipes (round 6):
View (MODEL 0, the headline), Positional (`make original`, MODEL 7), PosLinearView (`make dnerf`'s canonical model, MODEL 8; with
explicit points and three refl_latent columns as DynamicNeRF hands them over).  Whole 800 x 800 x 128 frame, f16x, HIP events on
the launch stream, 1 warm-up + `iters` launches; and the unfused operator chain (bf16x3) on a 200 x 800 slab next to it.

    python tools/head_bench.py [iters]
"""
import json
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nerf_atlas_amd.nerf as nerf  # noqa: E402
import nerf_atlas_amd.refl as refl  # noqa: E402
from nerf_atlas_amd import config, ops  # noqa: E402

PEAK = 2.5e15
FLOP = {"view": 1192960, "pos": 1410048, "pos-linear-view": 1131776}


def timed(fn, iters):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    size, T = 800, 128
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rows = []
    with torch.no_grad():
        for kind, n_rl, crop in (("view", 0, None), ("pos", 0, None), ("pos-linear-view", 0, None), ("pos-linear-view", 3, None),
                                 ("pos", 0, (0, 0, 200, 800)), ("pos-linear-view", 0, (0, 0, 200, 800))):
            torch.manual_seed(2)
            m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", bg="black")
            if kind != "view":
                m.set_refl(refl.refl_kinds[kind](latent_size=64 + n_rl, act="upshifted", out_features=3))
            m = m.to(dev).eval()
            rays = ops.raygen(c2w, focal, size, crop or (0, 0, size, size))
            R = rays.numel() // 6
            flop = FLOP[kind] + (n_rl * 2 * (256 * 2 + 128 * 2) if n_rl else 0)
            if crop is None:
                config.set_precision("f16x")
                if n_rl:
                    pts, ts, r_o, r_d, _ = nerf.compute_pts_ts(rays, 2.0, 6.0, T)
                    rl = torch.randn((T,) + tuple(rays.shape[:-1]) + (n_rl,), device=dev) * 0.3
                    fn = lambda: m.from_pts(pts, ts, r_o, r_d, refl_latent=rl, rays=rays)  # noqa: E731
                else:
                    fn = lambda: m(rays, want_weights=False)  # noqa: E731
                what = "one launch, f16x"
            else:
                config.set_precision("bf16x3")
                fn = lambda: m(rays)  # noqa: E731
                what = "unfused operator chain, bf16x3 (the path before round 6)"
            ms = timed(fn, iters if crop is None else 2)
            n = R * T
            rows.append({"head": kind, "n_rl": n_rl, "workload": f"{R} rays x {T}", "path": what, "ms": round(ms, 3),
                         "Msamples_s": round(n / ms / 1e3, 1), "flop_per_sample": flop, "frac": round(n * flop / (ms * 1e-3) / PEAK, 4)})
            print(json.dumps(rows[-1]), flush=True)
            del m
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "head_bench.json"), "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
