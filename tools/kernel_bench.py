#!/usr/bin/env python3
"""Achieved HBM bandwidth of the standalone (unfused) memory-bound kernels of the path -- the encode / sample /
composite operators SURVEY 8(d) prices against the HBM roofline (peak ~8 TB/s, MI355X_MICROARCH.md) -- and the
sample rate of the generic fused SkipConnMLP kernels of the other configs (TinyNeRF, VolSDF, D-NeRF deformation).

Timing: torch.cuda events on the current stream (the stream every kernel is launched on), 20 launches after 3
warm-ups, outputs pre-allocated by the wrappers (their torch.empty is a caching-allocator hit).  Algorithmic bytes
= compulsory reads + writes of the operator (inputs once, outputs once; the 8 MiB hash table is L2/MALL-resident and
not counted).  Run it under `rocprofv3 --kernel-trace --stats` to cross-check the per-kernel averages.

    python tools/kernel_bench.py [--rays 160000] [--steps 128] [--json out.json]
"""
import argparse
import json
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

PEAK_GBS = 8000.0


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=160000)   # a 400 x 400 tile
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from nerf_atlas_amd import ops
    import nerf_atlas_amd.neural_blocks as nb
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    side = int(math.isqrt(a.rays))
    R, T = side * side, a.steps
    N = R * T
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    crop = (0, 0, side, side)
    rays = ops.raygen(c2w, focal, size, crop)
    ts = ops.compute_ts(2.0, 6.0, T, dev)[0]
    pts = ops.compute_pts(rays, ts)
    flat = pts.reshape(-1, 3)
    tables = torch.randn(8, 65536, 4, device=dev)
    basis = torch.randn(3, 128, device=dev) * 16
    bands = torch.linspace(1, 64, 64, device=dev)
    density = torch.randn(T, 1, side, side, device=dev)
    rgb = torch.rand(T, 1, side, side, 3, device=dev)
    weights = torch.rand(T, 1, side, side, device=dev)
    dirs = rays[..., 3:].reshape(-1, 3).contiguous()
    est = torch.randn(N, 13, device=dev)
    tt = torch.rand(N, device=dev)
    rows = []

    def row(name, unit_count, unit, bytes_per_unit, fn):
        dt = timed(fn)
        gbs = unit_count * bytes_per_unit / dt / 1e9
        rows.append(dict(kernel=name, units=unit_count, unit=unit, bytes_per_unit=bytes_per_unit, us=round(dt * 1e6, 1),
                         GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK_GBS, 3),
                         Munits_per_s=round(unit_count / dt / 1e6, 1)))
        print(f"{name:28s} {dt * 1e6:9.1f} us  {gbs:8.1f} GB/s  ({gbs / PEAK_GBS:5.1%} of 8 TB/s)  {unit_count / dt / 1e6:9.1f} M{unit}/s")

    row("raygen", R, "rays", 24, lambda: ops.raygen(c2w, focal, size, crop))
    row("compute_pts", N, "samples", 12 + 24.0 / T, lambda: ops.compute_pts(rays, ts))
    row("hash_encode (+x)", N, "samples", 12 + 140, lambda: ops.hash_encode(flat, tables, True))
    row("fourier_encode 3->256", N // 8, "samples", 12 + 1024, lambda: ops.fourier_encode(flat[: N // 8], basis, 1.0))
    row("positional_encode 3->384", N // 8, "samples", 12 + 1536, lambda: ops.positional_encode(flat[: N // 8], bands))
    row("view_elaz", R, "rays", 12 + 8, lambda: ops.view_elaz(dirs))
    row("sigmoid(upshifted) x3", N, "samples", 24, lambda: ops.sigmoid(rgb, "upshifted"))
    row("composite (+alpha,weights)", N, "samples", 24 + 12.0 / T,
        lambda: ops.composite(density, rgb, ts, rays, softplus=True, bg="black"))
    row("composite (no aux)", N, "samples", 16 + 12.0 / T,
        lambda: ops.composite(density, rgb, ts, rays, softplus=True, bg="black", want_weights=False))
    row("integrate (weights x rgb)", N, "samples", 16 + 12.0 / T, lambda: ops.integrate(weights, rgb))
    row("laplace_density", N, "samples", 8, lambda: ops.laplace_density(density, torch.tensor(0.1, device=dev)))
    row("bezier_warp n=4", N, "samples", 13 * 4 + 12 + 4 + 12 + 12 + 4, lambda: ops.bezier_warp(est, flat, tt, 4))

    # generic fused SkipConnMLP (MFMA-bound; reported in samples/s and bf16-peak fraction, FLOP = sum 2*in*out)
    from nerf_atlas_amd import config
    mlps = {
        "tiny estim (3->4, 6x256)": (dict(in_size=3, out=4, num_layers=6, hidden_size=256, init="xavier"), None, 793088),
        "plain first (hash, 4x256)": (dict(in_size=3, out=65, num_layers=4, hidden_size=256, enc=nb.HashEncoder()), None, 596480),
        "volsdf sdf (fourier, 6x256)": (dict(in_size=3, out=65, num_layers=6, hidden_size=256,
                                             enc=nb.FourierEncoder(input_dims=3, sigma=1 << 4)), None, 1217536),
        "dnerf delta (hash, 5x256)": (dict(in_size=3, out=19, num_layers=5, hidden_size=256, enc=nb.HashEncoder()), None, 723456),
    }
    M = min(N, 8 * 1024 * 1024)
    x = flat[:M].contiguous()
    for prec in ("bf16", "f16", "bf16x3"):
        config.set_precision(prec)
        for name, (kw, _, flop) in mlps.items():
            m = nb.SkipConnMLP(**kw).to(dev)
            with torch.no_grad():
                dt = timed(lambda: m(x), iters=10)
            fl = M * flop / dt
            peak = 2.5e15 / 3 if prec == "bf16x3" else 2.5e15
            rows.append(dict(kernel=f"mlp_forward {name} [{prec}]", units=M, unit="samples", us=round(dt * 1e6, 1),
                             Msamples_per_s=round(M / dt / 1e6, 1), TFLOPs=round(fl / 1e12, 1),
                             frac_of_mfma_peak=round(fl / peak, 3)))
            print(f"mlp {name:30s} [{prec:6s}] {dt * 1e6:9.1f} us  {M / dt / 1e6:8.1f} Msamples/s  {fl / 1e12:7.1f} TFLOP/s "
                  f"({fl / peak:5.1%} of {'bf16/3' if prec == 'bf16x3' else 'bf16'} peak)")
    # whole models of the other BASELINE configs on one tile (operator chains except PlainNeRF(view) and TinyNeRF)
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.refl as refl
    import nerf_atlas_amd.sdf as sdf
    from nerf_atlas_amd.utils import load_mip
    import types
    side_m = min(side, 200)
    rays_m = ops.raygen(c2w, focal, size, (300, 300, side_m, side_m))
    Rm = side_m * side_m

    def volsdf(kind):
        under = sdf.sdf_kinds[kind](intermediate_size=64)
        r = refl.View(latent_size=64, act="upshifted", out_features=3)
        return nerf.VolSDF(sdf=sdf.SDF(under, r, isect=None, t_near=0.3, t_far=1.8), steps=T, t_near=0.3, t_far=1.8,
                           sigmoid_kind="upshifted")
    common = dict(steps=T, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted")
    def tiny_chain():
        m = nerf.TinyNeRF(**common)
        m._fusable = lambda: False  # the operator chain (generic fused MLP -> sigmoid -> composite) instead of the one kernel
        return m
    models = {
        "1 TinyNeRF (one kernel)": (lambda: nerf.TinyNeRF(**common), 793088, False),
        "1 TinyNeRF (operator chain)": (tiny_chain, 793088, False),
        "2 PlainNeRF(view) fused": (lambda: nerf.PlainNeRF(intermediate_size=64, **common), 1192960, False),
        "3 PlainNeRF + mip cylinder": (lambda: nerf.PlainNeRF(intermediate_size=64, mip=load_mip(types.SimpleNamespace(mip="cylinder")), **common), 1389568, False),
        "4 D-NeRF spline 6": (lambda: nerf.DynamicNeRF(canonical=nerf.PlainNeRF(intermediate_size=64, **common), spline=6), 1916416, True),
        "5 VolSDF mlp": (lambda: volsdf("mlp"), 1814016, False),
        "5 VolSDF siren": (lambda: volsdf("siren"), 1289728, False),
    }
    for prec in ("bf16", "f16", "bf16x3"):
        config.set_precision(prec)
        for name, (cons, flop, dyn) in models.items():
            try:
                m = cons().to(dev).eval()
                inp = (rays_m, torch.tensor([0.5], device=dev)) if dyn else rays_m
                with torch.no_grad():
                    dt = timed(lambda: m(inp), iters=5, warm=2)
            except Exception as e:  # noqa: BLE001
                print(f"model {name}: {type(e).__name__}: {e}")
                continue
            n = Rm * T
            peak = 2.5e15 / 3 if prec == "bf16x3" else 2.5e15
            rows.append(dict(kernel=f"model {name} [{prec}]", units=n, unit="samples", us=round(dt * 1e6, 1),
                             Msamples_per_s=round(n / dt / 1e6, 1), frac_of_mfma_peak=round(n * flop / dt / peak, 3)))
            print(f"model {name:30s} [{prec:6s}] {dt * 1e3:8.2f} ms  {n / dt / 1e6:8.1f} Msamples/s  ({n * flop / dt / peak:5.1%} of peak)")
    config.set_precision("bf16x3")
    # ---- N4: SDF marching of an 800 x 800 frame (src/march.py:27-47, 147-180), dense vs compacted.
    # "samples" = SDF network rows a DENSE march would evaluate (rays x iterations): the rate is per unit of the reference's
    # nominal work, so the compacted path's gain shows as a higher rate.  SDF: the SIREN network (fused MLP kernel) shifted so
    # that its zero level set is a blob in front of the camera, and the analytic two-sphere SDF of the fixtures.
    from nerf_atlas_amd import march, cameras
    cam = cameras.NeRFCamera(cam_to_world=torch.tensor([[[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]]]),
                             focal=focal).to(dev)
    fr = cam.sample_positions((0, 0, size, size), size=size, with_noise=False)[0]
    r_o, r_d = fr[..., :3].contiguous(), torch.nn.functional.normalize(fr[..., 3:], dim=-1)

    def analytic(p):
        a_ = torch.linalg.norm(p - torch.tensor([0.1, -0.2, 0.0], device=p.device), dim=-1) - 1.1
        b_ = torch.linalg.norm(p - torch.tensor([0.9, 0.6, 0.3], device=p.device), dim=-1) - 0.5
        return torch.minimum(a_, b_).unsqueeze(-1)
    torch.manual_seed(1)
    siren = sdf.SIREN(intermediate_size=0).to(dev).eval()

    def siren_blob(p):  # 0.4 * siren(p) + (|p| - 1): a perturbed unit sphere, evaluated by the fused MLP kernel
        return (0.4 * siren(p)[..., :1] + (torch.linalg.norm(p, dim=-1, keepdim=True) - 1.0)) * 0.5
    for name, fn in (("analytic two spheres", analytic), ("SIREN blob (fused MLP)", siren_blob)):
        for iters_ in (32,):
            res = {}
            for mode, comp in (("dense", False), ("compacted", True)):
                with torch.no_grad():
                    dt = timed(lambda: march.sphere_march(fn, r_o, r_d, iters=iters_, eps=1e-3, near=1.0, far=6.0, compact=comp), iters=3, warm=1)
                    out = march.sphere_march(fn, r_o, r_d, iters=iters_, eps=1e-3, near=1.0, far=6.0, compact=comp)
                st = dict(march.last_stats)
                res[mode] = (dt, st, out)
                n = st["dense_rows"]
                rows.append(dict(kernel=f"sphere_march {name} [{mode}]", units=n, unit="ray-iterations", us=round(dt * 1e6, 1),
                                 Msamples_per_s=round(n / dt / 1e6, 1), mlp_rows=st["mlp_rows"], iters_run=st["iters"]))
                print(f"sphere_march {name:24s} [{mode:9s}] {dt * 1e3:8.2f} ms  {n / dt / 1e6:8.1f} M ray-iterations/s  "
                      f"network rows {st['mlp_rows']} of {n} ({n / max(st['mlp_rows'], 1):.1f}x fewer), hit fraction {float(out[1].float().mean()):.2f}")
            same = all(torch.equal(x, y) for x, y in zip(res["dense"][2][:3], res["compacted"][2][:3]))
            print(f"   compacted == dense bit for bit: {same}; speed-up {res['dense'][0] / res['compacted'][0]:.2f}x")
            rows.append(dict(kernel=f"sphere_march {name} compacted == dense", identical=bool(same),
                             speedup=round(res["dense"][0] / res["compacted"][0], 2)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(rays=R, steps=T, samples=N, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
