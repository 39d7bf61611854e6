#!/usr/bin/env python3
"""A/B of the two fused PlainNeRF(view) engines on the headline frame (800x800x128): Msamples/s and fraction of the
bf16 MFMA peak, interleaved rounds in one process (tools/ls_bench.py [rounds])."""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main(rounds=3):
    from nerf_atlas_amd import ops
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    focal = 0.5 * bench.SIZE / math.tan(0.5 * bench.FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, bench.SIZE, (0, 0, bench.SIZE, bench.SIZE))
    ts, _ = ops.compute_ts(bench.NEAR, bench.FAR, bench.STEPS_PER_RAY, dev)
    tables = model.first.enc.tables()
    n = bench.SIZE * bench.SIZE * bench.STEPS_PER_RAY
    runs = {}
    for prec in ("bf16", "bf16x3"):
        _, pf = model.first.packed(prec, "plain_first")
        _, pv = model.refl.mlp.packed(prec, "plain_view")
        pl = model.packed_ls(prec)
        runs[("reg", prec)] = lambda pf=pf, pv=pv, prec=prec: ops.render_plain_view(rays, ts, tables, pf, pv, prec, "upshifted", "black")
        runs[("ls", prec)] = lambda pl=pl, prec=prec: ops.render_plain_view_ls(rays, ts, tables, pl, prec, "upshifted", "black")
    plh = model.packed_ls("f16")  # f16 operands: layer-synchronous engine only
    runs[("ls", "f16")] = lambda: ops.render_plain_view_ls(rays, ts, tables, plh, "f16", "upshifted", "black")
    outs = {}
    for k, f in runs.items():
        outs[k] = f()[0]
    torch.cuda.synchronize()
    for prec in ("bf16", "bf16x3"):
        print(f"{prec}: max |ls - reg| = {float((outs[('ls', prec)] - outs[('reg', prec)]).abs().max()):.3e}")
    for prec in ("bf16", "f16"):
        print(f"{prec}: max |ls {prec} - ls bf16x3| = {float((outs[('ls', prec)] - outs[('ls', 'bf16x3')]).abs().max()):.3e}")
    res = {k: [] for k in runs}
    for _ in range(rounds):
        for k, f in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 3)
    for k, v in res.items():
        ms = sorted(v)[len(v) // 2]
        print(f"{k[0]:4s} {k[1]:7s} {ms:8.2f} ms/frame  {n / ms / 1e3:8.1f} Msamples/s  "
              f"{n * bench.FLOP_PER_SAMPLE / (ms * 1e-3) / bench.PEAK_BF16:6.1%} of bf16 MFMA peak   all: {[round(x, 2) for x in v]}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
