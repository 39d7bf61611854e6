#!/usr/bin/env python3
"""Headline benchmark: PlainNeRF (hash MLP + View head) volume rendering of ONE synthetic 800x800 frame at
128 samples/ray (BASELINE.json configs[1]; 81.92 M MLP-evaluated samples per step), on N GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = ray generation -> fused sample/hash/MLP/MLP/composite kernel -> finalize, for this rank's row band of
the frame, plus ONE RCCL gather of the finished RGB band to rank 0 (strong scaling: the frame is fixed, rays shard).
Synthetic data: camera c2w=[I|(0,0,4)], fov 0.6911, near 2, far 6; random-init weights of the reference's
architecture (default nn.Linear init / siren init / N(0,1) hash tables, seed 2).

Prints ONE JSON line (rank 0).  `value` is Msamples/s over all ranks with everything already in HBM.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = 1_192_960          # sum 2*in*out over both MLPs, unpadded
PEAK_BF16 = 2.5e15                   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# measured with rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes, KiB units, FETCH doubled for gfx950):
# (2 * 51177 + 80000) KiB per launch of the full frame -> 0.19 GB vs 97.7 TFLOP: the kernel is nowhere near HBM-bound
# (algorithmic: 15.4 MB rays + 8.4 MB tables + 1.3 MB weights read, 82 MB of per-block partials written)
HBM_TRAFFIC_FULL_FRAME = int((2 * 51177 + 80000) * 1024)
SIZE, STEPS_PER_RAY, FOV, NEAR, FAR = 800, 128, 0.6911, 2.0, 6.0


def build_model(device, seed=2):
    import nerf_atlas_amd.nerf as nerf
    torch.manual_seed(seed)
    m = nerf.PlainNeRF(steps=STEPS_PER_RAY, t_near=NEAR, t_far=FAR, intermediate_size=64, sigmoid_kind="upshifted",
                       bg="black")
    return m.to(device).eval()


def cpu_baseline(model, sample_hw=96):
    """The CPU oracle ("port" of the reference, torch fp32 on the host cores) on a bounded sample of the same
    workload: one sample_hw^2 tile of the 800^2 frame x 128 steps."""
    import oracle as O
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    focal = 0.5 * SIZE / math.tan(0.5 * FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    def run(hw):
        crop = (SIZE // 2 - hw // 2, SIZE // 2 - hw // 2, hw, hw)
        rays = O.nerf_camera_rays(O.pixel_grid(SIZE, crop), c2w, focal, SIZE)
        t0 = time.perf_counter()
        out = O.plain_nerf(params, rays, NEAR, FAR, STEPS_PER_RAY, "view", act="upshifted")
        return time.perf_counter() - t0, out, rays
    run(16)  # warm-up
    dt, out, rays = run(sample_hw)
    n = sample_hw * sample_hw * STEPS_PER_RAY
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample_hw}x{sample_hw} tile of the 800x800 frame x {STEPS_PER_RAY} steps "
                      f"({n} samples, {dt:.1f} s, torch-CPU fp32 oracle)"}, out, rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from nerf_atlas_amd import config, ops, dist as nd
    import torch.distributed as dist
    rank, world, local = nd.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    # NA_DIST_BACKEND=gloo lets several ranks share one GPU (flow test on a 1-GPU box); normally one rank per GPU
    ndev = torch.cuda.device_count()
    local = local % ndev if os.environ.get("NA_DIST_BACKEND") == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    config.set_precision(args.precision)

    model = build_model(dev)
    focal = 0.5 * SIZE / math.tan(0.5 * FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    r0, nrows = nd.row_bands(SIZE, world)[rank]
    prec = args.precision
    _, pf = model.first.packed(prec, "plain_first")
    _, pv = model.refl.mlp.packed(prec, "plain_view")
    tables = model.first.enc.tables()
    ts, _ = ops.compute_ts(NEAR, FAR, STEPS_PER_RAY, dev)
    R = nrows * SIZE
    ws = torch.empty(int(ops._lib.load().na_render_workspace_bytes(STEPS_PER_RAY, R)), device=dev, dtype=torch.uint8)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        rays = ops.raygen(c2w, focal, SIZE, (r0, 0, nrows, SIZE))
        if i is not None: ev[i][0].record()
        out, _, _ = ops.render_plain_view(rays, ts, tables, pf, pv, prec, "upshifted", "black", False, ws)
        if i is not None: ev[i][1].record()
        return nd.gather_bands(out.reshape(nrows, SIZE, 3), SIZE, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame = step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame = step(i)
    fence()
    dt = time.perf_counter() - t0
    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    tmax = torch.tensor([dt], device=red_dev)
    if world > 1: dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps  # this rank's fused kernel (+ finalize), HIP events
    kt = torch.tensor([kern_ms], device=red_dev)
    if world > 1: dist.all_reduce(kt, op=dist.ReduceOp.MAX)
    kern_ms = float(kt)

    if rank == 0:
        samples = SIZE * SIZE * STEPS_PER_RAY
        value = samples * args.steps / dt / 1e6
        launch_samples = R * STEPS_PER_RAY
        achieved = launch_samples * FLOP_PER_SAMPLE / (kern_ms * 1e-3)
        res = {
            "metric": "Msamples/sec (rays x samples) at 800^2 x 128", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if prec == "bf16" else "bf16x3 (2-way split bf16, 3 MFMA products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "PlainNeRF(hash MLP 4x256 + View head 4x256) 800x800 frame x 128 samples/ray, B=1",
                       "rays": SIZE * SIZE, "samples_per_ray": STEPS_PER_RAY, "flop_per_sample": FLOP_PER_SAMPLE,
                       "precision": prec, "parallelism": f"rays sharded in {world} row band(s) + 1 RCCL gather"},
            "roofline": {"bound": "mfma", "kernel": "render_plain_view_kernel", "achieved": round(achieved / 1e12, 2),
                         "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16, 4),
                         "kernel_ms": round(kern_ms, 3),
                         # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
                         # (profiles/r01/pmc_*.json; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM"); None when the
                         # shard differs from the profiled full frame
                         "traffic": HBM_TRAFFIC_FULL_FRAME if world == 1 else None,
                         "traffic_unit": "bytes/launch"},
        }
        if frame is not None:
            res["config"]["frame_checksum"] = round(float(frame.double().sum()), 3)
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, rays_cpu = cpu_baseline(model)
            res["cpu_baseline"] = cb
            # same tile through the HIP path: the benchmarked kernel is the parity-checked one
            got, _, _ = ops.render_plain_view(rays_cpu.to(dev), ts, tables, pf, pv, prec, "upshifted", "black")
            res["parity_sample_linf_vs_cpu_oracle"] = float((got.cpu() - ref).abs().max())
            # the other precision on the same frame (2 timed frames), so one line carries both modes
            other = "bf16x3" if prec == "bf16" else "bf16"
            _, pf2 = model.first.packed(other, "plain_first")
            _, pv2 = model.refl.mlp.packed(other, "plain_view")
            rays_full = ops.raygen(c2w, focal, SIZE, (0, 0, SIZE, SIZE))
            ops.render_plain_view(rays_full, ts, tables, pf2, pv2, other, "upshifted", "black", False, ws)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                ops.render_plain_view(rays_full, ts, tables, pf2, pv2, other, "upshifted", "black", False, ws)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t1) / 2
            got2, _, _ = ops.render_plain_view(rays_cpu.to(dev), ts, tables, pf2, pv2, other, "upshifted", "black")
            res["other_precision"] = {"precision": other, "value": round(samples / dt2 / 1e6, 2), "unit": "Msamples/s",
                                      "ms_per_frame": round(dt2 * 1e3, 3),
                                      "frac_of_bf16_mfma_peak": round(samples * FLOP_PER_SAMPLE / dt2 / PEAK_BF16, 4),
                                      "linf_vs_cpu_oracle": float((got2.cpu() - ref).abs().max())}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
