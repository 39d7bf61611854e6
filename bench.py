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

Prints ONE JSON line (rank 0).  `value` is Msamples/s over all ranks with everything already in HBM.  The line carries
the precision modes on the same frame and `--steps` (the primary one in the top-level fields, BASELINE's named bf16
under `other_precision`, the layer-synchronous engine's f16-operand mode under `f16_precision`), each with its own
roofline object and its L-inf against the CPU oracle on the CPU-baseline tile.
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = 1_192_960          # sum 2*in*out over both MLPs, unpadded
PEAK_BF16 = 2.5e15                   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
SIZE, STEPS_PER_RAY, FOV, NEAR, FAR = 800, 128, 0.6911, 2.0, 6.0
# HBM bytes per full-frame launch of the fused kernel, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
# command (separate passes, KiB units, FETCH doubled for gfx950 per MI355X_MICROARCH.md "HBM"); profiles/r02/pmc_*.json.
# None = not measured for that (engine, precision).
HBM_TRAFFIC_FULL_FRAME = {("reg", "bf16"): int((2 * 51177 + 80000) * 1024)}
try:
    with open(os.path.join(REPO, "profiles", "r02", "hbm_traffic.json")) as _f:
        for _k, _v in json.load(_f).items():
            HBM_TRAFFIC_FULL_FRAME[tuple(_k.split("/"))] = int(_v)
except (OSError, ValueError):
    pass
DTYPE_NAME = {"bf16": "bf16", "bf16x3": "bf16x3 (2-way split bf16, 3 MFMA products, fp32 accumulate)",
              "f16": "f16 (IEEE half operands, 1 MFMA product, fp32 accumulate)"}


def build_model(device, seed=2):
    import nerf_atlas_amd.nerf as nerf
    torch.manual_seed(seed)
    m = nerf.PlainNeRF(steps=STEPS_PER_RAY, t_near=NEAR, t_far=FAR, intermediate_size=64, sigmoid_kind="upshifted",
                       bg="black")
    return m.to(device).eval()


def cpu_baseline(model, sample_hw=100, repeats=3):
    """The CPU oracle ("port" of the reference, torch fp32 on the host cores) on a bounded sample of the same workload
    (BASELINE.md 3b): one 100x100 tile of the 800^2 frame x 128 steps, `repeats` runs after a discarded first one,
    median; the thread count is picked by a short sweep on a 32x32 tile (all cores is not the fastest setting)."""
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
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    sweep = {}
    # (all cores is never the fastest setting for this oracle and costs minutes on a 256-thread host: not swept)
    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu}):
        torch.set_num_threads(th)
        run(16)
        sweep[th] = min(run(32)[0] for _ in range(2))
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    run(sample_hw)  # discarded
    times = []
    for _ in range(repeats):
        dt, out, rays = run(sample_hw)
        times.append(dt)
    torch.set_num_threads(default_threads)
    med = statistics.median(times)
    n = sample_hw * sample_hw * STEPS_PER_RAY
    return {"value": round(n / med / 1e6, 4), "unit": "Msamples/s", "cores": best, "host_cores": ncpu, "kind": "port",
            "thread_sweep_s_32x32": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": f"{sample_hw}x{sample_hw} tile of the 800x800 frame x {STEPS_PER_RAY} steps ({n} samples), "
                      f"median of {repeats} runs after one discarded ({', '.join(f'{t:.1f}' for t in times)} s), "
                      f"torch-CPU fp32 oracle on {best} threads"}, out, rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # primary = the mode that meets north_star's 1e-4 L-inf bar (split bf16, fp32-class); plain bf16 (BASELINE's named
    # dtype, ~3e-3 L-inf) is timed on the same frame and reported under `other_precision`
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "bf16x3", "f16"])
    ap.add_argument("--engine", default=None, choices=["ls", "reg"], help="fused renderer engine (default: config.engine)")
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
    if args.engine is not None:
        config.set_engine(args.engine)
    engine = config.engine

    model = build_model(dev)
    focal = 0.5 * SIZE / math.tan(0.5 * FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    r0, nrows = nd.row_bands(SIZE, world)[rank]
    tables = model.first.enc.tables()
    ts, _ = ops.compute_ts(NEAR, FAR, STEPS_PER_RAY, dev)
    R = nrows * SIZE
    lib = ops._lib.load()
    ws_bytes = max(int(lib.na_render_workspace_bytes(STEPS_PER_RAY, R)), int(lib.na_render_ls_workspace_bytes(STEPS_PER_RAY, R)))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)

    def renderer(prec):
        """rays -> rgb through the fused kernel of the selected engine (weights packed once, outside the timed region)"""
        if engine == "ls":
            packed = model.packed_ls(prec)
            return lambda rays: ops.render_plain_view_ls(rays, ts, tables, packed, prec, "upshifted", "black", False, ws)[0]
        _, pf = model.first.packed(prec, "plain_first")
        _, pv = model.refl.mlp.packed(prec, "plain_view")
        return lambda rays: ops.render_plain_view(rays, ts, tables, pf, pv, prec, "upshifted", "black", False, ws)[0]

    def fence():
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
        torch.cuda.synchronize()

    red_dev = dev if (world == 1 or dist.get_backend() == "nccl") else "cpu"

    def rmax(x):
        t = torch.tensor([x], device=red_dev, dtype=torch.float64)
        if world > 1: dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def timed(prec, steps, warmup):
        """W untimed + K timed steps; returns (wall s, max-over-ranks kernel ms, max-over-ranks gather ms, per-rank kernel ms,
        frame).  The kernel bracket = HIP events on the launch stream around fused kernel + finalize."""
        render = renderer(prec)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

        def step(i=None):
            rays = ops.raygen(c2w, focal, SIZE, (r0, 0, nrows, SIZE))
            if i is not None: ev[i][0].record()
            out = render(rays)
            if i is not None: ev[i][1].record()
            frame = nd.gather_bands(out.reshape(nrows, SIZE, 3), SIZE, rank, world)
            if i is not None: ev[i][2].record()
            return frame
        for _ in range(warmup):
            frame = step()
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            frame = step(i)
        fence()
        dt = rmax(time.perf_counter() - t0)
        kern = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        gath = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        per_rank = [kern]
        if world > 1:
            lst = [None] * world
            dist.all_gather_object(lst, kern)
            per_rank = lst
        return dt, rmax(kern), rmax(gath), per_rank, frame

    def roofline(prec, kern_ms):
        launch_samples = R * STEPS_PER_RAY
        achieved = launch_samples * FLOP_PER_SAMPLE / (kern_ms * 1e-3)
        kname = "render_ls_kernel" if engine == "ls" else "render_plain_view_kernel"
        return {"bound": "mfma", "kernel": kname, "achieved": round(achieved / 1e12, 2), "peak": PEAK_BF16 / 1e12,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16, 4), "kernel_ms": round(kern_ms, 3),
                # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE
                # doubled per MI355X_MICROARCH.md "HBM"); None when the shard differs from the profiled full frame
                "traffic": HBM_TRAFFIC_FULL_FRAME.get((engine, prec)) if world == 1 else None,
                "traffic_unit": "bytes/launch"}

    prec = args.precision
    dt, kern_ms, gath_ms, per_rank, frame = timed(prec, args.steps, args.warmup)
    other = "bf16x3" if prec == "bf16" else "bf16"
    dt2, kern2_ms, _, _, _ = timed(other, args.steps, 1)
    # third line: the f16-operand mode of the layer-synchronous engine (the fast mode's speed class, 11-bit operands)
    third = "f16" if (engine == "ls" and "f16" not in (prec, other)) else None
    if third is not None:
        dt3, kern3_ms, _, _, _ = timed(third, args.steps, 1)

    if rank == 0:
        samples = SIZE * SIZE * STEPS_PER_RAY
        value = samples * args.steps / dt / 1e6
        res = {
            "metric": "Msamples/sec (rays x samples) at 800^2 x 128", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NAME[prec],
            "data": "synthetic",
            "config": {"workload": "PlainNeRF(hash MLP 4x256 + View head 4x256) 800x800 frame x 128 samples/ray, B=1",
                       "rays": SIZE * SIZE, "samples_per_ray": STEPS_PER_RAY, "flop_per_sample": FLOP_PER_SAMPLE,
                       "precision": prec, "engine": engine,
                       "parallelism": f"rays sharded in {world} row band(s) + 1 RCCL gather"},
            "roofline": roofline(prec, kern_ms),
            "per_rank_kernel_ms": [round(x, 3) for x in per_rank], "gather_ms": round(gath_ms, 3),
        }
        if frame is not None:
            res["config"]["frame_checksum"] = round(float(frame.double().sum()), 3)
        res["other_precision"] = {"precision": other, "dtype": DTYPE_NAME[other], "value": round(samples * args.steps / dt2 / 1e6, 2),
                                  "unit": "Msamples/s", "steps": args.steps, "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                                  "roofline": roofline(other, kern2_ms)}
        if third is not None:
            res["f16_precision"] = {"precision": third, "dtype": DTYPE_NAME[third], "value": round(samples * args.steps / dt3 / 1e6, 2),
                                    "unit": "Msamples/s", "steps": args.steps, "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                                    "roofline": roofline(third, kern3_ms)}
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, rays_cpu = cpu_baseline(model)
            res["cpu_baseline"] = cb
            # same tile through the HIP path: the benchmarked kernels are the parity-checked ones
            rd = rays_cpu.to(dev)
            res["parity_sample_linf_vs_cpu_oracle"] = float((renderer(prec)(rd).cpu() - ref).abs().max())
            res["other_precision"]["linf_vs_cpu_oracle"] = float((renderer(other)(rd).cpu() - ref).abs().max())
            if third is not None:
                res["f16_precision"]["linf_vs_cpu_oracle"] = float((renderer(third)(rd).cpu() - ref).abs().max())
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
