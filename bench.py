#!/usr/bin/env python3
"""Headline benchmark: PlainNeRF (hash MLP + View head) volume rendering of ONE synthetic 800x800 frame at
128 samples/ray (BASELINE.json configs[1]; 81.92 M MLP-evaluated samples per step), on N GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no torchrun environment: spawns the N ranks itself (127.0.0.1, a free port)

A step = ray generation -> fused sample/hash/MLP/MLP/composite kernel -> finalize, for this rank's row band of
the frame, plus ONE RCCL gather of the finished RGB band to rank 0 (strong scaling: the frame is fixed, rays shard).
Synthetic data: camera c2w=[I|(0,0,4)], fov 0.6911, near 2, far 6; random-init weights of the reference's
architecture (default nn.Linear init / siren init / N(0,1) hash tables, seed 2).

Prints ONE JSON line (rank 0).  `value` is Msamples/s over all ranks with everything already in HBM.  The line carries
the precision modes on the same frame and `--steps` (the primary one -- f16x, the fastest mode within north_star's 1e-4 --
in the top-level fields, BASELINE's named bf16 under `other_precision`, the f16-operand mode under `f16_precision`, the
3-product bf16 split under `bf16x3_precision`), each with its own roofline object and its L-inf against the CPU oracle on
the CPU-baseline tile.
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
TRAFFIC_SOURCE = {("reg", "bf16"): "profiles/r01/pmc_render_plain_view_bf16_v3.json"}
for _rel in ("profiles/r02/hbm_traffic.json", "profiles/r03/hbm_traffic.json", "profiles/r04/hbm_traffic.json"):  # later rounds override
    try:
        with open(os.path.join(REPO, _rel)) as _f:
            for _k, _v in json.load(_f).items():
                HBM_TRAFFIC_FULL_FRAME[tuple(_k.split("/"))] = int(_v)
                TRAFFIC_SOURCE[tuple(_k.split("/"))] = _rel + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of this command, separate passes)"
    except (OSError, ValueError):
        pass
DTYPE_NAME = {"bf16": "bf16", "bf16x3": "bf16x3 (2-way split bf16, 3 MFMA products, fp32 accumulate)",
              "f16": "f16 (IEEE half operands, 1 MFMA product, fp32 accumulate)",
              "f16x": "f16x (f16 product + two MX-fp6 correction products on v_mfma_scale_f32_32x32x64_f8f6f4: 1.5 MFMA "
                      "products per k, fp32 accumulate -- hidden AND init / skip groups; the View MLP's 5-wide geometry chunk f16 hi + lo)"}


def clock_under_load(render, seconds=2.5):
    """Shader clock and socket power WHILE the fused kernel runs (rocm-smi polled from a thread during an untimed loop of the
    same launches): the 2.5 PFLOP/s the roofline is priced against assumes 2.4 GHz; under this kernel's load the firmware holds
    1.7-1.9 GHz at ~1.25-1.3 kW of the 1.4 kW cap (profiles/r04/power_probe_*.log), box to box.  None if rocm-smi is missing."""
    import re
    import subprocess
    import threading
    samples = []
    stop = threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                w = re.search(r"Power \(W\): ([0-9.]+)", out)
                if m:
                    samples.append((int(m.group(1)), float(w.group(1)) if w else None))
            except Exception:  # noqa: BLE001
                return
            stop.wait(0.3)
    try:
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        render()
        torch.cuda.synchronize()
        th.start()
        while time.perf_counter() - t0 < seconds:
            render()
            torch.cuda.synchronize()
        stop.set()
        th.join(timeout=6)
    except Exception:  # noqa: BLE001
        return None
    samples = [x for x in samples if x[0] > 500]  # (a sample taken between launches reads the idle level)
    if not samples:
        return None
    clk = statistics.median(x[0] for x in samples)
    pw = [x[1] for x in samples if x[1] is not None]
    return {"sclk_mhz": clk, "power_w": statistics.median(pw) if pw else None, "samples": len(samples)}


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


KERNELS_F16X = {"3": "ONE f16x launch of the layer-synchronous engine (MODEL 6: IPE groups generated in the kernel for the four Linears that take them)",
                "2-pos": "ONE f16x launch of the layer-synchronous engine (MODEL 7: both hash grids gathered in the kernel)",
                "2-plv": "ONE f16x launch of the layer-synchronous engine (MODEL 8)",
                "4": "deformation MLP: one bf16x3 launch of the layer-synchronous engine (MODEL 4) + canonical model f16x (one launch)",
                "4-plv": "deformation MLP: one bf16x3 launch of the layer-synchronous engine (MODEL 4, 38 output rows) + na_bezier_warp_latent + canonical model f16x (MODEL 8, one launch: warped points and the 3 latent columns by pitch)",
                "5m": "SDF MLP: ONE f16x launch of the layer-synchronous engine (Fourier features generated in the kernel) + View half f16x"}
FULL_FRAME = (0, 0, SIZE, SIZE)    # BASELINE's 1 x MI355X configs (1, 2, 3): the whole 800 x 800 frame, 81.92 M samples
OTHER_SLAB = (300, 0, 100, SIZE)   # BASELINE's 8 x MI355X configs (4, 5): ONE GPU's shard of the frame = a 100-row band (rows 300..399), 10.24 M samples


def train_algorithmic_bytes_per_sample(forward="ls"):
    """HBM bytes per sample a training step of PlainNeRF(view) has to move: every Linear's backward reads dY and the forward
    input once and writes the input gradient once (4 (out + 2 in) B); its forward writes its output row once (4 out B) and -- in the
    LAYER-BY-LAYER forward ("layers": csrc/train_fwd.hip) -- reads its input row(s) once (4 in B); the one-launch forward ("ls",
    round 6: csrc/ls_kernel.h MODEL 9) keeps the rows in LDS from layer to layer and reads none back.  fp32 rows; the two networks'
    Linears (src/nerf.py:320-324, src/refl.py:201-204).  Encoder, compositing and weight traffic are < 2 % and not counted."""
    first = [(38, 256), (256 + 38, 256), (256, 256), (256, 256), (256, 256), (256, 65)]
    view = [(69, 256), (256 + 69, 256), (256, 256), (256, 256), (256, 256), (256, 3)]
    return sum(4 * o + (4 * i if forward == "layers" else 0) + 4 * (o + 2 * i) for i, o in first + view)


def train_step(dev, crop=64, steps_per_ray=64, iters=10):
    """SURVEY 8(f) N1 in the driver-run line: one PlainNeRF(view) training step (forward + backward HIP kernels in the split-bf16
    parity-class arithmetic + the product's Adam, train.load_optim) on crop x crop rays x 64 samples (64: 262 144 samples,
    tools/train_bench.py's workload; 128: 1 048 576)"""
    import types
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import ops, train
    torch.manual_seed(0)
    focal = 0.5 * SIZE / math.tan(0.5 * FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    m = nerf.PlainNeRF(steps=steps_per_ray, t_near=NEAR, t_far=FAR, intermediate_size=64, sigmoid_kind="upshifted").to(dev)
    m.eval()  # deterministic sampling; gradients still flow
    opt = train.load_optim(types.SimpleNamespace(opt_kind="adam", learning_rate=2e-4, decay=0), m.parameters())
    target = torch.rand(1, crop, crop, 3, device=dev)
    c0 = (SIZE - crop) // 2

    def step():
        rays = ops.raygen(c2w, focal, SIZE, (c0, c0, crop, crop))
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(rays), target)
        loss.backward()
        opt.step()
        return loss
    t_all = time.perf_counter()
    with torch.enable_grad():
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters): step()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    n = crop * crop * steps_per_ray
    res = {"workload": f"PlainNeRF(view) training step, {crop} x {crop} rays x {steps_per_ray} samples, fwd + bwd + Adam",
           "dtype": "bf16x3", "samples_per_step": n, "ms_per_step": round(dt * 1e3, 2), "Msamples_s": round(n / dt / 1e6, 2),
           "iters": iters, "seconds": round(time.perf_counter() - t_all, 2)}
    # HBM-bound path.  `achieved` / `frac` = ALGORITHMIC bytes of the design that ran (train_algorithmic_bytes_per_sample) over
    # this run's step time; `traffic_gb_per_step` = the PMC counters of the committed profile of the 262 144-sample workload (NOT
    # measured by this run; keyed to that size only)
    from nerf_atlas_amd import config
    fwd = config.train_forward if 8192 <= n <= ops.TRAIN_LS_MAX_ROWS else "layers"
    res["forward"] = {"ls": "one launch of the layer-synchronous engine, every Linear's output rows written once (MODEL 9)",
                      "layers": "one training Linear per layer"}[fwd]
    alg = train_algorithmic_bytes_per_sample(fwd) * n / 1e9
    res["roofline"] = {"bound": "hbm", "algorithmic_gb": round(alg, 2), "achieved": round(alg / dt / 1e3, 2), "peak": 8.0, "unit": "TB/s",
                       "frac": round(alg / dt / 1e3 / 8.0, 3),
                       "algorithmic_gb_layer_by_layer": round(train_algorithmic_bytes_per_sample("layers") * n / 1e9, 2)}
    if n == 262144:
        for rel in ("profiles/r06/train_hbm.json", "profiles/r05/train_hbm.json"):
            try:
                with open(os.path.join(REPO, rel)) as f:
                    prof = json.load(f)
                if prof.get("train_forward", "layers") != fwd:
                    continue  # (a profile of the other forward: not this step's traffic)
                gb = float(prof["GB_per_step"])
                res["roofline"]["traffic_gb_per_step"] = gb
                res["roofline"]["traffic_source"] = rel + " (tools/train_hbm.py: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE over all kernels of the step, separate passes; a committed profile, not this run)"
                break
            except (OSError, ValueError, KeyError):
                continue
    return res


def coarse_fine(dev, rays, coarse=64, fine=128, iters=3):
    """BASELINE config 2 as it is named ("64 + 128"): a coarse pass of 64 shared steps, inverse-cdf resampling of 128 new positions
    per ray from its weights (na_resample_ts; the intended reading of the reference's dead sample_pdf), and a fine pass over
    the 192 per-ray steps -- three launches, one network, f16x.  `Msamples_s` counts the 64 + 192 network evaluations per ray."""
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import config
    torch.manual_seed(2)
    m = nerf.PlainNeRF(steps=coarse, t_near=NEAR, t_far=FAR, intermediate_size=64, sigmoid_kind="upshifted", bg="black").to(dev).eval()
    keep = config.precision
    config.set_precision("f16x")
    try:
        with torch.no_grad():
            m.forward_coarse_fine(rays, fine, want_weights=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                out = m.forward_coarse_fine(rays, fine, want_weights=False)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        R = rays.numel() // 6
        n = R * (coarse + coarse + fine)
        return {"workload": f"{SIZE}x{SIZE} frame: {coarse}-step pass (with weights) + resampling of {fine} positions per ray + "
                            f"{coarse + fine}-step pass with per-ray steps", "dtype": "f16x", "ms_per_frame": round(ms, 3),
                "evaluated_samples": n, "Msamples_s": round(n / ms / 1e3, 1), "finite": bool(torch.isfinite(out).all()),
                "frac": round(n * FLOP_PER_SAMPLE / (ms * 1e-3) / PEAK_BF16, 4)}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        config.set_precision(keep)


def other_configs(dev, precisions=("f16x", "bf16x3", "bf16"), iters=5, only=None):
    """BASELINE configs 1, 3, 4 and 5 (both SDF networks) and the reference's two shipped recipes with other colour heads (`make
    original`: PlainNeRF + Positional, makefile:8-13; `make dnerf`: D-NeRF over PlainNeRF + PosLinearView with --dyn-refl-latent 3,
    makefile:106-114) through the model layer: whole forward (every launch of the config's inference path), HIP events on the launch
    stream, one warm-up + `iters` timed calls per (config, precision).  The 1 x MI355X configs (1, 3, 2-pos, 2-plv) are timed on the WHOLE
    800 x 800 x 128 frame; the 8 x MI355X configs (4, 5) on one GPU's shard of it (a 100-row band = 10.24 M samples).  FLOP/sample =
    sum 2 * in * out over the config's MLPs (tools/kernel_bench.py uses the same numbers); `frac` is against the dense bf16
    MFMA peak for every precision.  "f16x" rows: the one-kernel renderers run f16x, the generic fused MLP launches of a config (D-NeRF:
    the deformation network) stay in bf16x3 -- `kernels` says which.  Rows whose arithmetic is OUTSIDE north_star's 1e-4 on the
    reference's goldens are returned separately (third value)."""
    import types
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.refl as refl
    import nerf_atlas_amd.sdf as sdf
    from nerf_atlas_amd import config, ops
    from nerf_atlas_amd.utils import load_mip
    T = STEPS_PER_RAY
    focal = 0.5 * SIZE / math.tan(0.5 * FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    common = dict(steps=T, t_near=NEAR, t_far=FAR, sigmoid_kind="upshifted")

    def volsdf(kind):
        r = refl.View(latent_size=64, act="upshifted", out_features=3)
        return nerf.VolSDF(sdf=sdf.SDF(sdf.sdf_kinds[kind](intermediate_size=64), r, isect=None, t_near=0.3, t_far=1.8),
                           steps=T, t_near=0.3, t_far=1.8, sigmoid_kind="upshifted")

    def plain_pos():
        m = nerf.PlainNeRF(intermediate_size=64, **common)
        m.set_refl(refl.refl_kinds["pos"](latent_size=64, act="upshifted", out_features=3))
        return m

    def plain_plv():
        m = nerf.PlainNeRF(intermediate_size=64, **common)
        m.set_refl(refl.refl_kinds["pos-linear-view"](latent_size=64, act="upshifted", out_features=3))
        return m

    def dnerf_plv():
        m = nerf.DynamicNeRF(canonical=nerf.PlainNeRF(intermediate_size=64, **common), spline=6, refl_latent=3)
        m.set_refl(refl.refl_kinds["pos-linear-view"](latent_size=m.intermediate_size, act="upshifted", out_features=3))
        return m
    # (name, constructor, FLOP/sample, takes (rays, t), crop, precisions)
    models = [
        ("1 TinyNeRF", lambda: nerf.TinyNeRF(**common), 793088, False, FULL_FRAME, precisions),
        ("2-pos PlainNeRF + Positional head (`make original`)", plain_pos, 1410048, False, FULL_FRAME, ("f16x",)),
        ("2-plv PlainNeRF + PosLinearView head (the canonical model of `make dnerf`, alone)", plain_plv, 1131776, False, FULL_FRAME, ("f16x",)),
        ("3 PlainNeRF + mip (cylinder IPE)", lambda: nerf.PlainNeRF(intermediate_size=64, mip=load_mip(types.SimpleNamespace(mip="cylinder")), **common), 1389568, False, FULL_FRAME, precisions),
        ("4 D-NeRF (spline 6) at t = 0.5", lambda: nerf.DynamicNeRF(canonical=nerf.PlainNeRF(intermediate_size=64, **common), spline=6), 1916416, True, OTHER_SLAB, precisions),
        ("4-plv D-NeRF (spline 6, refl_latent 3) over PlainNeRF + PosLinearView (`make dnerf`) at t = 0.5", dnerf_plv, 1855232, True, OTHER_SLAB, ("f16x",)),
        ("5m VolSDF, Fourier-MLP SDF", lambda: volsdf("mlp"), 1814016, False, OTHER_SLAB, precisions),
        ("5 VolSDF, SIREN SDF", lambda: volsdf("siren"), 1289728, False, OTHER_SLAB, precisions),
    ]
    rows, outside = [], []
    keep = config.precision
    t_all = time.perf_counter()
    for name, cons, flop, dyn, crop, precs in models:
        key = name.split()[0]
        if only is not None and key not in only:
            continue
        torch.manual_seed(2)
        try:
            m = cons().to(dev).eval()
        except Exception as e:  # noqa: BLE001
            rows.append({"config": name, "error": f"{type(e).__name__}: {e}"})
            continue
        rays = ops.raygen(c2w, focal, SIZE, crop)
        n = crop[2] * crop[3] * T
        what = (f"whole {SIZE}x{SIZE} frame x {T} samples/ray ({n} samples)" if crop == FULL_FRAME else
                f"per-GPU shard of the 8-GPU config: {crop[2]}x{crop[3]} row band x {T} samples/ray ({n} samples)")
        inp = (rays, torch.tensor([0.5], device=dev)) if dyn else rays
        plist = list(precs)
        if key == "4" and "f16x" in plist:
            plist.append("f16x+ls-deformation")  # the deformation network on the LS engine in f16x too (opt-in; outside 1e-4 on g9)
        for prec in plist:
            lsdef = prec == "f16x+ls-deformation"
            try:
                config.set_precision("f16x" if lsdef else prec)
                config.set_deformation_engine("ls" if lsdef else "ls-bf16x3")
                with torch.no_grad():
                    m(inp)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        m(inp)
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                row = {"config": name, "workload": what, "launches_timed": iters,
                       "dtype": prec, "kernels": ("canonical model: one kernel, f16x; deformation MLP: ONE f16x launch of the layer-synchronous engine "
                                                  "(opt-in: its rows carry 3x the error of the split-bf16 rows -- reference golden 1.0-1.6e-4 end to end, "
                                                  "trained model 2.4e-5)") if lsdef
                       else KERNELS_F16X.get(key, "f16x") if prec == "f16x" else prec,
                       "Msamples_s": round(n / ms / 1e3, 1), "kernel_ms": round(ms, 3),
                       "flop_per_sample": flop, "frac": round(n * flop / (ms * 1e-3) / PEAK_BF16, 4)}
                (outside if lsdef else rows).append(row)
            except Exception as e:  # noqa: BLE001
                rows.append({"config": name, "dtype": prec, "error": f"{type(e).__name__}: {e}"})
        del m
    config.set_precision(keep)
    config.set_deformation_engine("ls-bf16x3")
    return rows, round(time.perf_counter() - t_all, 2), outside


def self_spawn(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set like torch.distributed.run does), rank 0 inherits stdout and prints the one JSON
    line.  A rank that fails takes the others down (by PID) and its exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NA_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    while procs:
        for p in list(procs):
            c = p.poll()
            if c is None:
                continue
            procs.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in procs:
                    q.terminate()
        time.sleep(0.05)
    return rc


def measure_traffic(prec, engine, kernel_name):
    """roofline.traffic and roofline.mfma_busy_frac measured by THIS run: three child runs of this script under `rocprofv3 --pmc`
    -- FETCH_SIZE, WRITE_SIZE and (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) in SEPARATE passes, counters only (no trace domain), as
    MI355X_MICROARCH.md "HBM" prescribes -- each rendering the same full frame three times in the primary mode; bytes per launch
    = (2 x FETCH_SIZE + WRITE_SIZE) KiB (the guide's gfx950 correction: FETCH_SIZE counts half of a wide coalesced read);
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs) (tools/pmc_collect.py's derivation).
    Returns (bytes per launch or None, source text, mfma_busy_frac or None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "not measured: rocprofv3 not on PATH", None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "not measured: this run is itself under a profiler", None
    work = tempfile.mkdtemp(prefix="bench_traffic_")
    cmd = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--precision", prec, "--engine", engine, "--steps", "2", "--warmup", "1"]
    vals, why = {}, None
    try:
        for tag, ctrs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("MFMA", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])):
            out = os.path.join(work, tag)
            try:
                r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", out, "--"] + cmd, cwd="/tmp",
                                   env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=60)
            except subprocess.TimeoutExpired:
                why = f"rocprofv3 --pmc {tag}: timed out"
                break
            total, launches = {c: 0.0 for c in ctrs}, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_name in row.get("Kernel_Name", "") and row.get("Counter_Name") in total:
                        total[row["Counter_Name"]] += float(row["Counter_Value"])
                        launches.add(row.get("Dispatch_Id"))
            if not launches:
                why = f"rocprofv3 --pmc {tag}: no rows for {kernel_name} (rc={r.returncode})"
                break
            for c in ctrs:
                vals[c] = total[c] / len(launches)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    busy = None
    if vals.get("GRBM_GUI_ACTIVE"):
        busy = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, f"not measured: {why}", busy
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0), \
        "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of `bench.py --traffic-child` (same frame, same mode), (2 x FETCH + WRITE) KiB per launch", busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # primary = the FASTEST mode that meets north_star's 1e-4 L-inf bar: f16x (f16 + two MX-fp6 correction products, L-inf
    # ~1e-5; layer-synchronous engine).  Plain bf16 (BASELINE's named dtype, ~3e-3 L-inf) is timed on the same frame and
    # reported under `other_precision`, the f16 mode under `f16_precision`, the 3-product bf16 split (round 1-2's parity
    # mode) under `bf16x3_precision`.
    ap.add_argument("--precision", default=None, choices=["bf16", "bf16x3", "f16", "f16x"])
    ap.add_argument("--engine", default=None, choices=["ls", "reg"], help="fused renderer engine (default: config.engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the slab timings of BASELINE configs 1, 3, 4, 5")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic live (two rocprofv3 --pmc child runs, ~1 min)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # the child of measure_traffic: primary mode only, no JSON
    args = ap.parse_args()
    if args.engine == "reg" and args.precision in ("f16", "f16x"):
        ap.error(f"--precision {args.precision} exists on the layer-synchronous engine only (--engine ls)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args.gpus))

    from nerf_atlas_amd import config, ops, dist as nd
    import torch.distributed as dist
    rank, world, local = nd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or without any torchrun environment (bench.py then spawns its ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    # NA_DIST_BACKEND=gloo lets several ranks share one GPU (flow test on a 1-GPU box); normally one rank per GPU
    ndev = torch.cuda.device_count()
    if os.environ.get("NA_DIST_BACKEND") == "gloo":
        local = local % ndev
    elif local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local} but only {ndev} are visible (one rank per GPU; "
                         f"NA_DIST_BACKEND=gloo shares one GPU between ranks for flow tests)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.engine is not None:
        config.set_engine(args.engine)
    engine = config.engine
    if args.precision is None:
        args.precision = "f16x" if engine == "ls" else "bf16x3"

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
    # the group's first collective, with the rank / device / backend printed if it fails (the "nccl" = RCCL branch runs for the
    # first time on the driver's multi-GPU node); the count it returns goes into the JSON line
    rccl_ranks, backend = nd.first_collective(dev)
    gather = nd.BandGather(SIZE, 3, rank, world, dev)  # receive / send buffers of the frame gather, allocated once

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
            frame = gather(out.reshape(nrows, SIZE, 3))
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
                "traffic_unit": "bytes/launch",
                # (an earlier profile's figure; the primary mode's is replaced by a live measurement: measure_traffic)
                "traffic_source": TRAFFIC_SOURCE.get((engine, prec)) if world == 1 else None}

    prec = args.precision
    if args.traffic_child:  # (under rocprofv3 --pmc: the primary mode's launches only)
        timed(prec, args.steps, args.warmup)
        return
    dt, kern_ms, gath_ms, per_rank, frame = timed(prec, args.steps, args.warmup)
    # (now: with N > 1 `frame` is the gather's persistent receive buffer, which the next timed() call overwrites)
    checksum = round(float(frame.double().sum()), 3) if frame is not None else None
    other = "bf16x3" if prec == "bf16" else "bf16"
    dt2, kern2_ms, _, _, _ = timed(other, args.steps, 1)
    # third line: the f16-operand mode of the layer-synchronous engine (the fast mode's speed class, 11-bit operands)
    third = "f16" if (engine == "ls" and "f16" not in (prec, other)) else None
    if third is not None:
        dt3, kern3_ms, _, _, _ = timed(third, args.steps, 1)
    # fourth: the other parity mode (3-product bf16 split) when the primary one is f16x
    fourth = "bf16x3" if "bf16x3" not in (prec, other) else None
    if fourth is not None:
        dt4, kern4_ms, _, _, _ = timed(fourth, args.steps, 1)

    # N > 1: what one GPU needs for the WHOLE frame, measured in this very run on rank 0 after the timed regions (the other
    # ranks wait at the final barrier), so that the line explains itself: scaling_efficiency = that / (N x slowest rank's kernel)
    full_frame_ms = None
    if world > 1:
        if rank == 0:
            ws_full = torch.empty(max(int(lib.na_render_workspace_bytes(STEPS_PER_RAY, SIZE * SIZE)),
                                      int(lib.na_render_ls_workspace_bytes(STEPS_PER_RAY, SIZE * SIZE))), device=dev, dtype=torch.uint8)
            rays_all = ops.raygen(c2w, focal, SIZE, (0, 0, SIZE, SIZE))
            if engine == "ls":
                pk = model.packed_ls(prec)
                full = lambda: ops.render_plain_view_ls(rays_all, ts, tables, pk, prec, "upshifted", "black", False, ws_full)[0]
            else:
                _, pf = model.first.packed(prec, "plain_first")
                _, pv = model.refl.mlp.packed(prec, "plain_view")
                full = lambda: ops.render_plain_view(rays_all, ts, tables, pf, pv, prec, "upshifted", "black", False, ws_full)[0]
            full()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): full()
            e1.record()
            torch.cuda.synchronize()
            full_frame_ms = e0.elapsed_time(e1) / 3
        fence()

    if rank == 0:
        samples = SIZE * SIZE * STEPS_PER_RAY
        value = samples * args.steps / dt / 1e6
        primary_roofline = roofline(prec, kern_ms)
        if world == 1 and not args.no_traffic:
            t_live, t_src, mfma_busy = measure_traffic(prec, engine, primary_roofline["kernel"])
            # ONE method: the live PMC child passes of this very command, or null with the reason -- an earlier
            # profile's constant is never substituted for the primary mode
            primary_roofline["traffic"], primary_roofline["traffic_source"] = t_live, t_src
            if mfma_busy is not None:
                primary_roofline["mfma_busy_frac"] = mfma_busy
        res = {
            "metric": "Msamples/sec (rays x samples) at 800^2 x 128", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NAME[prec],
            "data": "synthetic",
            "config": {"workload": "PlainNeRF(hash MLP 4x256 + View head 4x256) 800x800 frame x 128 samples/ray, B=1",
                       "rays": SIZE * SIZE, "samples_per_ray": STEPS_PER_RAY, "flop_per_sample": FLOP_PER_SAMPLE,
                       "precision": prec, "engine": engine,
                       "parallelism": f"rays sharded in {world} row band(s) + 1 RCCL gather"},
            "roofline": primary_roofline,
            "per_rank_kernel_ms": [round(x, 3) for x in per_rank], "gather_ms": round(gath_ms, 3),
            # torch.distributed backend of the run ("nccl" IS RCCL on ROCm; "none" for one process) and the ranks an
            # all-reduce of ones on the device counted; the frame gather's implementation (dist.gather, or all_gather if the
            # backend build lacks it)
            "backend": backend, "rccl_ranks": rccl_ranks, "gather_impl": gather.impl if world > 1 else "none",
        }
        if full_frame_ms is not None:
            res["single_gpu_full_frame_kernel_ms"] = round(full_frame_ms, 3)
            res["scaling_efficiency"] = round(full_frame_ms / (world * kern_ms), 4)   # kernel time only; `value` also pays the gather
        if checksum is not None:
            res["config"]["frame_checksum"] = checksum
        if world == 1:
            rays_full = ops.raygen(c2w, focal, SIZE, (r0, 0, nrows, SIZE))
            rfn = renderer(prec)
            clk = clock_under_load(lambda: rfn(rays_full))
            if clk is not None:
                # the same achieved rate against the MFMA peak AT THE CLOCK THE CHIP HELD (2.5 PFLOP/s is 2.4 GHz x 256 CUs x 4 x 512
                # MAC/clk x 2): an extra key -- `roofline.frac` stays priced against the nominal peak
                clk["frac_of_mfma_peak_at_this_clock"] = round(res["roofline"]["frac"] * 2400.0 / clk["sclk_mhz"], 4)
                res["clock_under_load"] = clk
        res["other_precision"] = {"precision": other, "dtype": DTYPE_NAME[other], "value": round(samples * args.steps / dt2 / 1e6, 2),
                                  "unit": "Msamples/s", "steps": args.steps, "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                                  "roofline": roofline(other, kern2_ms)}
        if third is not None:
            res["f16_precision"] = {"precision": third, "dtype": DTYPE_NAME[third], "value": round(samples * args.steps / dt3 / 1e6, 2),
                                    "unit": "Msamples/s", "steps": args.steps, "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                                    "roofline": roofline(third, kern3_ms)}
        if fourth is not None:
            res["bf16x3_precision"] = {"precision": fourth, "dtype": DTYPE_NAME[fourth], "value": round(samples * args.steps / dt4 / 1e6, 2),
                                       "unit": "Msamples/s", "steps": args.steps, "ms_per_step": round(dt4 / args.steps * 1e3, 3),
                                       "roofline": roofline(fourth, kern4_ms)}
        if world == 1 and not args.no_other_configs:
            res["other_configs"], res["other_configs_s"], res["outside_tolerance"] = other_configs(dev)
            res["train_step"] = train_step(dev)
            res["train_step_1m"] = train_step(dev, crop=128, iters=5)
            res["coarse_fine"] = coarse_fine(dev, ops.raygen(c2w, focal, SIZE, (0, 0, SIZE, SIZE)))
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, rays_cpu = cpu_baseline(model)
            res["cpu_baseline"] = cb
            # same tile through the HIP path: the benchmarked kernels are the parity-checked ones
            rd = rays_cpu.to(dev)
            res["parity_sample_linf_vs_cpu_oracle"] = float((renderer(prec)(rd).cpu() - ref).abs().max())
            res["other_precision"]["linf_vs_cpu_oracle"] = float((renderer(other)(rd).cpu() - ref).abs().max())
            if third is not None:
                res["f16_precision"]["linf_vs_cpu_oracle"] = float((renderer(third)(rd).cpu() - ref).abs().max())
            if fourth is not None:
                res["bf16x3_precision"]["linf_vs_cpu_oracle"] = float((renderer(fourth)(rd).cpu() - ref).abs().max())
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
