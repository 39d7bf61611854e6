#!/usr/bin/env python3
"""Run-to-run reproducibility of the layer-synchronous renderer on a slab that fills every workgroup (24 x 800 rays: 512
sample groups, ~37 rays each): the same call N times, colour / alpha / weights compared element by element with the
per-element median over the runs.  This is the test that exposed the timing-dependent differences of DESIGN 3b
"reproducibility" (16 samples of one block, lanes 16..31, off by ~1e-3 relative in one run of ~20..10^4).

    python tools/ls_repeat.py [f16x|bf16x3|bf16|f16] [--tiny | --volsdf-mlp | --volsdf-siren] [N]
    python tools/ls_repeat.py variants        # here: timing-stress builds (group lag 3 / 5 / 9, no XCD-aware order) under
                                              # gpurun_ablate/repeat_<name>/; run each with  --lib <dir>
"""
import math
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"lag3": "-DNA_LS_LAG_OVERRIDE=3", "lag5_noxcd": "-DNA_LS_LAG_OVERRIDE=5 -DNA_LS_NO_XCD_MAP=1", "lag9": "-DNA_LS_LAG_OVERRIDE=9",
            "noxcd": "-DNA_LS_NO_XCD_MAP=1"}


def variants():
    import shutil
    for name, flags in VARIANTS.items():
        d = os.path.join(REPO, "gpurun_ablate", "repeat_" + name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        shutil.copytree(os.path.join(REPO, "nerf_atlas_amd"), os.path.join(d, "nerf_atlas_amd"),
                        ignore=shutil.ignore_patterns("build", "*.so", "__pycache__"))
        shutil.copytree(os.path.join(REPO, "include"), os.path.join(d, "include"))
        subprocess.run([sys.executable, "-m", "nerf_atlas_amd.build"], cwd=d, check=True, env={**os.environ, "NA_EXTRA_HIPCC_FLAGS": flags})
        print("built", d)


def main(argv):
    lib = None
    if "--lib" in argv:
        i = argv.index("--lib")
        lib = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    sys.path.insert(0, lib or REPO)
    import torch
    from nerf_atlas_amd import nerf, config, cameras, ops
    precs = [argv[0]] if argv and argv[0] in ("bf16", "bf16x3", "f16", "f16x") else ["f16x", "bf16x3", "bf16", "f16"]
    n = int([x for x in argv if x.isdigit()][-1]) if any(x.isdigit() for x in argv) else 200
    torch.manual_seed(0)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    cam = cameras.NeRFCamera(cam_to_world=c2w, focal=0.5 * 800 / math.tan(0.5 * 0.6911)).cuda()
    rays = cam.sample_positions((300, 0, 24, 800), size=800)
    R = rays.numel() // 6
    ts, _ = ops.compute_ts(2.0, 6.0, 128, "cuda")
    total_bad = 0
    tiny = "--tiny" in argv  # the TinyNeRF schedule of the same engine
    for prec in precs:
        config.set_precision(prec)
        volsdf = "mlp" if "--volsdf-mlp" in argv else "siren" if "--volsdf-siren" in argv else None
        if volsdf:
            from nerf_atlas_amd import sdf as nsdf, refl
            m = nerf.VolSDF(sdf=nsdf.SDF(nsdf.sdf_kinds[volsdf](intermediate_size=64), refl.View(latent_size=64, act="upshifted", out_features=3),
                                         t_near=2.0, t_far=6.0), steps=128, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted").cuda().eval()

            def call(m=m):
                out = m(rays)  # SDF MLP + View-half kernel (mlp) or the one-kernel model (siren)
                return out, m.alpha, m.weights
        elif tiny:
            m = nerf.TinyNeRF(steps=128, t_near=2.0, t_far=6.0, sigmoid_kind="upshifted").cuda().eval()
            call = lambda: ops.render_tiny_ls(rays, ts, m.packed_ls(prec), prec, "upshifted", "black", True)
        else:
            m = nerf.PlainNeRF(steps=128, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
            call = lambda: m._render_fused(rays, ts, True)  # (config.precision picks the kernel: f16x included)
        with torch.no_grad():
            outs = []
            for i in range(n):
                torch.empty(1 + (i * 7919) % 100000, device="cuda")  # perturb the allocator
                outs.append([t.clone() for t in call()])
        bad_runs = set()
        for j, name in enumerate(("colour", "alpha", "weights")):
            x = torch.stack([o[j] for o in outs]).reshape(n, -1)
            ref = x.median(0).values
            diff = x != ref
            nbad = int(diff.sum())
            if nbad:
                runs = diff.any(1).nonzero().flatten().tolist()
                bad_runs.update(runs)
                if name != "colour":
                    lanes = sorted(set(((diff.nonzero()[:, 1] // R) % 32).tolist()))
                    print(f"{prec} {name}: {nbad} elements differ in {len(runs)} runs; step-in-block {lanes}")
                else:
                    print(f"{prec} {name}: {nbad} elements differ in {len(runs)} runs")
        print(f"{prec}: {n} runs, {len(bad_runs)} irreproducible")
        total_bad += len(bad_runs)
    print(f"{total_bad} irreproducible runs")
    return 1 if total_bad else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        variants()
    else:
        sys.exit(main(sys.argv[1:]))
