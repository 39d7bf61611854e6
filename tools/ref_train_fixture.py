#!/usr/bin/env python3
"""Train the REAL reference (/root/reference runner.main, CPU, in-process shims of SURVEY.md App. A) on the analytic
scene of tools/make_scene.py and record what the build's training step must reproduce (SURVEY 8(d) PSNR item (iii)):

    tests/golden/train_parity_<name>.json = { recipe (argv, seeds), per-iteration l2 losses, test-set PSNRs }

Determinism contract shared with nerf_atlas_amd/train.py:
  * the dataset is regenerated bit-identically by make_scene on either side;
  * initial parameters are procedural (oracle/procedural.proc_param keyed by state_dict name), written into the
    reference model right after runner.load_model;
  * `random.seed(seed)` (crops, view indices) and, after the parameters are written, `torch.manual_seed(seed + 1)`;
    every stochastic tensor of an iteration (pixel jitter u then v, stratified `rand[T]`, density noise) is then a
    draw from torch's CPU generator in the reference's order, which the build replays.

Runs only in the build container; nothing of the reference is copied or travels.

    python tools/ref_train_fixture.py plain [--epochs 300]
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.procedural import proc_param  # noqa: E402
from tools.make_scene import make_scene  # noqa: E402

REF = "/root/reference"

RECIPES = {
    # name: (dynamic scene?, reference argv after the data flags)
    "plain": (False, ["--model", "plain", "--refl-kind", "view"]),
    # `make original` (reference makefile:8-13: --model plain --refl-kind pos -lr 2e-4 --loss-fns l2) on the small scene
    "original": (False, ["--model", "plain", "--refl-kind", "pos", "--loss-fns", "l2", "-lr", "2e-4"]),
    "dnerf": (True, ["--model", "plain", "--refl-kind", "view", "--data-kind", "dnerf", "--dyn-model", "plain",
                     "--spline", "4"]),
    # `make dnerf`'s regularisers (reference makefile:106-114) on the small scene: NR-NeRF offset decay + the FFJORD
    # divergence estimate (whose randn_like draw advances the RNG stream every iteration), opt-step 3, upshifted sigmoid
    "dnerf_make": (True, ["--model", "plain", "--refl-kind", "pos-linear-view", "--data-kind", "dnerf", "--dyn-model",
                          "plain", "--spline", "6", "--higher-end-chance", "1", "--offset-decay", "60",
                          "--ffjord-div-decay", "0.5", "--sigmoid-kind", "upshifted", "--opt-step", "3"]),
    # `make dnerf` AS SHIPPED (makefile:106-114): the same + `--dyn-refl-latent 3` -- the deformation network's extra columns ride
    # through the spline into the PosLinearView head (round 6: src/nerf.py:1245-1248, 1272-1278, 1303)
    "dnerf_make_rl3": (True, ["--model", "plain", "--refl-kind", "pos-linear-view", "--data-kind", "dnerf", "--dyn-model",
                              "plain", "--spline", "6", "--higher-end-chance", "1", "--offset-decay", "60",
                              "--ffjord-div-decay", "0.5", "--sigmoid-kind", "upshifted", "--opt-step", "3",
                              "--dyn-refl-latent", "3"]),
    # the "exact divergence" regulariser of the deformation field (runner.py:694-696): autograd of model.dp w.r.t. model.pts with
    # create_graph -- a double backward through the hash encoder, the deformation MLP and the spline in the reference
    "dnerf_div": (True, ["--model", "plain", "--refl-kind", "view", "--data-kind", "dnerf", "--dyn-model", "plain",
                         "--spline", "4", "--dyn-diverge-decay", "0.05"]),
    "volsdf": (False, ["--model", "volsdf", "--sdf-kind", "siren", "--refl-kind", "view", "--near", "2", "--far", "6"]),
    # BASELINE config 5's other SDF network (`make volsdf`'s --sdf-kind mlp: Fourier-encoded SkipConnMLP, src/sdf.py:250-258) with
    # the eikonal term of the VolSDF recipes and the upshifted sigmoid (makefile:21-28, 127-133)
    # (`make dnerf_volsdf`, makefile:127-133 -- D-NeRF over a VolSDF canonical model with --sdf-eikonal -- cannot be recorded: the reference
    # raises at runner.py:807, `pts + dp` with dp the (dp, enc) tuple of DynamicNeRF.time_estim.)
    "volsdf_mlp": (False, ["--model", "volsdf", "--sdf-kind", "mlp", "--refl-kind", "view", "--near", "2", "--far", "6",
                           "--sdf-eikonal", "1e-5", "--sigmoid-kind", "upshifted", "-lr", "3e-4"]),
    # the SDF regularisers of the reference's VolSDF recipes (makefile:85-95): eikonal + normal smoothing by the unisurf
    # epsilon perturbation with a random radius (one random.random() and two randn draws per iteration in the RNG streams)
    "volsdf_smooth": (False, ["--model", "volsdf", "--sdf-kind", "siren", "--refl-kind", "view", "--near", "2", "--far", "6",
                              "--sdf-eikonal", "1e-2", "--smooth-normals", "1e-2", "--smooth-eps-rng"]),
}


def shims():
    def stub(name, subs=()):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        for s in subs:
            sm = types.ModuleType(f"{name}.{s}")
            sm.__path__ = []
            sys.modules[f"{name}.{s}"] = sm
            setattr(m, s, sm)
    stub("torchvision", ["models", "transforms", "io"])
    tf = types.ModuleType("torchvision.transforms.functional")
    sys.modules["torchvision.transforms.functional"] = tf
    sys.modules["torchvision.transforms"].functional = tf
    stub("imageio")
    sys.path.insert(0, REF)
    nn.Module.cuda = lambda self, *a, **k: self


def fill_procedural(module):
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if name.endswith("primes") or t.numel() == 0 or name == "scale" or name.endswith(".scale"):
                continue
            if name.startswith("delta_estim.out."):
                continue  # the deformation head keeps the reference's zero initialisation (src/nerf.py:1256)
            v = torch.from_numpy(proc_param(name, tuple(t.shape)))
            if name.endswith("basis"):
                v = v * (16.0 if "sdf" in name else 32.0)
            t.copy_(v.to(t.dtype))


def main():
    a = argparse.ArgumentParser()
    a.add_argument("name", choices=sorted(RECIPES))
    a.add_argument("--epochs", type=int, default=300)
    a.add_argument("--size", type=int, default=48)
    a.add_argument("--crop-size", type=int, default=24)
    a.add_argument("--batch-size", type=int, default=2)
    a.add_argument("--steps", type=int, default=48)
    a.add_argument("--seed", type=int, default=1337)
    a.add_argument("--threads", type=int, default=8)
    a.add_argument("--out", default=None, help="write the fixture here instead of tests/golden/train_parity_<name>.json "
                   "(sensitivity runs: the same recipe under another thread count = another summation order)")
    cfg = a.parse_args()
    torch.set_num_threads(cfg.threads)
    dynamic, model_argv = RECIPES[cfg.name]
    scene = dict(size=cfg.size, n_train=12, n_test=3, dynamic=dynamic)

    shims()
    import src.nerf as rnerf
    import src.utils as rutils
    rnerf.with_transmission = False
    rutils.git_hash = lambda: "nogit"
    import runner
    runner.git_hash = lambda: "nogit"
    runner.device = "cpu"
    runner.save_plot = lambda *a, **k: None

    captured = {"losses": None, "psnr": []}
    real_psnr = rutils.mse2psnr

    def mse2psnr(x):
        v = real_psnr(x)
        captured["psnr"].append(float(v))
        return v
    rutils.mse2psnr = mse2psnr
    real_div = rutils.divergence

    def divergence(x, field):  # (the --dyn-diverge-decay term per iteration: a diagnostic recorded next to the losses)
        v = real_div(x, field)
        captured.setdefault("reg_terms", []).append(float(v.mean().detach()))
        return v
    rutils.divergence = divergence
    runner.save_losses = lambda args, losses: captured.__setitem__("losses", list(losses))
    real_load_model = runner.load_model

    def load_model(args, light, is_dyn=False):
        m = real_load_model(args, light, is_dyn)
        fill_procedural(m)
        torch.manual_seed(cfg.seed + 1)
        return m
    runner.load_model = load_model

    with tempfile.TemporaryDirectory() as td:
        data = make_scene(os.path.join(td, "scene"), **scene) + "/"
        out = os.path.join(td, "out")
        argv = ["-d", data, "--size", str(cfg.size), "--crop-size", str(cfg.crop_size), "--test-crop-size",
                str(cfg.size), "--batch-size", str(cfg.batch_size), "--steps", str(cfg.steps), "--epochs",
                str(cfg.epochs), "--seed", str(cfg.seed), "--nosave", "--quiet", "--notraintest", "--valid-freq",
                "1000000", "--outdir", out] + model_argv
        sys.argv = ["runner.py"] + argv
        t0 = time.time()
        runner.main()
        dt = time.time() - t0
        results = open(os.path.join(out, "results.txt")).read()
    printed = [float(line.split("PSNR")[1]) for line in results.splitlines() if "PSNR" in line]
    psnrs = captured["psnr"][-len(printed):]  # full precision of what results.txt rounds to 3 decimals
    assert all(abs(a - b) < 6e-4 for a, b in zip(psnrs, printed)), (psnrs, printed)
    mean = float(np.mean(psnrs))
    fixture = dict(
        what="reference runner.main() on the analytic scene; see tools/ref_train_fixture.py",
        name=cfg.name, scene=scene, seed=cfg.seed,
        argv=[x for x in argv if x not in (data, out, "-d", "--outdir")],
        recipe=dict(size=cfg.size, crop_size=cfg.crop_size, batch_size=cfg.batch_size, steps=cfg.steps,
                    epochs=cfg.epochs, model_argv=model_argv, lr=5e-4, sched_min=5e-5, adam_eps=1e-7, near=2.0, far=6.0),
        losses=[float(x) for x in captured["losses"]],
        test_psnr=psnrs, test_psnr_mean=mean, reg_terms=captured.get("reg_terms", []),
        torch=torch.__version__, threads=cfg.threads, wall_s=round(dt, 1),
    )
    path = cfg.out or os.path.join(REPO, "tests", "golden", f"train_parity_{cfg.name}.json")
    with open(path, "w") as f:
        json.dump(fixture, f, indent=1)
    print(f"{path}: {len(fixture['losses'])} iterations in {dt:.0f} s, loss {fixture['losses'][0]:.4f} -> "
          f"{np.mean(fixture['losses'][-20:]):.4f}, test PSNR {psnrs} mean {mean}")


if __name__ == "__main__":
    main()
