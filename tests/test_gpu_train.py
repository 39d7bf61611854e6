"""Training parity with the reference itself (SURVEY 8(d) PSNR item (iii), 8(f) N1+N2).

tests/golden/train_parity_*.json hold what the REAL reference (runner.main on CPU, tools/ref_train_fixture.py) did on
the analytic Blender-format scene of tools/make_scene.py: per-iteration losses and test-set PSNRs.  Here the same
recipe runs through this repo's loaders, HIP forward/backward kernels and training loop on the GPU, replaying the
reference's random stream.  Bars (written here): first 10 losses within 2e-4 absolute (same trajectory, rounding only);
every test-view PSNR within 0.01 dB (exact-fp32 training GEMMs) / 0.1 dB (split-bf16 training GEMMs, the default) after
the full budget (plain D-NeRF recipe: chaotic trajectory, its end point is checked against the reference's OWN end-point ensemble,
tests/golden/train_spread.json, see below; `make dnerf`'s regularised recipe gets the strict bars), rendered by the
fused bf16x3 kernel; the fast bf16 renderer within 0.1 dB of that mean as well."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.procedural import proc_param  # noqa: E402  (checker-side weight generator)
from tools.make_scene import make_scene  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def procedural_init(model):
    with torch.no_grad():
        for name, t in model.state_dict().items():
            if name.endswith("primes") or t.numel() == 0 or name == "scale" or name.endswith(".scale"):
                continue
            if name.startswith("delta_estim.out."):
                continue  # the deformation head keeps the reference's zero initialisation (src/nerf.py:1256)
            v = torch.from_numpy(proc_param(name, tuple(t.shape)))
            if name.endswith("basis"):
                v = v * (16.0 if "sdf" in name else 32.0)
            t.copy_(v.to(t.dtype))


@pytest.mark.parametrize("name,train_prec", [("plain", "fp32"), ("plain", "bf16x3"), ("dnerf", "bf16x3"),
                                             ("volsdf", "bf16x3"), ("dnerf", "fp32"), ("volsdf", "fp32"),
                                             ("dnerf_make", "bf16x3"), ("dnerf_make", "fp32"),
                                             ("dnerf_make_rl3", "bf16x3"), ("dnerf_make_rl3", "fp32"),
                                             ("original", "bf16x3"), ("original", "fp32"),
                                             ("volsdf_mlp", "bf16x3"), ("volsdf_mlp", "fp32"),
                                             ("volsdf_smooth", "bf16x3"), ("volsdf_smooth", "fp32"),
                                             ("dnerf_div", "bf16x3"), ("dnerf_div", "fp32")])
def test_training_tracks_the_reference(name, train_prec, tmp_path):
    path = os.path.join(GOLDEN, f"train_parity_{name}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    fx = json.load(open(path))
    import nerf_atlas_amd.train as T
    from nerf_atlas_amd import config
    data = make_scene(str(tmp_path / "scene"), **fx["scene"]) + "/"
    argv = [x for x in fx["argv"] if x not in ("-d", "--outdir")]  # their values (temp paths) were dropped
    args = T.args_from_argv(["-d", data] + argv)
    assert args.epochs == len(fx["losses"])
    config.set_precision("bf16x3")
    prev = config.train_precision
    config.set_train_precision(train_prec)
    config.set_deterministic(True)  # fixed-point accumulation of the cross-workgroup sums: reproducible trajectories
    try:
        res = T.fit(args, replay_reference_rng=True, init=procedural_init)
    finally:
        config.set_train_precision(prev)
        config.set_deterministic(False)
    got, ref = np.array(res["losses"]), np.array(fx["losses"])
    d = np.abs(np.array(res["test_psnr"]) - np.array(fx["test_psnr"]))
    print(f"\n[{name}/{train_prec}] |loss - ref| first 10: {np.abs(got[:10] - ref[:10]).max():.2e}, first 5: "
          f"{np.abs(got[:5] - ref[:5]).max():.2e}; test PSNR build {np.round(res['test_psnr'], 4).tolist()} vs reference "
          f"{np.round(fx['test_psnr'], 4).tolist()} (max diff {d.max():.4f} dB)")
    # same trajectory while rounding noise has not been amplified yet.  D-NeRF: predicted positions cross hash-cell faces,
    # so the rounding-order difference between the reference's CPU kernels and these ones is amplified after ~5
    # iterations.  With config.set_deterministic the build's own trajectory is bit-reproducible (test below), so the
    # numbers are fixed per build: first-10 deviation 1.9e-4 (bf16x3) / 1.5e-4 (fp32), per-view PSNR 0.38 / 0.41 dB (below).
    assert np.abs(got[:5] - ref[:5]).max() <= 2e-4, (got[:5], ref[:5])
    # original = `make original` (makefile:8-13: --refl-kind pos -lr 2e-4 --loss-fns l2; round 6).
    # dnerf_make_rl3 = `make dnerf` AS SHIPPED (round 6: + --dyn-refl-latent 3, the deformation network's latent columns through the
    # spline into the PosLinearView head and back through na_bezier_warp_latent_backward).
    # dnerf_make = `make dnerf`'s regularisers (offset decay 60, the FFJORD estimate whose randn draw advances the RNG
    # stream, opt-step 3, pos-linear-view): not chaotic -- it tracks the reference to 2e-6 in the loss and 0.0007 dB, so it
    # gets the strict bars; only the plain `dnerf` recipe needs the chaotic ones
    dyn = name in ("dnerf", "dnerf_div")  # the plain D-NeRF recipe, with or without the divergence term, is the chaotic one
    assert np.abs(got[:10] - ref[:10]).max() <= (1e-3 if dyn else 2e-4), (got[:10], ref[:10])
    rel50 = float((np.abs(got[:50] - ref[:50]) / ref[:50]).max())
    print(f"[{name}/{train_prec}] first 50 iterations: max relative loss deviation {rel50:.3e}")
    # the whole curve stays on the reference's (smoothed: single iterations are noisy by design)
    k = 20
    sm = lambda v: np.convolve(v, np.ones(k) / k, mode="valid")
    dev = np.abs(sm(got) - sm(ref)).max() / sm(ref).max()
    print(f"[{name}/{train_prec}] smoothed-curve deviation {dev:.4f} of the curve's maximum")
    assert dev <= (0.35 if dyn else 0.1), dev
    # (dnerf_make steps the optimiser every third iteration: 67 updates in the 200 iterations)
    # (volsdf_mlp drops from 0.17 to 0.03 within its first 20 iterations: its start is the first three losses)
    start = ref[:3].mean() if name == "volsdf_mlp" else ref[:k].mean()
    assert ref[-k:].mean() < (0.7 if name.startswith("dnerf_make") else 0.5) * start, "the recipe must actually learn"
    if dyn:
        # Chaotic recipe.  What is NOT chaotic is pinned strictly: the first losses above, and -- for the divergence recipe -- the
        # regulariser's own effect on the trajectory: over the first 10 iterations the reference's `dnerf_div` losses differ from
        # its plain `dnerf` losses by up to 6e-4 (same seed, same stream), and the build must reproduce the regularised trajectory
        # to a small fraction of that effect (measured: 2 % with fp32 GEMMs, 8 % with split bf16).
        if name == "dnerf_div":
            plain = np.array(json.load(open(os.path.join(GOLDEN, "train_parity_dnerf.json")))["losses"][:10])
            effect = np.abs(ref[:10] - plain).max()
            frac = np.abs(got[:10] - ref[:10]).max() / effect
            print(f"[{name}/{train_prec}] regulariser effect on the first 10 losses {effect:.2e}; build deviation = {frac:.3f} of it")
            assert effect >= 3e-4 and frac <= (0.05 if train_prec == "fp32" else 0.15), (effect, frac)
            # ... and the regulariser ITSELF over the whole budget.  `divergence(pts, dp).mean()` enters the loss unsquared
            # (runner.py:694-696, with the reference's own "TODO maybe this is wrong?"): it is unbounded below and training drives it
            # from 0 to -4.6e6 (x 0.05 against an L2 loss of 0.02).  The reference's trace of the term (instrumented run of
            # tools/ref_train_fixture.py, same recipe / seed / stream) and the build's must agree through those eight orders of
            # magnitude: every one of the first 10 iterations and every 20-iteration window mean (measured against the nearer of the
            # reference's two traces: <= 1.8 % / 1.2 % with split bf16, 0.2 % / 0.9 % with fp32 GEMMs).  THIS is the parity statement of the recipe.
            sp_reg = json.load(open(os.path.join(GOLDEN, "train_spread.json")))[name]["reference_reg"]
            refs = [np.array(r["reg_terms"]) for r in sp_reg]
            br = np.array(res["reg_terms"])
            assert len(refs) >= 2 and all(len(r) == len(br) == len(ref) and r[-1] < -1e6 for r in refs), [len(r) for r in refs]
            win = lambda x, r: np.array([abs(x[a:a + 20].mean() / r[a:a + 20].mean() - 1) for a in range(0, len(r), 20)])
            # the reference's OWN traces (two thread counts) drift apart by up to 3.3 % per window once the trajectories decorrelate
            # (0.2 % over the first 10 iterations): the build must stay within 3 % of the NEARER one and within 3 % + that spread
            # of every one
            spread = max(win(refs[i], refs[j]).max() for i in range(len(refs)) for j in range(i))
            e10 = min(np.abs(br[1:10] / r[1:10] - 1).max() for r in refs)
            wins = np.array([win(br, r) for r in refs])
            ew_near, ew_far = wins.min(axis=0).max(), wins.max()
            print(f"[{name}/{train_prec}] divergence term: reference {refs[0][-1]:.4e} / build {br[-1]:.4e} at the end; deviation first 10 "
                  f"iterations {e10:.4f}, worst 20-iteration window {ew_near:.4f} (nearer reference trace) / {ew_far:.4f} (any); the "
                  f"reference's own traces differ by {spread:.4f}")
            assert e10 <= (0.01 if train_prec == "fp32" else 0.04) and ew_near <= 0.03 and ew_far <= 0.03 + spread, (e10, ew_near, ew_far, spread)
        # The END POINT is a distribution, on both sides.  tests/golden/train_spread.json holds the reference's own end points
        # under a last-bit perturbation (same recipe, seed and random stream at several thread counts = another summation order
        # in its CPU kernels) next to this build's (fp32-atomic accumulation; tools/train_spread.py).
        #   dnerf: the bars are DERIVED from the reference's spread: 3 x the range its own runs span, per view and on the mean
        #     (the range of n = 3 .. 7 runs is 1.7 .. 2.7 sigma: 3 x range is a generous 5 .. 8 sigma), capped by 1.3 / 0.9 dB.
        #   dnerf_div: the L2 part of that loss rides on the rounding noise of the unbounded term above (its gradient is 1e5 x the L2
        #     gradient in the sums Adam normalises), so the end point measures ACCUMULATION PRECISION, not parity: the reference's
        #     own runs span 0.9 dB per view and 0.46 dB on the mean; the build's ensemble sits 0.2 - 0.4 dB ABOVE the reference's, its deterministic runs
        #     (64-bit fixed-point gradient sums, which keep the L2 part of a sum that fp32 rounds away) highest, and one of ten
        #     fp32-atomic runs fell out of the basin (15.99 dB).  Bars: the caps alone (1.3 dB per view = 1.5 x the reference's own
        #     range; 0.9 dB on the mean = 2 x), as a "same basin" statement; the derived bars are printed next to them.
        sp = json.load(open(os.path.join(GOLDEN, "train_spread.json")))[name]
        ref_runs = np.array([r["test_psnr"] + [r["test_psnr_mean"]] for r in sp["reference_runs"]])
        assert len(ref_runs) >= 3, len(ref_runs)
        rng_ = ref_runs.max(axis=0) - ref_runs.min(axis=0)
        bar_view, bar_mean = min(1.3, 3.0 * rng_[:-1].max()), min(0.9, 3.0 * rng_[-1])
        dev_view = np.abs(np.array(res["test_psnr"]) - ref_runs[:, :-1].mean(axis=0)).max()
        dev_mean = abs(res["test_psnr_mean"] - ref_runs[:, -1].mean())
        print(f"[{name}/{train_prec}] end point vs the reference's own ensemble (n = {len(ref_runs)}): range per view "
              f"{np.round(rng_[:-1], 3).tolist()}, of the mean {rng_[-1]:.3f} -> derived bars {bar_view:.3f} / {bar_mean:.3f} dB; this run "
              f"{dev_view:.3f} / {dev_mean:.3f} dB from the ensemble mean")
        if name == "dnerf_div":
            bar_view, bar_mean = 1.3, 0.9
        assert dev_view <= bar_view and dev_mean <= bar_mean, (res["test_psnr"], ref_runs.tolist(), bar_view, bar_mean)
    elif name == "volsdf_smooth":
        # eikonal + normal smoothing (the reference's VolSDF regularisers, makefile:85-95): the smoothing term is a difference
        # of normals <= 1e-3 apart, which amplifies rounding-order differences over the 200 iterations; its tangent sweeps
        # run in exact fp32 in both modes (train.py).  Measured: loss within 2.7e-5 over the first 10 iterations, per-view
        # PSNR within 0.10 dB of the reference's run (north_star's bar); 1.5x for another toolchain
        assert d.max() <= 0.15, (res["test_psnr"], fx["test_psnr"])
    else:
        assert d.max() <= (0.01 if train_prec == "fp32" else 0.1), (res["test_psnr"], fx["test_psnr"])
    # the headline parity mode on TRAINED weights (VERDICT r03 "weak" 1: every f16x assertion of the suite used procedural or
    # random-init weights): the fused f16x renderers on the model this run just trained -- every test view within 0.01 dB of the
    # bf16x3 render, and view 0 within 1e-4 L-inf of the CPU oracle evaluated on the trained state_dict
    if train_prec == "bf16x3" and name in ("plain", "volsdf", "dnerf", "original", "dnerf_make_rl3"):
        import oracle as O
        model = res["model"]
        cam, labels = _test_set(T, args)
        config.set_precision("f16x")
        try:
            px, frames = T.test(model, cam, labels, args)
        finally:
            config.set_precision("bf16x3")
        dpx = np.abs(np.array(px) - np.array(res["test_psnr"]))
        params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        size = args.render_size
        rays = cam[0:1].sample_positions((0, 0, size, size), size=size, with_noise=False).cpu()
        if name == "plain":
            ref = O.plain_nerf(params, rays, args.near, args.far, args.steps, "view", act=args.sigmoid_kind)
        elif name == "volsdf":
            ref = O.volsdf(params, rays, args.near, args.far, args.steps, sdf_kind="siren", act=args.sigmoid_kind)
        elif name == "original":   # round 6: PlainNeRF + Positional through the one-launch renderer (MODEL 7)
            ref = O.plain_nerf(params, rays, args.near, args.far, args.steps, "pos", act=args.sigmoid_kind)
        elif name == "dnerf_make_rl3":   # D-NeRF with three refl_latent columns over PlainNeRF + PosLinearView (MODEL 8)
            ref = O.dynamic_nerf_spline(params, rays, labels[-1][0:1], args.near, args.far, args.steps, 6, "pos-linear-view",
                                        act=args.sigmoid_kind, refl_latent=3)
        else:
            ref = O.dynamic_nerf_spline(params, rays, labels[-1][0:1], args.near, args.far, args.steps, 4, act=args.sigmoid_kind)
        err = float((frames[0].cpu() - ref[0]).abs().max())
        print(f"[{name}] trained weights in f16x: PSNR vs the bf16x3 render {np.round(dpx, 5).tolist()} dB, view 0 L-inf vs the "
              f"CPU oracle {err:.2e}")
        assert torch.isfinite(frames[0]).all() and err <= 1e-4, err
        assert dpx.max() <= 0.01, (px, res["test_psnr"])
        if name == "dnerf":  # the deformation network on the layer-synchronous engine (opt-in, config.deformation_engine)
            config.set_precision("f16x")
            config.set_deformation_engine("ls")
            try:
                px2, frames2 = T.test(model, cam, labels, args)
            finally:
                config.set_precision("bf16x3")
                config.set_deformation_engine("ls-bf16x3")
            err2 = float((frames2[0].cpu() - ref[0]).abs().max())
            print(f"[dnerf] f16x with the LS deformation kernel: view 0 L-inf vs the CPU oracle {err2:.2e}")
            assert err2 <= 1e-4 and np.abs(np.array(px2) - np.array(res["test_psnr"])).max() <= 0.01
    # the fast renderer on the trained model
    if name == "plain":
        config.set_precision("bf16")
        try:
            fast, _ = T.test(res["model"], *_test_set(T, args), args)
        finally:
            config.set_precision("bf16x3")
        assert abs(np.mean(fast) - fx["test_psnr_mean"]) <= 0.1, (fast, fx["test_psnr"])


@pytest.mark.parametrize("name", ["dnerf", "volsdf"])
def test_deterministic_training_is_bitwise_reproducible(name, tmp_path):
    """config.set_deterministic: the gradients that are summed across workgroups (dW/db, hash-table scatter, d/dbeta)
    accumulate in 64-bit fixed point, so two runs of the chaotic D-NeRF recipe produce the same losses and the same
    parameters bit for bit (with fp32 atomics they decorrelate after ~5 iterations)."""
    fx = json.load(open(os.path.join(GOLDEN, f"train_parity_{name}.json")))
    import nerf_atlas_amd.train as T
    from nerf_atlas_amd import config
    data = make_scene(str(tmp_path / "scene"), **fx["scene"]) + "/"
    argv = [x for x in fx["argv"] if x not in ("-d", "--outdir")]
    args = T.args_from_argv(["-d", data] + argv)
    args.epochs = 40
    runs = []
    config.set_deterministic(True)
    try:
        for _ in range(2):
            res = T.fit(args, replay_reference_rng=True, init=procedural_init)
            runs.append((res["losses"], {k: v.detach().clone() for k, v in res["model"].state_dict().items()}))
    finally:
        config.set_deterministic(False)
    assert runs[0][0] == runs[1][0], np.abs(np.array(runs[0][0]) - np.array(runs[1][0])).max()
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k


def _test_set(T, args):
    labels, cam, _ = T.loaders.load(args, training=False)
    return cam.cuda(), labels


def test_unsupported_regularisers_raise_and_eikonal_trains(tmp_path):
    import nerf_atlas_amd.train as T
    data = make_scene(str(tmp_path / "s"), size=16, n_train=2, n_test=1) + "/"
    args = T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=1, dyn_diverge_decay=0.1)
    with pytest.raises(ValueError):  # the deformation's divergence needs a dynamic model
        T.fit(args)
    with pytest.raises(ValueError):  # the FFJORD estimate reads the deformation field of a dynamic model
        T.fit(T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=1, ffjord_div_decay=0.1))
    with pytest.raises(ValueError):  # the eikonal term needs an SDF model
        T.fit(T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=1, sdf_eikonal=0.1))
    with pytest.raises(ValueError):  # so does the normal smoothing
        T.fit(T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=1, smooth_normals=0.1))
    with pytest.raises(NotImplementedError):  # its double-backward form (eps = 0) is not implemented; the default form is
        T.fit(T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=1, model="volsdf", sdf_kind="siren",
                          smooth_normals=0.1, smooth_eps=0.0))
    # `make dtu`-style recipe: VolSDF + --sdf-eikonal runs and lowers E[(|n|-1)^2] of the SDF
    args = T.make_args(data=data, size=16, crop_size=8, batch_size=1, steps=8, epochs=25, model="volsdf", sdf_kind="siren",
                       near=0.3, far=1.8, sdf_eikonal=0.5, learning_rate=2e-4)
    import nerf_atlas_amd.autograd as ag
    torch.manual_seed(0)
    res = T.fit(args)
    m = res["model"]
    pts = 5 * torch.randn(4096, 3, device="cuda")
    after = float(ag.EikonalFn.apply(m.sdf.underlying.normals_tangent_major(pts)).detach())
    torch.manual_seed(0)
    fresh = T.load_model(args)
    before = float(ag.EikonalFn.apply(fresh.sdf.underlying.normals_tangent_major(pts)).detach())
    assert after < before, (before, after)


def test_runner_cli_writes_results(tmp_path):
    """python -m nerf_atlas_amd.runner with the reference's flags: trains a few iterations, renders the test set with
    the fused kernels, writes results.txt in the reference's format and the state_dict under the reference's keys."""
    from nerf_atlas_amd import runner
    data = make_scene(str(tmp_path / "s"), size=32, n_train=4, n_test=2) + "/"
    out = tmp_path / "out"
    res = runner.main(["-d", data, "--size", "32", "--crop-size", "16", "--test-crop-size", "32", "--batch-size", "2",
                       "--steps", "24", "--epochs", "12", "--quiet", "--model", "plain", "--refl-kind", "view",
                       "--outdir", str(out), "--save", str(tmp_path / "m.pt")])
    txt = (out / "results.txt").read_text()
    assert "[Summary" in txt and "mean" in txt and txt.count("PSNR") == 2
    assert (out / "test_000.png").exists() and len(res["losses"]) == 12
    sd = torch.load(tmp_path / "m.pt")
    assert "first.init.weight" in sd and "refl.mlp.out.bias" in sd and "first.enc.embs.7.weight" in sd


def test_f16x_fourier_sdf_schedule_on_self_trained_weights(tmp_path):
    """Round 4's MODEL 5 + 2 (VolSDF with the Fourier-MLP SDF network as one f16x launch + the View half) on TRAINED weights:
    there is no reference run for this recipe, so the model is trained here for 80 iterations by this repo's own training step
    and the f16x render of the trained state is checked like the reference-trained ones above: view 0 within 1e-4 L-inf of the
    CPU oracle on the trained state_dict, every test view within 0.01 dB of the bf16x3 render.  (MODEL 6, mip, has no training
    recipe on either side: the reference's composed IPE path is NaN at HEAD and this repo's mip path is inference only.)"""
    import oracle as O
    import nerf_atlas_amd.train as T
    from nerf_atlas_amd import config
    data = make_scene(str(tmp_path / "scene"), size=48, n_train=12, n_test=3, dynamic=False) + "/"
    args = T.args_from_argv(["-d", data, "--size", "48", "--crop-size", "24", "--test-crop-size", "48", "--batch-size", "2", "--steps",
                             "48", "--epochs", "80", "--seed", "1337", "--nosave", "--quiet", "--notraintest", "--valid-freq", "1000000",
                             "--model", "volsdf", "--sdf-kind", "mlp", "--refl-kind", "view", "--near", "2", "--far", "6"])
    config.set_precision("bf16x3")
    res = T.fit(args, init=procedural_init)
    losses = np.array(res["losses"])
    assert losses[-10:].mean() < 0.6 * losses[:10].mean(), "the recipe must actually learn"
    model = res["model"]
    cam, labels = _test_set(T, args)
    config.set_precision("f16x")
    try:
        assert model._fusable_fourier_sdf()
        px, frames = T.test(model, cam, labels, args)
    finally:
        config.set_precision("bf16x3")
    dpx = np.abs(np.array(px) - np.array(res["test_psnr"]))
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    size = args.render_size
    rays = cam[0:1].sample_positions((0, 0, size, size), size=size, with_noise=False).cpu()
    ref = O.volsdf(params, rays, args.near, args.far, args.steps, sdf_kind="mlp", act=args.sigmoid_kind)
    err = float((frames[0].cpu() - ref[0]).abs().max())
    print(f"[volsdf_mlp] self-trained weights in f16x: PSNR vs the bf16x3 render {np.round(dpx, 5).tolist()} dB, view 0 L-inf vs the CPU "
          f"oracle {err:.2e}; test PSNR {np.round(px, 2).tolist()}")
    assert torch.isfinite(frames[0]).all() and err <= 1e-4, err
    assert dpx.max() <= 0.01, (px, res["test_psnr"])


def test_one_launch_adam_is_torch_foreach_adam_bit_for_bit():
    """train.NaAdam (na_adam_step: ONE launch per step) against torch.optim.Adam's default foreach implementation: the same
    parameters, moments and step counts after every one of 12 steps, bit for bit -- tensors of 1 .. 262 144 elements (ragged
    tails, a 1-element bias), gradients over eight orders of magnitude, a changing learning rate, a parameter that only gets
    its gradient from step 4 on (its own step count), state_dict interchange.  Prints which contraction masks reproduce torch."""
    import nerf_atlas_amd.train as T
    torch.manual_seed(11)
    shapes = [(256, 256), (65,), (1,), (8, 65536, 4)[1:], (256, 294), (3, 256), (19,), (1023,), (1025,)]

    def make():
        torch.manual_seed(12)
        return [torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.1) for s in shapes]

    def grads(step):
        g = torch.Generator(device="cuda").manual_seed(100 + step)
        return [torch.randn(*s, device="cuda", generator=g) * (10.0 ** ((i % 9) - 6)) for i, s in enumerate(shapes)]

    def run(opt_cls, ps, **kw):
        opt = opt_cls(ps, lr=5e-4, eps=1e-7, **kw)
        for step in range(12):
            for i, (p, g) in enumerate(zip(ps, grads(step))):
                p.grad = None if (i == 1 and step < 4) else g
            if step == 6:
                opt.param_groups[0]["lr"] = 2e-4
            opt.step()
        return opt

    ref_p = make()
    ref = run(torch.optim.Adam, ref_p, foreach=True)
    matching = []
    for mask in range(8):
        T.NaAdam.FMA_MASK, keep = mask, T.NaAdam.FMA_MASK
        try:
            ps = make()
            opt = run(T.NaAdam, ps)
        finally:
            T.NaAdam.FMA_MASK = keep
        same = all(torch.equal(a, b) for a, b in zip(ps, ref_p)) and all(
            torch.equal(opt.state[a][k], ref.state[b][k]) for a, b in zip(ps, ref_p) for k in ("exp_avg", "exp_avg_sq"))
        if same:
            matching.append(mask)
    print(f"\n[adam] contraction masks that reproduce torch's foreach Adam bit for bit: {matching}; pinned: {T.NaAdam.FMA_MASK}")
    assert T.NaAdam.FMA_MASK in matching, matching
    ps = make()
    opt = run(T.NaAdam, ps)
    assert all(float(opt.state[a]["step"]) == float(ref.state[b]["step"]) for a, b in zip(ps, ref_p))
    # state_dicts interchange
    ref2 = torch.optim.Adam(make(), lr=5e-4, eps=1e-7)
    ref2.load_state_dict(opt.state_dict())
    opt2 = T.NaAdam(make(), lr=5e-4, eps=1e-7)
    opt2.load_state_dict(ref.state_dict())
    # weight decay: torch's own step for that group (same class, same results as torch)
    pa, pb = make(), make()
    oa, ob = T.NaAdam(pa, lr=5e-4, eps=1e-7, weight_decay=0.01), torch.optim.Adam(pb, lr=5e-4, eps=1e-7, weight_decay=0.01)
    for p, q, g in zip(pa, pb, grads(0)):
        p.grad, q.grad = g, g.clone()
    oa.step(); ob.step()
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
