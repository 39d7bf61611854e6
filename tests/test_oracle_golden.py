"""Pin the CPU oracle (oracle/) against fixtures generated from the REAL reference
(tools/gen_golden.py).  CPU-only; runs under -m "not gpu"."""

import numpy as np
import pytest
import torch

import oracle as O
from conftest import load_golden, golden_params

EXACT = dict(rtol=0, atol=0)


def close(a, b, atol=1e-6, rtol=1e-6):
    a = torch.as_tensor(a); b = torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), float((a - b).abs().max())


def test_g1_pixels_and_rays_bit_exact():
    g = load_golden("g1_nerf_camera")
    size = int(g["size"])
    for i, crop in enumerate(g["crops"].tolist()):
        pos = O.pixel_grid(size, tuple(crop))
        assert torch.equal(pos, g[f"pos{i}"])
        rays = O.nerf_camera_rays(pos, g["c2w"], float(g["focal"]), size)
        assert torch.equal(rays, g[f"rays{i}"])
    pos = O.pixel_grid(size, tuple(g["crops"].tolist()[1]))
    rays = O.nerf_camera_rays(pos, g["c2w"], float(g["focal"]), size, noise=g["noise"], with_noise=0.1)
    assert torch.equal(rays, g["rays_noise"])
    # KAT from SURVEY 8(c)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    r = O.nerf_camera_rays(O.pixel_grid(16), c2w, 100.0, 16)
    close(r[0, 0, 0], torch.tensor([0, 0, 4, -0.08, 0.08, -1.0]), 1e-7)
    close(r[0, 15, 15], torch.tensor([0, 0, 4, 0.07, -0.07, -1.0]), 1e-7)


def test_g1_dtu_camera():
    g = load_golden("g1_dtu_camera")
    rays = O.dtu_camera_rays(g["pos"], g["pose"], g["intrinsic"], int(g["size"]))
    assert torch.equal(rays, g["rays"])


def test_g2_sampling():
    g = load_golden("g2_sampling")
    for tag in ("lin", "disp", "lin128"):
        near, far, T, lind = g[f"cfg_{tag}"].tolist()
        ts, _ = O.compute_ts(near, far, int(T), bool(lind))
        assert torch.equal(ts, g[f"ts_{tag}"])
    ts, mids = O.compute_ts(2.0, 6.0, 16, perturb=1.0, rand=g["rand"])
    assert torch.equal(ts, g["ts_perturb"]) and torch.equal(mids, g["mids"])
    ts, _ = O.compute_ts(2.0, 6.0, 16)
    pts = O.compute_pts(g["rays"][..., :3], g["rays"][..., 3:], ts)
    assert torch.equal(pts, g["pts"])


def test_g3_composite():
    g = load_golden("g3_composite")
    for tag, (t, sp) in {"softplus": ("ts", True), "relu": ("ts", False), "zero": ("ts_zero", True)}.items():
        a, w = O.alpha_from_density(g["density"], g[t], g["r_d"], softplus=sp)
        assert torch.equal(a, g[f"alpha_{tag}"]) and torch.equal(w, g[f"weights_{tag}"])
        out = O.volumetric_integrate(w, g["rgb"])
        assert torch.equal(out, g[f"out_{tag}"])
        assert torch.equal(out + O.sky_white(w), g[f"white_{tag}"])
    a, w = O.alpha_from_density(torch.tensor([0.5, 1, 2, -1.0]).reshape(4, 1, 1, 1), torch.linspace(2, 6, 4),
                                torch.tensor([0, 0, -1.0]).reshape(1, 1, 1, 3))
    close(a.reshape(-1), torch.tensor([0.46852690, 0.60314965, 0.82640243, 1.0]), 1e-7)
    close(w.reshape(-1), torch.tensor([0.46852690, 0.32055780, 0.17430089, 0.03661438]), 1e-7)


def test_g4_hash_indices_bit_exact_and_features():
    g = load_golden("g4_hash")
    assert np.allclose(np.array(O.hash_resolutions()), g["resolutions"].numpy(), rtol=0, atol=0)
    idx = O.hash_corner_indices(g["x"])
    assert torch.equal(idx, g["idx"])
    # KAT from SURVEY 8(c): x=(0.3,-1.7,2.2)
    assert idx[0, 0, 0].item() == 18943 and idx[0, 7, 0].item() == 31396
    assert idx[7, 0, 0].item() == 47093 and idx[7, 7, 0].item() == 62770
    p = golden_params(g)
    feats = O.hash_encode(g["x"], [p[f"embs.{i}.weight"] for i in range(8)])
    assert torch.equal(feats, g["feats"])


def test_g5_fourier_positional():
    from oracle.procedural import proc_param
    g = load_golden("g5_fourier")
    for sigma in (16, 32):
        basis = torch.from_numpy(proc_param("basis", (3, 128))) * sigma
        out = O.fourier_encode(g[f"x_{sigma}"], basis)
        assert torch.equal(out, g[f"out_{sigma}"])
        # feature-level fp32 noise vs fp64 is large for Fourier features: documented bound 2e-4
        assert (out.double() - g[f"out64_{sigma}"]).abs().max() < 2e-4
    assert torch.equal(O.positional_encode(g["pe_x"], g["pe_bands"]), g["pe_out"])


MLP_CASES = ["tiny", "first", "view", "posrefl", "delta6", "sdfmlp", "siren", "mipfirst", "plv_view", "plv_pos"]


@pytest.mark.parametrize("case", MLP_CASES)
def test_g6_skip_mlp(case):
    g = load_golden(f"g6_mlp_{case}")
    p = golden_params(g, sigma=16.0)
    enc = None
    if str(g["enc"]) == "hash":
        enc = lambda x: O.hash_encode(x, [p[f"enc.embs.{i}.weight"] for i in range(8)])
    elif str(g["enc"]) == "fourier16":
        enc = lambda x: O.fourier_encode(x, p["enc.basis"])
    inter = []
    y = O.skip_mlp(p, "", g["p"], g.get("latent"), act=str(g["act"]), enc=enc, collect=inter)
    close(y, g["y"], 2e-6, 1e-5)
    for i, t in enumerate(inter):
        close(t, g[f"inter{i}"], 2e-6, 1e-5)
    shapes = O.mlp_linear_shapes(p["init.weight"].shape[1], int(g["layers"]), p["init.weight"].shape[0], int(g["out"]))
    assert shapes[0] == tuple(reversed(p["init.weight"].shape))
    for i in range(int(g["layers"])):
        assert shapes[1 + i] == tuple(reversed(p[f"layers.{i}.weight"].shape))


def test_g7_elaz_sigmoids_heads():
    g = load_golden("g7_elaz_sigmoid")
    assert torch.equal(O.dir_to_elev_azim(g["dirs"]), g["elaz"])
    close(O.dir_to_elev_azim(torch.tensor([[-0.08, 0.08, -1.0]])), torch.tensor([[3.02893472, 2.35619450]]), 1e-6)
    for k in ["normal", "thin", "fat", "tanh", "upshifted", "relu", "sin", "leaky_relu", "upshifted_softplus",
              "upshifted_relu", "cyclic"]:
        close(O.sigmoid(k)(g["sig_in"]), g[f"sig_{k}"], 1e-7, 0)
    for kind, act in [("view", "thin"), ("view", "upshifted"), ("pos", "thin"), ("pos-linear-view", "thin")]:
        h = load_golden(f"g7_refl_{kind}_{act}")
        p = golden_params(h)
        if kind == "view":
            rgb = O.view_refl(p, "", h["x"], h["view"], h["latent"], act)
        elif kind == "pos":
            rgb = O.positional_refl(p, "", h["x"], h["latent"], act)
        else:
            rgb = O.pos_linear_view_refl(p, "", h["x"], h["view"], h["latent"], act)
        close(rgb, h["rgb"], 2e-6, 1e-5)


def test_g8_mip_primitives():
    g = load_golden("g8_mip")
    y, yv = O.expected_sin(g["x"], g["var"])
    assert torch.equal(y, g["es_y"]) and torch.equal(yv, g["es_var"])
    assert torch.equal(O.integrated_pos_enc_diag(g["x"], g["var"], 0, 16), g["ipe"])
    for i in range(3):
        assert torch.equal(O.radii_x(g[f"rd{i}"]), g[f"radii{i}"])
    tm, tv, rv = O.cylinder_moments(g["t0"], g["t1"], g["rad"])
    close(tm, g["cyl_tmean"], 0, 0)
    # reference cov layout is [3, 1, T] (transposed): row 2 = t_var (z), row 0 = r_var (x)
    close(tv, g["cyl_cov"][2, 0], 1e-9, 1e-6); close(rv.expand(16), g["cyl_cov"][0, 0], 1e-12, 1e-6)
    tm, tv, rv = O.cone_moments(g["t0"], g["t1"], g["rad"])
    close(tm, g["cone_tmean"], 1e-6, 1e-6)
    close(tv, g["cone_cov"][2, 0], 1e-7, 1e-5); close(rv, g["cone_cov"][0, 0], 1e-10, 1e-5)
    # intended composed latent: finite, right shape
    rd = g["rd1"]
    lat = O.mip_latent_intended(torch.zeros_like(rd), rd, g["t0"], "cylinder", end=6.0)
    assert lat.shape == (16,) + rd.shape[:-1] + (96,) and torch.isfinite(lat).all()
    lat = O.mip_latent_intended(torch.zeros_like(rd), rd, g["t0"], "cone", end=6.0)
    assert torch.isfinite(lat).all()


def test_g9_bezier_and_dnerf():
    g = load_golden("g9_bezier")
    for n in range(2, 7):
        assert torch.equal(O.de_casteljau(g[f"coeffs{n}"], g["t"], n), g[f"dc{n}"])
    assert torch.equal(O.cubic_bezier(g["coeffs4"], g["t"], 4), g["cubic"])
    for spline in (6, 4):
        h = load_golden(f"g9_dnerf_spline{spline}")
        aux = {}
        out = O.dynamic_nerf_spline(golden_params(h), h["rays"], h["times"], float(h["near"]), float(h["far"]),
                                    int(h["steps"]), spline, act="upshifted", aux=aux)
        close(out, h["out"], 2e-6, 1e-5)
        close(aux["rigidity"], h["rigidity"], 1e-6, 1e-5)
        close(aux["dp"], h["dp"], 2e-6, 1e-5)
    # --dyn-refl-latent (make dnerf: spline 6, 3 columns, pos-linear-view): the latent rides through the spline
    for name in ("spline6_rl3_plv", "spline6_rl3_view", "spline4_rl2_plv"):
        h = load_golden(f"g9_dnerf_{name}")
        aux = {}
        out = O.dynamic_nerf_spline(golden_params(h), h["rays"], h["times"], float(h["near"]), float(h["far"]), int(h["steps"]),
                                    int(name[6]), str(h["refl_kind"]), act="upshifted", aux=aux, refl_latent=int(h["n_rl"]))
        close(out, h["out"], 2e-6, 1e-5)
        close(aux["refl_latent"], h["refl_latent"], 2e-6, 1e-5)
        close(aux["dp"], h["dp"], 2e-6, 1e-5)


def test_g10_laplace_volsdf():
    g = load_golden("g10_laplace")
    for sc in (0.1, 0.02, 1.5):
        assert torch.equal(O.laplace_cdf(g["sdf"], torch.tensor(sc)), g[f"cdf_{sc}"])
    for kind in ("mlp", "siren"):
        h = load_golden(f"g10_volsdf_{kind}")
        p = golden_params(h)
        p["scale"] = h["scale"]
        aux = {}
        out = O.volsdf(p, h["rays"], float(h["near"]), float(h["far"]), int(h["steps"]), kind, act="upshifted", aux=aux)
        close(out, h["out"], 5e-6, 1e-5)
        close(aux["weights"], h["weights"], 5e-6, 1e-5)


@pytest.mark.parametrize("kind", ["view", "pos", "pos-linear-view"])
@pytest.mark.parametrize("B", [1, 2])
def test_g11_plain_nerf(kind, B):
    h = load_golden(f"g11_plain_{kind}_b{B}")
    aux = {}
    out = O.plain_nerf(golden_params(h), h["rays"], float(h["near"]), float(h["far"]), int(h["steps"]), kind,
                       act="upshifted", bg=str(h["bg"]), aux=aux)
    close(out, h["out"], 2e-6, 1e-5)
    assert torch.equal(aux["ts"], h["ts"])
    close(aux["alpha"], h["alpha"], 2e-6, 1e-5)
    close(aux["weights"], h["weights"], 2e-6, 1e-5)


def test_g11_fp64_headroom():
    h = load_golden("g11_plain_view_fp64")
    out = O.plain_nerf(golden_params(h), h["rays"], 2.0, 6.0, 16, "view", act="upshifted")
    assert (out.double() - h["out64"]).abs().max() < 5e-6  # reference's own fp32 noise is ~3e-7


def test_g12_tiled_frame_and_psnr():
    h = load_golden("g12_tiled_frame")
    p = golden_params(h)
    fn = lambda rays: O.plain_nerf(p, rays, float(h["near"]), float(h["far"]), int(h["steps"]), "view", act="upshifted")
    frame = O.render_tiled(fn, h["c2w"], float(h["focal"]), int(h["size"]), int(h["crop_size"]))
    close(frame, h["frame"], 2e-6, 1e-5)
    psnr = O.mse2psnr(torch.nn.functional.mse_loss(frame, h["exp"]))
    assert abs(float(psnr) - float(h["psnr"])) < 1e-4


def test_g13_tiny():
    h = load_golden("g13_tiny")
    aux = {}
    out = O.tiny_nerf(golden_params(h), h["rays"], float(h["near"]), float(h["far"]), int(h["steps"]), aux=aux)
    close(out, h["out"], 2e-6, 1e-5)
    close(aux["weights"], h["weights"], 2e-6, 1e-5)


def test_ffjord_divergence_estimate_matches_reference():
    """g17: utils.div_approx on DynamicNeRF.rigid_dp as runner.py:697-700 calls it (training-mode points, e recorded)."""
    g = load_golden("g17_ffjord")
    p = golden_params(g)
    div = O.ffjord_div(p, g["pts"], g["times"], 6, g["e"])
    scale = float(g["div"].abs().max())
    assert (div - g["div"]).abs().max() <= 2e-5 * scale
    t = g["times"][None, :, None, None, None].expand(*g["pts"].shape[:-1], 1)
    assert (O.dnerf_rigid_dp(p, g["pts"], t, 6) - g["rigid_dp"]).abs().max() <= 1e-6
    term = (g["alpha"] * div.abs().square()).mean()
    assert abs(float(term) - float(g["term"])) <= 1e-4 * float(g["term"])
    assert not bool(g["term_requires_grad"])  # the reference's term is a constant for the optimiser (no create_graph)


def test_g18_aux_maps_match_the_references_own_functions():
    """N3: runner.depth_vis / flow_vis / rigidity_vis (runner.py:511-538), the raw maps of the test() loop (runner.py:894-913)
    and utils.depth_to_normals, produced by the reference itself on DynamicNeRF(spline 6)."""
    g = load_golden("g18_aux_maps")
    p = golden_params(g)
    aux = {}
    out = O.dynamic_nerf_spline(p, g["rays"], g["times"], 2.0, 6.0, int(g["steps"]), 6, act="upshifted", aux=aux)
    close(out, g["out"], 2e-6, 1e-5)
    w, ts = aux["weights"], aux["ts"]
    close(w, g["weights"], 2e-6, 1e-5)
    raw = O.volumetric_integrate(w, ts[:, None, None, None, None])
    close(raw, g["raw_depth"], 1e-5, 1e-5)
    close(O.depth_to_normals(g["raw_depth"][0]), g["depth_normals"], 1e-6, 1e-6)
    close(O.volumetric_integrate(w, aux["rigid_dp"]), g["flow_raw"], 2e-6, 1e-5)
    close(O.flow_vis(g["weights"], aux["rigid_dp"]), g["flow_vis"], 5e-6, 1e-5)
    close(O.rigidity_vis(g["weights"], aux["rigidity"]), g["rigidity_vis"], 2e-6, 1e-5)
    close(g["weights"][:-1].sum(dim=0), g["acc"], 1e-6, 1e-6)
    # depth_vis: the reference's line as written (tensor near / far with far - near = 1) and the intended reading agree
    # wherever the written result lies in [0, 1]; outside the intent clamps
    dv, dn = O.depth_vis(g["weights"], ts, float(g["vis_near"]), float(g["vis_far"]), normals_from_depth=True)
    written = g["depth_vis"]
    inside = (written >= 0) & (written <= 1)
    assert inside.any() and (~inside).any()
    close(dv[inside], written[inside], 1e-5, 1e-5)
    assert torch.equal(dv[~inside], written[~inside].clamp(0, 1))
    same = inside[1:, 1:] & inside[:-1, 1:] & inside[1:, :-1]  # normals whose three depth samples were not clamped
    assert same.any()
    close(dn[same.expand_as(dn)], g["depth_normal_vis"][same.expand_as(dn)], 2e-4, 1e-4)


def test_g19_bg_random():
    """src/nerf.py:99-103: one draw per ray times the white-background remainder (PlainNeRF end to end + the bare function)."""
    g = load_golden("g19_bg_random")
    p = golden_params(g)
    out = O.plain_nerf(p, g["rays"], 2.0, 6.0, int(g["steps"]), "view", act="upshifted", bg=("random", g["rand"]))
    close(out, g["out"], 2e-6, 1e-5)
    close(O.plain_nerf(p, g["rays"], 2.0, 6.0, int(g["steps"]), "view", act="upshifted"), g["out_black"], 2e-6, 1e-5)
    assert torch.equal(O.sky_random(g["fn_weights"], g["fn_rand"]), g["fn_sky"])
    assert g["rand"].shape[-1] == 1 and float((g["out"] - g["out_black"]).abs().max()) > 1e-3
