"""GPU parity: every C-ABI operator vs the CPU oracle (oracle/) on the golden inputs and on seeded inputs.
Tolerances: indices / pixel grid / rays / ts bit-exact; fp32 operators <= 2e-6; see each test."""
import math

import pytest
import torch

import oracle as O
from conftest import load_golden, golden_params
from oracle.procedural import proc_uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from nerf_atlas_amd import ops as _ops
    from nerf_atlas_amd import _lib
    _lib.load()  # raises if the HIP extension is missing: no fallback
    return _ops


def dev(t):
    return t.cuda()


def bits_equal(a, b):
    """bit-for-bit equality (torch.equal treats -0.0 == +0.0; the sign of zero matters downstream of atan2)."""
    a, b = a.detach().cpu().contiguous(), torch.as_tensor(b).contiguous()
    return a.shape == b.shape and torch.equal(a.view(torch.int32), b.view(torch.int32))


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


# ------------------------------------------------------------------ A1/A2 rays: bit-exact
def test_raygen_bit_exact(ops):
    g = load_golden("g1_nerf_camera")
    size = int(g["size"])
    for i, crop in enumerate(g["crops"].tolist()):
        rays = ops.raygen(dev(g["c2w"]), float(g["focal"]), size, tuple(crop))
        assert bits_equal(rays, g[f"rays{i}"]), (i, maxdiff(rays, g[f"rays{i}"]))
    crop = tuple(g["crops"].tolist()[1])
    rays = ops.raygen(dev(g["c2w"]), float(g["focal"]), size, crop, noise=dev(g["noise"]), with_noise=0.1)
    assert bits_equal(rays, g["rays_noise"])


def test_raygen_full_size_matches_oracle(ops):
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]]])
    crop = (700, 650, 100, 100)
    rays = ops.raygen(dev(c2w), focal, size, crop)
    ref = O.nerf_camera_rays(O.pixel_grid(size, crop), c2w, focal, size)
    assert bits_equal(rays, ref)
    # identity pose, crop through the image centre: rows/cols with exact zeros (+0 vs -0 matters for azim)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    for sz, crop in ((64, (28, 28, 8, 8)), (800, (396, 396, 8, 8))):
        f = 0.5 * sz / math.tan(0.5 * 0.6911)
        assert bits_equal(ops.raygen(dev(c2w), f, sz, crop), O.nerf_camera_rays(O.pixel_grid(sz, crop), c2w, f, sz))


def test_raygen_dtu(ops):
    g = load_golden("g1_dtu_camera")
    rays = ops.raygen_dtu(dev(g["pose"]), dev(g["intrinsic"]), int(g["size"]), (2, 1, 7, 9))
    assert maxdiff(rays, g["rays"]) <= 2e-6


def test_raygen_empty_and_errors(ops):
    from nerf_atlas_amd._lib import NaError
    c2w = dev(torch.eye(3, 4)[None])
    assert ops.raygen(c2w, 10.0, 8, (8, 0, 4, 4)).shape == (1, 0, 4, 6)
    with pytest.raises(NaError):
        ops.raygen(c2w, 10.0, 8, (-1, 0, 4, 4))


# ------------------------------------------------------------------ A3 sampling: ts bit-exact
def test_compute_ts_bit_exact(ops):
    g = load_golden("g2_sampling")
    for tag in ("lin", "disp", "lin128"):
        near, far, T, lind = g[f"cfg_{tag}"].tolist()
        ts, _ = ops.compute_ts(near, far, int(T), "cuda", bool(lind))
        assert torch.equal(ts.cpu(), g[f"ts_{tag}"]), tag
    ts, mids = ops.compute_ts(2.0, 6.0, 16, "cuda", perturb=1.0, rand=dev(g["rand"]))
    assert torch.equal(ts.cpu(), g["ts_perturb"]) and torch.equal(mids.cpu(), g["mids"])
    for (a, b, n) in [(0.1, 7.3, 193), (2, 6, 64), (2, 6, 1), (0.3, 1.8, 192)]:
        ts, _ = ops.compute_ts(a, b, n, "cuda")
        assert torch.equal(ts.cpu(), torch.linspace(a, b, n))


def test_compute_pts_bit_exact(ops):
    g = load_golden("g2_sampling")
    ts, _ = ops.compute_ts(2.0, 6.0, 16, "cuda")
    pts = ops.compute_pts(dev(g["rays"]), ts)
    assert torch.equal(pts.cpu(), g["pts"])


# ------------------------------------------------------------------ A5 hash: indices bit-exact
def test_hash_indices_bit_exact_and_features(ops):
    g = load_golden("g4_hash")
    p = golden_params(g)
    tables = torch.stack([p[f"embs.{i}.weight"] for i in range(8)])
    feats, idx = ops.hash_encode(dev(g["x"]), dev(tables), want_indices=True)
    assert torch.equal(idx.cpu(), g["idx"])
    assert maxdiff(feats, g["feats"]) <= 2e-6
    # large seeded case vs the oracle, including far-away / negative coordinates
    x = torch.from_numpy(proc_uniform((20000, 3), 77, 60.0))
    feats, idx = ops.hash_encode(dev(x), dev(tables), want_indices=True)
    assert torch.equal(idx.cpu(), O.hash_corner_indices(x))
    assert maxdiff(feats, O.hash_encode(x, list(tables))) <= 4e-6


def test_fourier_positional(ops):
    from oracle.procedural import proc_param
    g = load_golden("g5_fourier")
    for sigma in (16, 32):
        basis = torch.from_numpy(proc_param("basis", (3, 128))) * sigma
        out = ops.fourier_encode(dev(g[f"x_{sigma}"]), dev(basis))
        # feature-level bound from SURVEY 8(c): |build - fp64| <= 2e-4 (fp32 dot-product order noise x sin)
        assert maxdiff(out, g[f"out64_{sigma}"]) <= 2e-4
    out = ops.positional_encode(dev(g["pe_x"]), dev(g["pe_bands"]))
    assert maxdiff(out, g["pe_out"]) <= 2e-5


def test_elaz_and_sigmoids(ops):
    g = load_golden("g7_elaz_sigmoid")
    assert maxdiff(ops.view_elaz(dev(g["dirs"])), g["elaz"]) <= 2e-6
    for k in ["normal", "thin", "fat", "tanh", "upshifted", "relu", "sin", "leaky_relu", "upshifted_softplus",
              "upshifted_relu", "cyclic"]:
        assert maxdiff(ops.sigmoid(dev(g["sig_in"]), k), g[f"sig_{k}"]) <= 1e-6, k
    with pytest.raises(NotImplementedError):
        ops.sigmoid(dev(g["sig_in"]), "nope")


# ------------------------------------------------------------------ A8 compositing
def test_composite_golden(ops):
    g = load_golden("g3_composite")
    for tag, (t, sp) in {"softplus": ("ts", True), "relu": ("ts", False), "zero": ("ts_zero", True)}.items():
        rays = torch.cat([torch.zeros_like(g["r_d"]), g["r_d"]], dim=-1)
        for bg, key in (("black", f"out_{tag}"), ("white", f"white_{tag}")):
            out, a, w = ops.composite(dev(g["density"]), dev(g["rgb"]), dev(g[t]), dev(rays), softplus=sp, bg=bg)
            assert maxdiff(a, g[f"alpha_{tag}"]) <= 1e-6
            assert maxdiff(w, g[f"weights_{tag}"]) <= 1e-6
            assert maxdiff(out, g[key]) <= 2e-6
    assert maxdiff(ops.integrate(dev(g["weights_softplus"]), dev(g["rgb"])), g["out_softplus"]) <= 2e-6


def test_composite_full_size_properties(ops):
    """800x800x128-sized sanity: weights are a partition of unity with the last interval (Q3)."""
    T, R = 128, 100 * 100
    gen = torch.Generator().manual_seed(0)
    density = torch.randn(T, R, generator=gen)
    rgb = torch.rand(T, R, 3, generator=gen)
    rays = torch.cat([torch.zeros(R, 3), torch.randn(R, 3, generator=gen)], -1)
    ts = torch.linspace(2, 6, T)
    out, a, w = ops.composite(dev(density), dev(rgb), dev(ts), dev(rays))
    assert float((w.sum(0) - 1).abs().max()) < 1e-5
    ref_a, ref_w = O.alpha_from_density(density[:, None, None, :1000].reshape(T, 1, 1, 1000), ts,
                                        rays[None, None, :1000, 3:])
    assert maxdiff(w[:, :1000], ref_w.reshape(T, 1000)) <= 2e-6


# ------------------------------------------------------------------ A6 / A11 / A12 small operators
def test_mip_encode_intended_layout(ops):
    g = load_golden("g8_mip")
    rd = g["rd1"]
    rays = torch.cat([torch.full_like(rd, 0.25), rd], -1)
    for kind in ("cylinder", "cone"):
        got = ops.mip_encode(dev(rays), dev(g["t0"]), kind, 6.0)
        ref = O.mip_latent_intended(rays[..., :3], rd, g["t0"], kind, end=6.0)
        assert got.shape == ref.shape
        assert maxdiff(got, ref) <= 5e-5  # sin of arguments up to 2^15 * |x|


def test_laplace_and_bezier(ops):
    g = load_golden("g10_laplace")
    for sc in (0.1, 0.02, 1.5):
        d = ops.laplace_density(dev(-g["sdf"]), dev(torch.tensor(sc)))
        assert maxdiff(d, g[f"cdf_{sc}"] / sc) <= 2e-6 * (1 / sc)
    b = load_golden("g9_bezier")
    t = b["t"]
    for n in range(2, 7):
        co = b[f"coeffs{n}"]  # [n, 7, 3]
        est = torch.cat([torch.zeros(7, 1), co.permute(1, 0, 2).reshape(7, 3 * n)], -1)
        pts = torch.zeros(7, 3)
        out, dp, rig = ops.bezier_warp(dev(est), dev(pts), dev(t.reshape(7)), n)
        ref = b["cubic"] if n == 4 else b[f"dc{n}"]
        assert maxdiff(dp, ref) <= 2e-6
        assert maxdiff(out, ref * 0.5) <= 2e-6 and maxdiff(rig, torch.full((7, 1), 0.5)) <= 1e-7


# ------------------------------------------------------------------ A4 SkipConnMLP
MLP_CASES = ["tiny", "first", "view", "posrefl", "delta6", "sdfmlp", "siren", "mipfirst", "plv_view", "plv_pos"]


def _mlp_oracle(g, p):
    enc = None
    if str(g["enc"]) == "hash":
        enc = lambda x: O.hash_encode(x, [p[f"enc.embs.{i}.weight"] for i in range(8)])
    elif str(g["enc"]) == "fourier16":
        enc = lambda x: O.fourier_encode(x, p["enc.basis"])
    return enc


def _linear_chain(ops, g, p):
    """SkipConnMLP through na_linear_f32 (exact fp32), encoders through their own kernels."""
    act = str(g["act"])
    x = dev(g["p"])
    parts = [x]
    if str(g["enc"]) == "hash":
        parts.append(ops.hash_encode(x, dev(torch.stack([p[f"enc.embs.{i}.weight"] for i in range(8)]))))
    elif str(g["enc"]) == "fourier16":
        parts.append(ops.fourier_encode(x, dev(p["enc.basis"])))
    if "latent" in g:
        parts.append(dev(g["latent"]))
    init = torch.cat(parts, -1)
    L = int(g["layers"])
    h = ops.linear_f32(init, dev(p["init.weight"]), dev(p["init.bias"]))
    inter = [h]
    for i in range(L):
        skip = (i % 3 == 0) and i != L - 1
        h = ops.linear_f32(h, dev(p[f"layers.{i}.weight"]), dev(p[f"layers.{i}.bias"]), pre_act=act,
                           x1=init if skip else None)
        inter.append(h)
    y = ops.linear_f32(h, dev(p["out.weight"]), dev(p["out.bias"]), pre_act=act)
    return y, inter


@pytest.mark.parametrize("case", MLP_CASES)
def test_skip_mlp_linear_f32_path(ops, case):
    g = load_golden(f"g6_mlp_{case}")
    p = golden_params(g, sigma=16.0)
    y, inter = _linear_chain(ops, g, p)
    tol = 2e-4 if case == "sdfmlp" else 2e-5  # Fourier features carry ~1e-4 fp32 order noise
    for i, t in enumerate(inter):
        assert maxdiff(t, g[f"inter{i}"]) <= tol * max(1.0, float(g[f"inter{i}"].abs().max())), (case, i)
    assert maxdiff(y, g["y"]) <= tol * max(1.0, float(g["y"].abs().max()))


def _desc_for(ops, g, p):
    enc = str(g["enc"])
    in_size = g["p"].shape[1]
    lat = g["latent"].shape[1] if "latent" in g else 0
    enc_kind, enc_dims = "none", 0
    if enc == "hash":
        enc_kind, enc_dims = "hash", 35
    elif enc == "fourier16":
        enc_kind, enc_dims = "fourier", 256
    return ops.make_desc(in_size, enc_kind, enc_dims, lat, int(g["layers"]), p["init.weight"].shape[0], int(g["out"]), 3,
                         str(g["act"]))


FUSED_CASES = ["tiny", "first", "view", "posrefl", "delta6", "sdfmlp", "siren", "mipfirst", "plv_pos"]


@pytest.mark.parametrize("case", FUSED_CASES)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_skip_mlp_fused_mfma(ops, case, precision):
    g = load_golden(f"g6_mlp_{case}")
    p = golden_params(g, sigma=16.0)
    desc = _desc_for(ops, g, p)
    L = int(g["layers"])
    ws = [p["init.weight"]] + [p[f"layers.{i}.weight"] for i in range(L)] + [p["out.weight"]]
    bs = [p["init.bias"]] + [p[f"layers.{i}.bias"] for i in range(L)] + [p["out.bias"]]
    packed = ops.mlp_pack(desc, precision, [dev(w) for w in ws], [dev(b) for b in bs])
    enc_params = None
    if str(g["enc"]) == "hash":
        enc_params = dev(torch.stack([p[f"enc.embs.{i}.weight"] for i in range(8)]))
    elif str(g["enc"]) == "fourier16":
        enc_params = dev(p["enc.basis"])
    y = ops.mlp_forward(desc, precision, packed, dev(g["p"]), dev(g["latent"]) if "latent" in g else None, enc_params)
    scale = max(1.0, float(g["y"].abs().max()))
    err = maxdiff(y, g["y"]) / scale
    # bf16x3 (3 MFMA products) is fp32-class; plain bf16 carries 2^-9 relative input rounding per layer
    tol = {"bf16x3": 3e-4 if case == "sdfmlp" else 5e-5, "bf16": 6e-2}[precision]
    assert err <= tol, (case, precision, err)


def test_fused_mlp_large_ragged_n(ops):
    """N not a multiple of the workgroup tile, many persistent iterations; fused == fp32 chain."""
    g = load_golden("g6_mlp_first")
    p = golden_params(g)
    desc = _desc_for(ops, g, p)
    ws = [p["init.weight"]] + [p[f"layers.{i}.weight"] for i in range(4)] + [p["out.weight"]]
    bs = [p["init.bias"]] + [p[f"layers.{i}.bias"] for i in range(4)] + [p["out.bias"]]
    tables = dev(torch.stack([p[f"enc.embs.{i}.weight"] for i in range(8)]))
    N = 256 * 300 + 77
    x = dev(torch.from_numpy(proc_uniform((N, 3), 9, 3.0)))
    g2 = dict(g)
    g2["p"] = x.cpu()
    y_ref, _ = _linear_chain(ops, g2, p)
    for precision, tol in (("bf16x3", 5e-5), ("bf16", 6e-2)):
        packed = ops.mlp_pack(desc, precision, [dev(w) for w in ws], [dev(b) for b in bs])
        y = ops.mlp_forward(desc, precision, packed, x, None, tables)
        scale = float(y_ref.abs().max())
        assert float((y - y_ref).abs().max()) / scale <= tol, precision


def test_unsupported_mlp_shape_fails_loudly(ops):
    from nerf_atlas_amd._lib import NaError
    desc = ops.make_desc(6, "none", 0, 128, 2, 128, 1, 3, "sin")
    assert ops.mlp_packed_bytes(desc, "bf16") == 0
    with pytest.raises(NaError):
        ops.mlp_pack(desc, "bf16", [], [])


def test_view_rows_is_the_cat_of_points_and_broadcast_elaz():
    """na_view_rows (round 5): [x | elev, azim] rows of the View reflectance's input in one launch = the elaz + expand + cat it replaces"""
    import torch
    from nerf_atlas_amd import ops
    torch.manual_seed(3)
    T, R = 7, 333
    pts = torch.randn(T, R, 3, device="cuda")
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
    rows = ops.view_rows(pts, dirs)
    ref = torch.cat([pts, ops.view_elaz(dirs).unsqueeze(0).expand(T, R, 2)], dim=-1)
    assert rows.shape == (T, R, 5) and torch.equal(rows, ref)
