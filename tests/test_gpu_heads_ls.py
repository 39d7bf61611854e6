"""GPU parity of the one-launch PlainNeRF renderers with the reference's other two colour heads (round 6; csrc/ls_sched_plain_pos.inc,
ls_sched_plain_plv.inc = MODEL 7 / 8 of render_ls_kernel, f16x):

  --refl-kind pos              `make original` (/root/reference makefile:8-13; src/refl.py:230-245)
  --refl-kind pos-linear-view  `make dnerf`    (makefile:106-114; src/refl.py:248-290), with DynamicNeRF's refl_latent columns

against the reference goldens g11 (1e-4 L-inf on RGB, alpha, weights -- north_star's bar) and, on bigger / ragged batches with
procedural weights, explicit points and refl_latent rows, against the unfused bf16x3 operator chain (the path these shapes took
before round 6, itself pinned by the same goldens in tests/test_gpu_models.py)."""
import pytest
import torch

from conftest import load_golden, golden_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


@pytest.fixture()
def na():
    assert torch.cuda.is_available()
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.refl as refl
    from nerf_atlas_amd import config, ops, utils

    class NS:
        pass
    ns = NS()
    ns.nerf, ns.refl, ns.config, ns.ops, ns.utils = nerf, refl, config, ops, utils
    keep = config.precision
    yield ns
    config.set_precision(keep)


def count_calls(monkeypatch, ops, name):
    calls = []
    fn = getattr(ops, name)

    def wrapped(*a, **k):
        calls.append(1)
        return fn(*a, **k)
    monkeypatch.setattr(ops, name, wrapped)
    return calls


def maxdiff(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def build(na, kind, steps, near, far, bg="black", n_rl=0, act="upshifted"):
    m = na.nerf.PlainNeRF(steps=steps, t_near=near, t_far=far, intermediate_size=64, sigmoid_kind=act, bg=bg)
    m.set_refl(na.refl.refl_kinds[kind](latent_size=64 + n_rl, act=act, out_features=3))
    return m.cuda().eval()


def procedural_(m):
    """oracle/procedural.py values for every parameter of the model (the recipe of the goldens: O(1) activations, unit-variance tables)"""
    from oracle.procedural import proc_param
    sd = m.state_dict()
    for k, v in sd.items():
        if v.numel() and v.dtype == torch.float32 and not k.endswith("primes"):
            v.copy_(torch.from_numpy(proc_param(k, tuple(v.shape))))


@pytest.mark.parametrize("kind,fn", [("pos", "render_plain_pos_ls"), ("pos-linear-view", "render_plain_plv_ls")])
@pytest.mark.parametrize("B", [1, 2])
def test_reference_goldens_through_the_one_launch_renderer(na, monkeypatch, kind, fn, B):
    h = load_golden(f"g11_plain_{kind}_b{B}")
    m = build(na, kind, int(h["steps"]), float(h["near"]), float(h["far"]), bg=str(h["bg"]))
    sd = m.state_dict()
    for k, v in golden_params(h).items():
        sd[k].copy_(v)
    na.config.set_precision("f16x")
    calls = count_calls(monkeypatch, na.ops, fn)
    noted = set(na.utils._noted)
    out = m(h["rays"].cuda())
    assert len(calls) == 1, "the fused launch did not run"
    assert set(na.utils._noted) == noted, "a fallback note was printed for a shape the fused schedules serve"
    assert maxdiff(out, h["out"]) <= 1e-4
    assert torch.equal(m.ts.cpu(), h["ts"])
    assert maxdiff(m.alpha, h["alpha"]) <= 1e-4 and maxdiff(m.weights, h["weights"]) <= 1e-4


@pytest.mark.parametrize("kind,n_rl", [("pos", 0), ("pos-linear-view", 0), ("pos-linear-view", 1), ("pos-linear-view", 2), ("pos-linear-view", 3)])
@pytest.mark.parametrize("shape,T", [((1, 7, 9), 48), ((2, 16, 16), 128), ((1, 33, 31), 70), ((3,), 1)])
def test_fused_heads_vs_the_unfused_chain(na, monkeypatch, kind, n_rl, shape, T):
    """ragged ray counts, T not a multiple of 32, T = 1, explicit (warped) points, refl_latent rows passed by pitch"""
    m = build(na, kind, T, 2.0, 6.0, bg="white" if n_rl == 1 else "black", n_rl=n_rl, act="thin" if n_rl == 2 else "upshifted")
    procedural_(m)
    g = torch.Generator().manual_seed(5 + n_rl)
    o = torch.tensor([0.1, -0.2, 4.0]) + 0.05 * torch.randn(shape + (3,), generator=g)
    d = torch.nn.functional.normalize(torch.tensor([0.0, 0.05, -1.0]) + 0.15 * torch.randn(shape + (3,), generator=g), dim=-1) * 1.1
    rays = torch.cat([o, d], dim=-1).cuda()
    # refl_latent as DynamicNeRF hands it over: a column slice of a wider row-major tensor
    wide = (0.7 * torch.randn((T,) + shape + (n_rl + 2,), generator=g)).cuda()
    rl = wide[..., 1:1 + n_rl] if n_rl else None
    fn = "render_plain_pos_ls" if kind == "pos" else "render_plain_plv_ls"
    calls = count_calls(monkeypatch, na.ops, fn)
    res = {}
    for prec in ("bf16x3", "f16x"):
        na.config.set_precision(prec)
        if n_rl or T == 70:
            pts, ts, r_o, r_d, _ = na.nerf.compute_pts_ts(rays, m.t_near, m.t_far, m.steps)
            pts = (pts + 0.02 * torch.sin(pts * 3.0)).contiguous()
            out = m.from_pts(pts, ts, r_o, r_d, refl_latent=rl, rays=rays)
        else:
            out = m(rays)
        res[prec] = (out.clone(), m.alpha.clone(), m.weights.clone())
    assert len(calls) == 1, "f16x must take the fused launch, bf16x3 the operator chain"
    assert torch.isfinite(res["f16x"][0]).all()
    for a, b in zip(res["f16x"], res["bf16x3"]):
        assert maxdiff(a, b) <= 1e-4
    # the weights partition unity with the background (a property of the compositing, any size)
    w = res["f16x"][2]
    assert float(w.sum(dim=0).max()) <= 1.0 + 1e-5


def test_refl_latent_columns_reach_both_mlps(na):
    """a change in ONE refl_latent column must move the colour (pos takes it in the [hash' | x] group, view in the geometry chunk);
    the fused launch and the operator chain must agree on by how much"""
    T, shape = 40, (2, 5, 6)
    m = build(na, "pos-linear-view", T, 2.0, 6.0, n_rl=3)
    procedural_(m)
    g = torch.Generator().manual_seed(11)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.05 * torch.randn(shape + (3,), generator=g)
    d = torch.tensor([0.0, 0.0, -1.0]) + 0.1 * torch.randn(shape + (3,), generator=g)
    rays = torch.cat([o, d], dim=-1).cuda()
    pts, ts, r_o, r_d, _ = na.nerf.compute_pts_ts(rays, 2.0, 6.0, T)
    rl = (0.5 * torch.randn((T,) + shape + (3,), generator=g)).cuda()
    outs = {}
    for prec in ("bf16x3", "f16x"):
        na.config.set_precision(prec)
        base = m.from_pts(pts, ts, r_o, r_d, refl_latent=rl, rays=rays).clone()
        deltas = []
        for j in range(3):
            rl2 = rl.clone()
            rl2[..., j] += 1.5
            deltas.append(m.from_pts(pts, ts, r_o, r_d, refl_latent=rl2, rays=rays) - base)
        outs[prec] = torch.stack(deltas)
    for j in range(3):
        assert float(outs["bf16x3"][j].abs().max()) > 1e-3, f"column {j} does not influence the reference chain: weak test"
    assert maxdiff(outs["f16x"], outs["bf16x3"]) <= 2e-4  # (a difference of two renders, each within 1e-4)


def test_whole_frame_band_equals_slab_rows(na):
    """size-independent property at full width: a row band rendered alone == the same rows of a taller slab, bit for bit (rays are
    independent; the workgroup-private park and the carried transmittance must not leak between rays)"""
    for kind in ("pos", "pos-linear-view"):
        m = build(na, kind, 128, 2.0, 6.0)
        procedural_(m)
        na.config.set_precision("f16x")
        import math
        size = 800
        focal = 0.5 * size / math.tan(0.5 * 0.6911)
        c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device="cuda")
        slab = na.ops.raygen(c2w, focal, size, (300, 0, 24, size))
        band = na.ops.raygen(c2w, focal, size, (307, 0, 5, size))
        a = m(slab, want_weights=False)
        b = m(band, want_weights=False)
        assert torch.isfinite(a).all()
        assert torch.equal(a[:, 7:12], b), kind
        assert torch.equal(m(slab, want_weights=False), a), "not reproducible"


@pytest.mark.parametrize("act", ["normal", "thin", "fat", "upshifted", "tanh", "cyclic"])
def test_pos_linear_view_activation_kinds(na, monkeypatch, act):
    """MODEL 8 applies the head's activation to 67 rows per sample inside the kernel and takes the four sigmoid-shaped kinds (round 6:
    the rolled loop that served the others cost every kind 32 spilled registers per pass); any other kind renders through the
    unfused operators -- same model object, same result class -- and the C ABI refuses it instead of computing something else."""
    from nerf_atlas_amd import _lib
    na.config.set_precision("f16x")
    m = build(na, "pos-linear-view", 16, 2.0, 6.0, act=act)
    torch.manual_seed(3)
    rays = torch.cat([torch.tensor([0.0, 0.0, 4.0]).expand(2, 5, 5, 3), torch.randn(2, 5, 5, 3) * 0.05 + torch.tensor([0.0, 0.0, -1.0])], -1).cuda().contiguous()
    calls = count_calls(monkeypatch, na.ops, "render_plain_plv_ls")
    with torch.no_grad():
        out = m(rays)
        fused = act in ("normal", "thin", "fat", "upshifted")
        assert (m._fusable_head() == "plv") == fused and len(calls) == (1 if fused else 0)
        na.config.set_precision("bf16x3")
        ref = m(rays)   # the unfused chain in the split-bf16 class
    assert len(calls) == (1 if fused else 0)
    assert maxdiff(out, ref) <= 1e-4
    if not fused:
        with pytest.raises(_lib.NaError):
            r = m.refl
            na.ops.render_plain_plv_ls(rays, m.ts if m.ts is not None else na.ops.compute_ts(2.0, 6.0, 16, rays.device)[0], m.first.enc.tables(), r.pos.enc.tables(),
                                       m.packed_head_ls("plv", "f16x", 0), "f16x", act, "black", False)
