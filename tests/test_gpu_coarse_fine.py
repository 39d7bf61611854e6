"""Coarse -> fine rendering (BASELINE config 2 "64 + 128"; VERDICT r03 missing 4).  The reference's sample_pdf / CoarseFineNeRF
(src/nerf.py:548-580, 1745-1779) is dead code, so this is PARITY UNPINNED by construction: the kernels are checked against the
fp64 restatement of the intended reading (oracle.sample_pdf_intended) and the oracle's per-ray-step forward.
  * na_resample_ts: the N new positions per ray within 2e-6 of the fp64 restatement (fp32 output rounding: |t| <= 6), for the
    deterministic linspace draw and for random draws, on peaked / flat / empty weight profiles; merged rows sorted, a permutation
    of (coarse, fine), ragged R (not a multiple of the 64-ray tile), T and N not multiples of 64.
  * na_render_plain_view_ls_rayts: per-ray steps equal to the shared ones reproduce na_render_plain_view_ls bit for bit; with
    resampled steps it matches the oracle's per-ray forward within north_star's 1e-4 in both parity modes.
  * PlainNeRF.forward_coarse_fine (64 + 128) end to end against the oracle chain on the reference's golden model."""
import pytest
import torch

import oracle as O
from conftest import load_golden, golden_params
from test_gpu_render_ls import pack_ls

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from nerf_atlas_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def _weights(T, R, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand(T, R, generator=g) ** 4                      # generic
    w[:, 0] = 0                                                 # empty ray: every interval gets the 1e-5 floor
    w[:, 1] = 0; w[T // 3, 1] = 0.9                             # one surface
    w[:, 2] = 1.0 / T                                           # flat
    w[:, 3] = 0; w[0, 3] = 0.5; w[T - 2, 3] = 0.5               # mass in the first and the last interval
    return w / w.sum(0, keepdim=True).clamp(min=1)


@pytest.mark.parametrize("T,N,R", [(64, 128, 1000), (33, 7, 65), (100, 200, 130), (2, 5, 5)])
@pytest.mark.parametrize("rand_u", [False, True])
def test_resample_ts_against_the_fp64_restatement(ops, T, N, R, rand_u):
    ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
    w = _weights(T, R, 7 * T + N)
    u = torch.rand(N, R, generator=torch.Generator().manual_seed(N)) if rand_u else None
    merged, fine = ops.resample_ts(ts, w.cuda(), N, None if u is None else u.cuda(), want_fine=True)
    assert fine.shape == (R, N) and merged.shape == (R, T + N)
    ref = O.sample_pdf_intended(ts.cpu(), w, N, u)               # [N, R] fp64
    err = float((fine.cpu().double() - ref.t()).abs().max())
    assert err <= 2e-6, err
    # the merged row: sorted, and exactly the multiset (coarse steps, new positions)
    m = merged.cpu()
    assert bool((m[:, 1:] >= m[:, :-1]).all())
    both = torch.cat([ts.cpu().expand(R, T), fine.cpu()], dim=1)
    assert torch.equal(m, torch.sort(both, dim=1, stable=True).values)
    ref_m = O.merge_ts_intended(ts.cpu().double(), ref)
    assert float((m.double() - ref_m.t()).abs().max()) <= 2e-6
    if not rand_u:  # the deterministic draw keeps the end points of the ray
        assert float((fine[:, 0].cpu() - ts[0].cpu()).abs().max()) <= 1e-6 and float((fine[:, -1].cpu() - ts[-1].cpu()).abs().max()) <= 1e-5


def test_resample_ts_errors(ops):
    from nerf_atlas_amd._lib import NaError
    ts, _ = ops.compute_ts(2.0, 6.0, 600, "cuda")
    with pytest.raises(NaError):
        ops.resample_ts(ts, torch.rand(600, 8, device="cuda"), 600)   # T + N beyond the kernel's LDS budget: loud, not wrong
    ts1, _ = ops.compute_ts(2.0, 6.0, 4, "cuda")
    out = ops.resample_ts(ts1, torch.rand(4, 0, device="cuda"), 8)     # empty batch
    assert out.shape == (0, 12)


def test_resample_ts_single_draw_and_non_finite_weights(ops):
    """(i) N = 1: torch.linspace(0, 1, 1) is [0], the single deterministic draw sits on the first coarse step (the
    kernel's own linspace used to give u = 1); (ii) NaN weights make NaN positions: they sort LAST and every slot of the merged row
    is written -- the T coarse steps are all there, the rest is NaN, nothing is uninitialised memory."""
    T, R = 16, 70
    ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
    w = _weights(T, R, 3).cuda()
    merged, fine = ops.resample_ts(ts, w, 1, want_fine=True)
    assert fine.shape == (R, 1) and bool((fine[:, 0] == ts[0]).all())
    assert torch.equal(merged.cpu(), torch.sort(torch.cat([ts.cpu().expand(R, T), fine.cpu()], 1), dim=1, stable=True).values)
    assert float((fine.cpu().double() - O.sample_pdf_intended(ts.cpu(), w.cpu(), 1).t()).abs().max()) <= 2e-6
    bad = w.clone()
    bad[5, 7] = float("nan")
    for _ in range(3):
        torch.empty(R * (T + 9), device="cuda").fill_(-7.0)   # (what a recycled allocation would hold)
        m = ops.resample_ts(ts, bad, 9).cpu()
        row = m[7]
        fin = row[torch.isfinite(row)]
        nan_at = torch.isnan(row).nonzero().flatten()
        assert bool((fin[1:] >= fin[:-1]).all()) and not bool((row == -7.0).any())        # written everywhere, sorted
        assert nan_at.numel() == 0 or int(nan_at.min()) == fin.numel()                     # NaN positions, if any, sort last
        assert all(bool((fin == t).any()) for t in ts.cpu()) and float(fin.min()) >= 2.0 and float(fin.max()) <= 6.0 + 1e-5
        ok = torch.ones(R, dtype=torch.bool); ok[7] = False
        assert bool(torch.isfinite(m[ok]).all()) and bool((m[ok][:, 1:] >= m[ok][:, :-1]).all())


@pytest.mark.parametrize("prec", ["f16x", "bf16x3", "bf16"])
def test_per_ray_steps_equal_to_the_shared_ones_are_the_same_launch(ops, prec):
    h = load_golden("g11_plain_view_b2")
    p = golden_params(h)
    packed, tables = pack_ls(ops, p, prec)
    rays = h["rays"].cuda()
    for T in (int(h["steps"]), 45):
        ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
        a, _, wa = ops.render_plain_view_ls(rays, ts, tables, packed, prec, "upshifted", "white", want_weights=True)
        ts_ray = ts.expand(tuple(rays.shape[:-1]) + (T,)).contiguous()
        b, _, wb = ops.render_plain_view_ls_rayts(rays, ts_ray, tables, packed, prec, "upshifted", "white", want_weights=True)
        assert torch.equal(a, b) and torch.equal(wa, wb)


@pytest.mark.parametrize("prec", ["f16x", "bf16x3"])
def test_fine_pass_with_resampled_steps_vs_oracle(ops, prec):
    h = load_golden("g11_plain_view_b2")
    p = golden_params(h)
    packed, tables = pack_ls(ops, p, prec)
    rays = h["rays"]
    Tc, N = 24, 40
    ts, _ = ops.compute_ts(2.0, 6.0, Tc, "cuda")
    _, _, w = ops.render_plain_view_ls(rays.cuda(), ts, tables, packed, prec, "upshifted", "black", want_weights=True)
    u = torch.rand((N,) + tuple(rays.shape[:-1]), generator=torch.Generator().manual_seed(3))
    merged = ops.resample_ts(ts, w, N, u.cuda())
    out, _, wf = ops.render_plain_view_ls_rayts(rays.cuda(), merged, tables, packed, prec, "upshifted", "white", want_weights=True)
    aux = {}
    ts_ray = merged.cpu().movedim(-1, 0).contiguous()             # the oracle takes [T, *batch]
    ref = O.plain_nerf_rayts(p, rays, ts_ray, "view", act="upshifted", bg="white", aux=aux)
    e = float((out.cpu() - ref).abs().max())
    print(f"fine pass [{prec}] on resampled steps: L-inf {e:.2e}")
    assert e <= 1e-4, e
    assert float((wf.cpu() - aux["weights"]).abs().max()) <= 1e-4


@pytest.mark.parametrize("prec", ["f16x", "bf16x3"])
def test_plain_nerf_coarse_fine_64_128_end_to_end(prec):
    """config 2's "64 + 128": model layer against the oracle chain (coarse forward -> fp64 resampling -> per-ray-step forward).  The
    oracle resamples from ITS OWN coarse weights; the new positions depend on them continuously, so the end-to-end bar is the
    renderer's 1e-4 plus the resampling's sensitivity, measured here at <= 2e-4 on this model and asserted at that."""
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import config
    from test_gpu_fullsize import load_params
    h = load_golden("g11_plain_view_b2")
    p = golden_params(h)
    m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
    load_params(m, p)
    rays = h["rays"]
    config.set_precision(prec)
    try:
        out = m.forward_coarse_fine(rays.cuda(), 128)
        assert m.ts.shape == (64,) and m.ts_ray.shape == tuple(rays.shape[:-1]) + (192,) and m.weights.shape[0] == 192
        assert float((m.weights.sum(0) - 1).abs().max()) <= 1e-5 or m.bg != "white"
        # the fine pass alone, on the build's own steps: the renderer's bar
        ref_same = O.plain_nerf_rayts(p, rays, m.ts_ray.cpu().movedim(-1, 0).contiguous(), "view", act="upshifted")
        e_same = float((out.cpu() - ref_same).abs().max())
        assert e_same <= 1e-4, e_same
        # the whole chain in the oracle
        aux = {}
        O.plain_nerf(p, rays, 2.0, 6.0, 64, "view", act="upshifted", aux=aux)
        fine = O.sample_pdf_intended(aux["ts"], aux["weights"], 128)
        ts_ray = O.merge_ts_intended(aux["ts"].double(), fine).float()
        ref = O.plain_nerf_rayts(p, rays, ts_ray, "view", act="upshifted")
        e = float((out.cpu() - ref).abs().max())
        print(f"coarse -> fine 64 + 128 [{prec}]: fine pass on the same steps {e_same:.2e}, whole chain {e:.2e}, "
              f"steps {float((m.ts_ray.cpu().movedim(-1, 0) - ts_ray).abs().max()):.2e}")
        assert e <= 2e-4, e
        # more samples where the mass is: the fine pass is not the coarse image
        assert not torch.equal(out, m.coarse)
        # the depth map of a coarse -> fine frame integrates the weights against the PER-RAY steps (model.ts stays
        # the [T] shared steps; render.depth_map used to broadcast it against [T + N] weight rows)
        from types import SimpleNamespace
        from nerf_atlas_amd import render
        depth = render.depth_map(SimpleNamespace(nerf=m))
        want = (m.weights.cpu().double() * m.ts_ray.cpu().movedim(-1, 0).double()).sum(0)
        assert depth.shape == tuple(rays.shape[:-1]) + (1,) and float((depth.cpu()[..., 0].double() - want).abs().max()) <= 1e-5
        m(rays.cuda())
        assert m.ts_ray is None and render.depth_map(SimpleNamespace(nerf=m)).shape == depth.shape
    finally:
        config.set_precision("bf16x3")


def test_coarse_fine_full_frame_properties():
    """BASELINE's full size (800 x 800 rays, 64 + 128): size-independent properties -- every ray's 192 steps are sorted, contain its
    64 coarse steps, stay inside [near, far]; the fine weights partition unity against a white background; a row band rendered
    alone reproduces the frame's rows bit for bit (resampling and the per-ray-step renderer are per-ray computations)."""
    import math
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import config, ops
    torch.manual_seed(2)
    m = nerf.PlainNeRF(steps=64, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", bg="white").cuda().eval()
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]]]).cuda()
    config.set_precision("f16x")
    try:
        rays = ops.raygen(c2w, focal, size, (0, 0, size, size))
        out = m.forward_coarse_fine(rays, 128)
        ts_f, w = m.ts_ray, m.weights
        assert out.shape == (1, size, size, 3) and torch.isfinite(out).all()
        assert ts_f.shape == (1, size, size, 192)
        assert bool((ts_f[..., 1:] >= ts_f[..., :-1]).all())
        assert float(ts_f.min()) >= 2.0 - 1e-6 and float(ts_f.max()) <= 6.0 + 1e-5
        coarse, _ = ops.compute_ts(2.0, 6.0, 64, "cuda")
        pos = torch.searchsorted(ts_f.reshape(-1, 192), coarse.expand(size * size, 64).contiguous())
        assert bool((torch.gather(ts_f.reshape(-1, 192), 1, pos.clamp(max=191)) == coarse).all())   # the coarse steps are among them
        assert float((w.sum(0) - 1).abs().max()) <= 1e-5
        band = ops.raygen(c2w, focal, size, (299, 0, 101, size))
        assert torch.equal(m.forward_coarse_fine(band, 128), out[:, 299:400])
    finally:
        config.set_precision("bf16x3")
