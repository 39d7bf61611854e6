"""N4 SDF marching (src/march.py): the oracle restatement against outputs of the reference's own functions
(tests/golden/g15_march.npz: analytic two-sphere SDF and the reference SIREN SDF with procedural weights), and the HIP
path (nerf_atlas_amd/march.py + csrc/march.hip) against the oracle."""
import pytest
import torch

import oracle as O
from conftest import load_golden, golden_params


def analytic(p):
    a = torch.linalg.norm(p - torch.tensor([0.1, -0.2, 0.0], device=p.device), dim=-1) - 1.1
    b = torch.linalg.norm(p - torch.tensor([0.9, 0.6, 0.3], device=p.device), dim=-1) - 0.5
    return torch.minimum(a, b).unsqueeze(-1)


def siren_fn(g):
    p = golden_params(g)
    return lambda x: O.skip_mlp(p, "siren.", x, act="sin")


def test_oracle_marching_matches_reference():
    g = load_golden("g15_march")
    r_o, r_d, jit = g["r_o"], g["r_d"], float(g["jitter"])
    for tag, fn, tol in (("an", analytic, 0.0), ("nn", siren_fn(g), 2e-5)):
        near, far = float(g[f"{tag}_near"]), float(g[f"{tag}_far"])
        pts, hits, dist, none = O.sphere_march(fn, r_o, r_d, iters=24, eps=1e-3, near=near, far=far)
        assert none is None and torch.equal(hits, g[f"{tag}_sm_hits"])
        assert (dist - g[f"{tag}_sm_dist"]).abs().max() <= tol and (pts - g[f"{tag}_sm_pts"]).abs().max() <= tol
        tput, best, lastp, firstn = O.throughput_with_sign_change(fn, r_o, r_d, near, far, 40, jit)
        assert torch.equal(lastp, g[f"{tag}_last"]) and torch.equal(firstn, g[f"{tag}_first"])
        assert (tput - g[f"{tag}_tp"]).abs().max() <= tol and (best - g[f"{tag}_best"]).abs().max() <= tol
        pts, hits, _, tput2 = O.bisect(fn, r_o, r_d, iters=40, near=near, far=far, jitter=jit)
        assert torch.equal(hits, g[f"{tag}_bi_hits"])
        assert (pts - g[f"{tag}_bi_pts"]).abs().max() <= max(tol, 1e-6) and (tput2 - g[f"{tag}_bi_tput"]).abs().max() <= tol
    assert bool(g["an_sm_hits"].any()) and not bool(g["an_sm_hits"].all())  # the fixture has hits and misses


@pytest.mark.gpu
def test_hip_marching_logic_is_exact_on_an_analytic_sdf():
    """The SDF callable is torch arithmetic (on the GPU here, on the CPU for the oracle: last-ulp differences in norm()),
    so decisions -- hits, step indices -- must be identical and distances within 5e-5 after 24 accumulated steps."""
    from nerf_atlas_amd import march
    g = load_golden("g15_march")
    r_o, r_d, jit = g["r_o"].cuda(), g["r_d"].cuda(), float(g["jitter"])
    near, far = float(g["an_near"]), float(g["an_far"])
    pts, hits, dist, none = march.sphere_march(analytic, r_o, r_d, iters=24, eps=1e-3, near=near, far=far)
    ref = O.sphere_march(analytic, g["r_o"], g["r_d"], iters=24, eps=1e-3, near=near, far=far)
    assert none is None and torch.equal(hits.cpu(), ref[1])
    assert (dist.cpu() - ref[2]).abs().max() <= 5e-5 and (pts.cpu() - ref[0]).abs().max() <= 5e-5
    assert torch.equal(hits.cpu(), g["an_sm_hits"])
    tput, best, lastp, firstn = march.throughput_with_sign_change(analytic, r_o, r_d, near, far, 40, jitter=jit)
    assert torch.equal(lastp.cpu(), g["an_last"]) and torch.equal(firstn.cpu(), g["an_first"])
    assert (tput.cpu() - g["an_tp"]).abs().max() <= 5e-6 and (best.cpu() - g["an_best"]).abs().max() <= 5e-6
    pts, hits, _, tput2 = march.bisect(analytic, r_o, r_d, iters=40, near=near, far=far, jitter=jit)
    assert torch.equal(hits.cpu(), g["an_bi_hits"]) and (pts.cpu() - g["an_bi_pts"]).abs().max() <= 5e-5
    assert march.load_intersection_kind("sphere") is march.sphere_march
    with pytest.raises(NotImplementedError):
        march.load_intersection_kind("secant")


@pytest.mark.gpu
def test_hip_marching_through_the_fused_siren_sdf():
    """SDF values now come from the fused MLP kernel (bf16x3: ~5e-6 off the fp32 reference), so a ray whose SDF
    grazes a threshold may decide differently: decisions must agree on >= 95 % of the rays and distances within 1e-3
    on the rays that agree."""
    from nerf_atlas_amd import march, config
    import nerf_atlas_amd.sdf as sdf
    g = load_golden("g15_march")
    config.set_precision("bf16x3")
    m = sdf.SIREN(intermediate_size=0).cuda().eval()
    sd = m.state_dict()
    for k, v in golden_params(g).items():
        sd[k].copy_(v)
    r_o, r_d, jit = g["r_o"].cuda(), g["r_d"].cuda(), float(g["jitter"])
    near, far = float(g["nn_near"]), float(g["nn_far"])
    pts, hits, dist, _ = march.sphere_march(m, r_o, r_d, iters=24, eps=1e-3, near=near, far=far)
    same = hits.cpu() == g["nn_sm_hits"]
    assert same.float().mean() >= 0.95
    assert ((dist.cpu() - g["nn_sm_dist"]).abs().squeeze(-1)[same] <= 1e-3).float().mean() >= 0.95
    pts, hits, _, tput = march.bisect(m, r_o, r_d, iters=40, near=near, far=far, jitter=jit)
    same = hits.cpu() == g["nn_bi_hits"]
    assert same.float().mean() >= 0.95
    assert (tput.cpu() - g["nn_bi_tput"]).abs().squeeze(-1)[same].median() <= 1e-4
    ok = (pts.cpu() - g["nn_bi_pts"]).abs().amax(-1) <= 1e-3
    assert ok.float().mean() >= 0.9


@pytest.mark.gpu
def test_hip_marching_disagreements_are_threshold_grazing():
    """Why the decisions through the fused MLP are not bit-identical to the golden, shown iteration by
    iteration with the oracle teacher-forced onto the GPU's own march state: (a) the fused bf16x3 SDF is within 2e-4 of the
    fp32 oracle SDF at the SAME points (measured 1.0e-4: SIREN values of several units, ~2e-5 relative); (b) given the SDF values the update kernels take exactly the reference's
    decisions; hence (c) every ray whose hit / stop decision differs from what fp32 arithmetic would decide at that
    state has |sdf - eps| (or |dist + sdf - far|) <= 2e-4 at the deciding iteration -- a threshold-grazing ray, not a
    logic error.  Rays are dropped from the audit after their first divergence (their states differ from then on)."""
    from nerf_atlas_amd import config, ops
    import nerf_atlas_amd.sdf as sdf
    g = load_golden("g15_march")
    config.set_precision("bf16x3")
    m = sdf.SIREN(intermediate_size=0).cuda().eval()
    sd = m.state_dict()
    for k, v in golden_params(g).items():
        sd[k].copy_(v)
    fn = siren_fn(g)
    r_o, r_d = g["r_o"].cuda().contiguous(), g["r_d"].cuda().contiguous()
    near, far, eps, tol = float(g["nn_near"]), float(g["nn_far"]), 1e-3, 2e-4
    batch = r_o.shape[:-1]
    dist = torch.full(batch + (1,), near, device="cuda")
    hits = torch.zeros(batch, device="cuda", dtype=torch.uint8)
    rem = torch.ones(batch, device="cuda", dtype=torch.uint8)
    audited = torch.ones(batch, dtype=torch.bool)
    grazing = torch.zeros(batch, dtype=torch.bool)
    worst_sdf_err = 0.0
    with torch.no_grad():
        for _ in range(24):
            pts = ops.ray_points(r_o, r_d, dist)
            s_gpu = m(pts).contiguous()
            s_ref = fn(pts.cpu())[..., 0]                        # fp32 oracle at the GPU's points
            worst_sdf_err = max(worst_sdf_err, float((s_gpu[..., 0].cpu() - s_ref).abs().max()))
            d0, h0, r0 = dist.cpu()[..., 0], hits.cpu().bool(), rem.cpu().bool()
            ops.sphere_march_update(s_gpu, eps, far, dist, hits, rem)
            # (b) the kernel's decisions are the reference's (src/march.py:39-45) for the SDF values it was given
            sg = s_gpu[..., 0].cpu()
            h_exp = h0 | (r0 & (sg < eps) & (d0 <= far))
            d_exp = torch.where(r0, d0 + sg, d0)
            r_exp = r0 & ~(h_exp | (d_exp > far))
            assert torch.equal(hits.cpu().bool(), h_exp) and torch.equal(rem.cpu().bool(), r_exp)
            assert torch.equal(dist.cpu()[..., 0], d_exp)
            # (c) what fp32 SDF values would have decided at the same state
            h_ref = h0 | (r0 & (s_ref < eps) & (d0 <= far))
            r_ref = r0 & ~(h_ref | (torch.where(r0, d0 + s_ref, d0) > far))
            differ = audited & ((h_ref != h_exp) | (r_ref != r_exp))
            margin = torch.minimum((s_ref - eps).abs(), (d0 + s_ref - far).abs())
            assert bool((margin[differ] <= tol).all()), float(margin[differ].max())
            grazing |= differ
            audited &= ~differ
    assert worst_sdf_err <= tol, worst_sdf_err
    final_differs = hits.cpu().bool() != g["nn_sm_hits"]
    # every ray whose final decision differs from the golden diverged at a grazing threshold (or never diverged from the
    # teacher-forced fp32 decisions at all, in which case the golden's own trajectory drifted by the accumulated 1e-5s)
    print(f"\n[march] fused-vs-fp32 SDF error {worst_sdf_err:.2e}; {int(grazing.sum())} grazing rays, "
          f"{int(final_differs.sum())} of {final_differs.numel()} final decisions differ from the golden")
    assert float(final_differs.float().mean()) <= 0.05


@pytest.mark.gpu
def test_compacted_marching_equals_dense_and_skips_finished_rays():
    """the compacted path (na_compact_rays -> SDF on the live rays -> indexed update; what the reference's
    boolean-mask indexing does, src/march.py:37-45, :164-179) gives the dense path's hits / distances / points BIT FOR BIT --
    on the g15 fixture rays (analytic SDF: also equal to the reference's own outputs; SIREN SDF through the fused MLP) and on
    a 160 x 160 frame -- and evaluates 2.5x fewer network rows on the fixture scene (printed; more on full frames)."""
    from nerf_atlas_amd import march, ops, config
    import nerf_atlas_amd.sdf as sdf
    g = load_golden("g15_march")
    config.set_precision("bf16x3")
    m = sdf.SIREN(intermediate_size=0).cuda().eval()
    sd = m.state_dict()
    for k, v in golden_params(g).items():
        sd[k].copy_(v)
    jit = float(g["jitter"])
    # ordered compaction: ascending indices, exact count, empty and full masks
    for n, p in ((1, 0.5), (255, 0.3), (4097, 0.9), (100000, 0.02), (640000, 0.5)):
        torch.manual_seed(n)
        live = (torch.rand(n, device="cuda") < p).to(torch.uint8)
        idx = torch.empty(n + 256, device="cuda", dtype=torch.int32)
        cnt = torch.empty(1, device="cuda", dtype=torch.int32)
        k = ops.compact_rays(live, idx, cnt)
        assert k == int(live.sum()) and torch.equal(idx[:k].long(), live.nonzero().flatten())
    z = torch.zeros(5000, device="cuda", dtype=torch.uint8)
    assert ops.compact_rays(z, torch.empty(5256, device="cuda", dtype=torch.int32), torch.empty(1, device="cuda", dtype=torch.int32)) == 0
    # (a) the fixture rays, both SDFs, compaction forced
    for tag, fn in (("an", analytic), ("nn", m)):
        r_o, r_d = g["r_o"].cuda(), g["r_d"].cuda()
        near, far = float(g[f"{tag}_near"]), float(g[f"{tag}_far"])
        dense = march.sphere_march(fn, r_o, r_d, iters=24, eps=1e-3, near=near, far=far, compact=False)
        comp = march.sphere_march(fn, r_o, r_d, iters=24, eps=1e-3, near=near, far=far, compact=True)
        st = dict(march.last_stats)
        assert torch.equal(dense[0], comp[0]) and torch.equal(dense[1], comp[1]) and torch.equal(dense[2], comp[2])
        print(f"\\n[march/{tag}] sphere_march: {st['mlp_rows']} network rows against {st['dense_rows']} dense "
              f"({st['dense_rows'] / st['mlp_rows']:.1f}x fewer), {st['iters']} of 24 iterations ran")
        # (measured: 2.5x on the analytic fixture -- a few grazing rays march through all 24 iterations -- and more on frames)
        assert st["dense_rows"] >= 2 * st["mlp_rows"] if tag == "an" else st["mlp_rows"] < st["dense_rows"], st
        if tag == "an":
            assert torch.equal(comp[1].cpu(), g["an_sm_hits"])  # and it is the reference's answer
        d2 = march.bisect(fn, r_o, r_d, iters=40, near=near, far=far, jitter=jit, compact=False)
        c2 = march.bisect(fn, r_o, r_d, iters=40, near=near, far=far, jitter=jit, compact=True)
        stb = dict(march.last_stats)
        assert all(torch.equal(a, b) for a, b in zip(d2, c2))
        print(f"[march/{tag}] bisection: {stb['mlp_rows']} network rows against {stb['dense_rows']} dense")
        assert stb["mlp_rows"] < stb["dense_rows"]
    # (b) a frame above COMPACT_MIN_RAYS: the default picks the compacted path and the answer is the dense one
    import math
    from nerf_atlas_amd import cameras
    size = 160
    cam = cameras.NeRFCamera(cam_to_world=torch.tensor([[[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]]]),
                             focal=0.5 * size / math.tan(0.5 * 0.6911)).cuda()
    rays = cam.sample_positions((0, 0, size, size), size=size, with_noise=False)[0]
    r_o, r_d = rays[..., :3].contiguous(), torch.nn.functional.normalize(rays[..., 3:], dim=-1)
    assert r_o.numel() // 3 >= march.COMPACT_MIN_RAYS
    for fn in (analytic, m):
        dense = march.sphere_march(fn, r_o, r_d, iters=32, eps=1e-3, near=1.0, far=6.0, compact=False)
        auto = march.sphere_march(fn, r_o, r_d, iters=32, eps=1e-3, near=1.0, far=6.0)
        st = dict(march.last_stats)
        assert st["mlp_rows"] < st["dense_rows"] and all(torch.equal(a, b) for a, b in zip(dense[:3], auto[:3]))
        print(f"[march/frame] {st['dense_rows'] / st['mlp_rows']:.1f}x fewer network rows, hit fraction {float(auto[1].float().mean()):.2f}")
