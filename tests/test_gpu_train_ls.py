"""The one-launch training forward of PlainNeRF(view) (round 6: csrc/ls_kernel.h MODEL 9, na_train_plain_view_ls, PlainNeRF._train_forward_ls):
the output rows it leaves for the backward pass against an fp64 restatement of src/neural_blocks.py:279-296 on the same weights, and a whole
training step (loss, all 32 gradients) against the layer-by-layer forward (csrc/train_fwd.hip) -- the same three-product bf16 arithmetic with
another summation order.  Bars written here: rows within 3e-5 of the plane's largest value (the layer path: the same bar); loss within 1e-6
relative; every gradient within 2e-3 of its largest element (what two split-bf16 summation orders differ by; each path is held to the reference's
own loss curves by tests/test_gpu_train.py, which runs with the one-launch forward by default)."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def _model_and_rays(crop, steps, seed=3, batch=1):
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    m = nerf.PlainNeRF(steps=steps, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev)
    m.eval()
    size = 800
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]] * batch, device=dev)
    rays = ops.raygen(c2w, focal, size, ((size - crop) // 2, (size - crop) // 2, crop, crop))
    return m, rays


def _mlp_fp64(m, init, act):
    """src/neural_blocks.py:288-296: x = init Linear; every layer x = layer(act(cat([x, init]) if skip else x)); out(act(x)) -- the rows BEFORE act"""
    lins = m._linears()
    init = init.double()
    rows, x = [], init @ lins[0].weight.double().T + lins[0].bias.double()
    n = len(m.layers)
    for i, lin in enumerate(lins[1:]):
        rows.append(x)
        skip = i < n and i != n - 1 and (i % m.skip) == 0
        h = act(torch.cat([x, init], dim=-1) if skip else x)
        x = h @ lin.weight.double().T + lin.bias.double()
    return rows, x


@pytest.mark.parametrize("crop,steps", [(16, 64), (24, 48), (13, 80), (24, 16), (15, 40)])
def test_rows_of_the_one_launch_forward_vs_fp64(crop, steps):
    from nerf_atlas_amd import ops
    from nerf_atlas_amd.nerf import compute_pts_ts
    m, rays = _model_and_rays(crop, steps)
    pts, ts, r_o, r_d, _ = compute_pts_ts(rays, 2.0, 6.0, steps, perturb=0)
    with torch.no_grad():
        planes, vrows_ls, dens_ls, rgb_pre, out = ops.train_plain_view_ls(rays.reshape(-1, 6), ts, pts, m.first.enc.tables(), m.packed_ls("bf16x3"), "upshifted")
        N = pts.numel() // 3
        init = ops.hash_encode_rows(pts.reshape(-1, 3), m.first.enc.tables(), True, 1)
        rows1, fo = _mlp_fp64(m.first, init, torch.nn.functional.leaky_relu)
        dens, vrows = ops.plain_head_rows(fo.float().contiguous(), pts.reshape(-1, 3), r_d.reshape(-1, 3).contiguous())
        rows2, rgb = _mlp_fp64(m.refl.mlp, vrows, torch.sin)
        worst = 0.0
        for p, want in enumerate(rows1 + rows2):
            err = float((planes[p].double() - want).abs().max() / want.abs().max())
            worst = max(worst, err)
            assert err <= 3e-5, (p, err)
        # first.out leaves the kernel as its consumers' inputs: density [N] and the View MLP's init rows [x, y, z, elev, azim | intermediate]
        assert torch.equal(vrows_ls[:, :5], vrows[:, :5])  # (the sample's position and the ray's angles: copies)
        assert float((vrows_ls[:, 5:].double() - fo[:, 1:]).abs().max() / fo.abs().max()) <= 3e-5
        assert float((dens_ls.double() - fo[:, 0]).abs().max() / fo.abs().max()) <= 3e-5
        assert float((rgb_pre.double() - rgb).abs().max() / rgb.abs().max().clamp_min(1.0)) <= 3e-5
        # the kernel's own composited colour = the inference renderer's (same schedule, same stream)
        ref, _, _ = ops.render_plain_view_ls(rays, ts, m.first.enc.tables(), m.packed_ls("bf16x3"), "bf16x3", "upshifted", "black", False, pts=pts)
        assert torch.equal(out.reshape(ref.shape), ref)
    print(f"crop {crop} x {steps}: N = {N}, worst plane error {worst:.2e} of the plane's maximum")


def _step(m, rays, target, kind):
    from nerf_atlas_amd import config
    config.set_train_forward(kind)
    try:
        for p in m.parameters():
            p.grad = None
        loss = torch.nn.functional.mse_loss(m(rays), target)
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        config.set_train_forward("ls")


@pytest.mark.parametrize("crop,steps,training", [(32, 64, False), (24, 48, True)])
def test_training_step_matches_the_layer_by_layer_forward(crop, steps, training):
    from nerf_atlas_amd import utils
    m, rays = _model_and_rays(crop, steps, batch=2 if training else 1)
    if training:
        m.train()
    target = torch.rand(rays.shape[:-1] + (3,), device=rays.device)
    res = {}
    for kind in ("layers", "ls"):
        torch.manual_seed(11)  # (training mode: the same perturbation and density noise in both runs)
        res[kind] = _step(m, rays, target, kind)
    (l0, g0), (l1, g1) = res["layers"], res["ls"]
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert g0.keys() == g1.keys() and len(g0) >= 32
    worst = 0.0
    for n in g0:
        err = float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30))
        worst = max(worst, err)
        assert err <= 2e-3, (n, err)
    print(f"loss {l0:.9f} / {l1:.9f}; worst gradient difference {worst:.2e} of the tensor's maximum")


def test_shapes_outside_the_kernel_take_the_layer_path():
    """fewer than 8 192 samples: the layer-by-layer forward (and its K-staged kernels) -- no error, the same model"""
    m, rays = _model_and_rays(8, 32)
    target = torch.rand(rays.shape[:-1] + (3,), device=rays.device)
    pre = m._train_forward_ls(rays, torch.linspace(2, 6, 32, device=rays.device), torch.zeros(32, 1, 8, 8, 3, device=rays.device), rays[..., 3:])
    assert pre is None
    l, g = _step(m, rays, target, "ls")
    assert math.isfinite(l) and len(g) >= 32


def test_one_launch_adam_invalidates_the_packed_streams():
    """train.NaAdam writes the parameters through raw pointers: it must bump their version counters like torch's own in-place update, or a
    validation render between two training steps (the fused renderers cache their packed weight streams per version) keeps showing the
    weights of its first call -- and the one-launch training forward would train on them."""
    import types
    from nerf_atlas_amd import train, utils
    m, rays = _model_and_rays(16, 64)
    opt = train.load_optim(types.SimpleNamespace(opt_kind="adam", learning_rate=1e-2, decay=0), m.parameters())
    target = torch.rand(rays.shape[:-1] + (3,), device=rays.device)
    with torch.no_grad():
        before = m(rays).clone()  # (packs the inference stream)
    v0 = m.first.init.weight._version
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(m(rays), target).backward()
        opt.step()
    assert m.first.init.weight._version > v0
    with torch.no_grad():
        after = m(rays).clone()
        utils.invalidate_packed(m)
        fresh = m(rays).clone()
    assert torch.equal(after, fresh)
    assert float((after - before).abs().max()) > 1e-3


def test_dnerf_step_through_the_one_launch_forward():
    """D-NeRF (src/nerf.py:1250-1303): the canonical PlainNeRF's points are the spline-warped ones and carry a gradient back into the
    deformation network -- the one-launch forward takes them as explicit positions, the backward nodes (hash input gradient, PlainHeadFn's
    point columns) are the layer path's: loss equal, every gradient incl. the deformation network's within 2e-3 of its maximum."""
    import nerf_atlas_amd.nerf as nerf
    m0, rays = _model_and_rays(16, 64, batch=2)
    torch.manual_seed(5)
    m = nerf.DynamicNeRF(canonical=m0, spline=4).to(rays.device).eval()
    with torch.no_grad():
        m.delta_estim.out.weight.normal_(0, 0.02)  # (the reference's zero initialisation would leave the warp at the identity)
    times = torch.tensor([0.3, 0.8], device=rays.device)
    target = torch.rand(rays.shape[:-1] + (3,), device=rays.device)
    res = {}
    from nerf_atlas_amd import config
    for kind in ("layers", "ls"):
        config.set_train_forward(kind)
        try:
            m.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(m((rays, times)), target)
            loss.backward()
            res[kind] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            config.set_train_forward("ls")
    (l0, g0), (l1, g1) = res["layers"], res["ls"]
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert g0.keys() == g1.keys() and any(n.startswith("delta_estim.") for n in g0)
    for n in g0:
        err = float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30))
        assert err <= 2e-3, (n, err)
    assert float(g1["delta_estim.init.weight"].abs().max()) > 0
