"""na_render_plain_mip_ls (render_ls.hip MODEL 6) at the C-ABI level: ragged step counts, the smallest crop, several crops in one
launch, stream / argument refusal, empty batch.  (The 800^2 tile, the tiled frame and the wide-band reproducibility live in
test_gpu_models.py, test_gpu_wholeframe.py and test_gpu_determinism.py.)"""
import math

import pytest
import torch

import oracle as O
from oracle.procedural import proc_param

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import ops
    from nerf_atlas_amd.utils import ConicGaussian
    m = nerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", mip=ConicGaussian()).cuda().eval()
    params = {}
    for k, v in m.state_dict().items():
        if k.endswith("primes") or v.numel() == 0:
            continue
        params[k] = torch.from_numpy(proc_param(k, tuple(v.shape)))
        v.copy_(params[k])
    return ops, m, params


def _rays(crop, B=1, size=800):
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]], [[0.8, -0.36, 0.48, 1.9], [0.0, 0.8, 0.6, 2.4], [-0.6, -0.48, 0.64, 2.6]]])[:B]
    return O.nerf_camera_rays(O.pixel_grid(size, crop), c2w, focal, size)


@pytest.mark.parametrize("T,crop,B", [(1, (10, 10, 2, 3), 1), (7, (400, 400, 2, 1), 1), (33, (0, 790, 5, 10), 2), (64, (397, 1, 3, 67), 2)])
def test_mip_ls_ragged_shapes_vs_oracle(setup, T, crop, B):
    ops, m, params = setup
    rays = _rays(crop, B)
    ts, _ = ops.compute_ts(2.0, 6.0, T, "cuda")
    packed = m.packed_mip_ls("f16x")
    out, alpha, w = ops.render_plain_mip_ls(rays.cuda(), ts, m.first.enc.tables(), packed, "f16x", "cone", float("nan"), 0, 16,
                                            "upshifted", "white", want_weights=True)
    aux = {}
    ref = O.plain_nerf(params, rays, 2.0, 6.0, T, "view", act="upshifted", bg="white", mip="cone", aux=aux) if T > 1 else None
    if T == 1:  # (the oracle closes the last interval at 2 ts[-1] - ts[-2]: needs two steps; one step closes at ts + 1 -- the kernel's rule)
        assert out.shape == (B, crop[2], crop[3], 3) and torch.isfinite(out).all()
        return
    assert float((out.cpu() - ref).abs().max()) <= 1e-4
    assert float((w.cpu() - aux["weights"]).abs().max()) <= 1e-4 and float((alpha.cpu() - aux["alpha"]).abs().max()) <= 1e-4


def test_mip_ls_refusals_and_empty_batch(setup):
    from nerf_atlas_amd._lib import NaError
    ops, m, _ = setup
    rays = _rays((100, 100, 4, 4)).cuda()
    ts, _ = ops.compute_ts(2.0, 6.0, 16, "cuda")
    tables, packed = m.first.enc.tables(), m.packed_mip_ls("f16x")
    good = ops.render_plain_mip_ls(rays, ts, tables, packed, "f16x", "cylinder", float("nan"), 0, 16, "upshifted", "black")[0]
    assert torch.isfinite(good).all()
    with pytest.raises(NaError):
        ops.render_plain_mip_ls(rays, ts, tables, packed, "bf16x3", "cylinder", float("nan"), 0, 16)    # f16x only
    with pytest.raises(NaError):
        ops.render_plain_mip_ls(rays, ts, tables, packed, "f16x", "cylinder", float("nan"), 0, 12)      # the schedule is built for 16 degrees
    with pytest.raises(NaError):
        ops.render_plain_mip_ls(rays[:, :1], ts, tables, packed, "f16x", "cylinder", float("nan"), 0, 16)  # H = 1: no pixel radius
    with pytest.raises(NaError):
        ops.render_plain_mip_ls(rays, ts, tables, packed, "f16x", "cylinder", float("nan"), 0, 16,
                                workspace=torch.empty(8, dtype=torch.uint8, device="cuda"))
    # a PlainNeRF (MODEL 0) f16x stream is not a mip stream: refused with NaN colours, never consumed
    import nerf_atlas_amd.nerf as nerf
    plain = nerf.PlainNeRF(steps=16, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").cuda().eval()
    other = plain.packed_ls("f16x")
    wrong = torch.cat([other, other])[: packed.numel()].contiguous()
    assert torch.isnan(ops.render_plain_mip_ls(rays, ts, tables, wrong, "f16x", "cylinder", float("nan"), 0, 16)[0]).all()
    out = ops.render_plain_mip_ls(rays[:0], ts, tables, packed, "f16x", "cylinder", float("nan"), 0, 16)[0]
    assert out.shape == (0, 4, 4, 3)
    # an explicit closing edge of the last interval instead of the default 2 ts[-1] - ts[-2]
    far = ops.render_plain_mip_ls(rays, ts, tables, packed, "f16x", "cylinder", 1e10, 0, 16, "upshifted", "black")[0]
    assert torch.isfinite(far).all() and not torch.equal(far, good)
