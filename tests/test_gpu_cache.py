"""The packed-weight caches of the fused renderers against parameter writes (VERDICT r03 "weak" 7): what re-packs on its
own, what needs `invalidate_packed()`, and the always-repack switch for callers that write behind torch's back."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _setup():
    import nerf_atlas_amd.nerf as nerf
    from nerf_atlas_amd import cameras, config, render
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    size, T = 64, 32
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]])
    model = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev).eval()
    cam = cameras.NeRFCamera(cam_to_world=c2w, focal=focal).to(dev)
    crop = (24, 24, 16, 16)
    config.set_precision("f16x")
    draw = lambda: render.render(model, cam, crop, size, with_noise=False)[0].clone()
    return model, draw, config


@pytest.mark.parametrize("prec", ["f16x", "bf16x3"])
def test_parameter_writes_reach_the_fused_render(prec):
    model, draw, config = _setup()
    config.set_precision(prec)
    try:
        a = draw()
        assert torch.equal(a, draw())  # (cached stream, same bits)
        w = model.refl.mlp.out.weight
        # 1. an in-place op on the Parameter bumps its version counter: re-packed on its own
        w.mul_(2.0)
        b = draw()
        assert float((a - b).abs().max()) > 1e-3
        # 2. a write through .data is invisible to the stamp: the documented contract is invalidate_packed() ...
        w.data.mul_(0.5)
        model.invalidate_packed()
        c = draw()
        assert torch.equal(a, c), "w * 2 * 0.5 must render the first frame again"
        # 3. ... or the always-repack switch, under which the .data write alone is enough
        config.set_repack_always(True)
        w.data.mul_(2.0)
        d = draw()
        assert torch.equal(b, d)
        model.first.enc.embs[0].weight.data.mul_(4.0)  # the stacked hash tables are such a cache too
        e = draw()
        assert float((d - e).abs().max()) > 1e-4
        config.set_repack_always(False)
        # 4. load_state_dict (torch copies under no_grad, and the post hook drops the caches regardless)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        sd["refl.mlp.out.weight"] = sd["refl.mlp.out.weight"] * 0.5
        sd["first.enc.embs.0.weight"] = sd["first.enc.embs.0.weight"] * 0.25
        model.load_state_dict(sd)
        f = draw()
        assert torch.equal(a, f)  # (powers of two: exact)
    finally:
        config.set_repack_always(False)
        config.set_precision("bf16x3")
