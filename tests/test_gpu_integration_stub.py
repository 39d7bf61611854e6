"""The ctypes stub printed in INTEGRATION.md section B (what a nerf_atlas maintainer would paste next to src/*.py) is
executed verbatim against modules with the reference's attribute names and compared with this package's own wrappers:
the documentation is code that runs."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_stub():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    src = [b for b in blocks if "hip_backend.py" in b][0]
    src = src.replace('C.CDLL("libnerf_atlas_amd.so")', f'C.CDLL("{os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so")}")')
    ns = {}
    exec(compile(src, "INTEGRATION.md:hip_backend", "exec"), ns)
    return ns


def test_documented_ctypes_stub_runs_and_matches():
    from nerf_atlas_amd import ops, config
    import nerf_atlas_amd.neural_blocks as nb
    ns = load_stub()
    torch.manual_seed(3)
    x = (torch.rand(1000, 3, device="cuda") * 4 - 2)
    enc = nb.HashEncoder().cuda()
    assert torch.equal(ns["hash_encoder_forward"](enc, x), ops.hash_encode(x, enc.tables(), True))
    # compositing
    T, R = 16, 50
    ts = torch.linspace(2, 6, T, device="cuda")
    rays = torch.randn(R, 6, device="cuda")
    density, rgb = torch.randn(T, R, device="cuda"), torch.rand(T, R, 3, device="cuda")
    a, w, out = ns["alpha_from_density_and_integrate"](density, rgb, ts, rays, softplus=True, white_bg=True)
    o2, a2, w2 = ops.composite(density, rgb, ts, rays, softplus=True, bg="white")
    assert torch.equal(out, o2) and torch.equal(a, a2) and torch.equal(w, w2)
    # fused MLP through the stub (precision 1 = bf16x3) vs the module's own forward
    config.set_precision("bf16x3")
    mlp = nb.SkipConnMLP(in_size=3, out=65, num_layers=4, hidden_size=256, enc=nb.HashEncoder()).cuda()
    with torch.no_grad():
        ref = mlp(x)
    got = ns["skip_conn_mlp_forward"](mlp, x, None, 1)
    assert torch.equal(got, ref)
    sir = nb.SkipConnMLP(in_size=5, out=3, latent_size=64, num_layers=4, hidden_size=256, init="siren", activation=torch.sin).cuda()
    p, lat = torch.randn(777, 5, device="cuda"), torch.randn(777, 64, device="cuda")
    with torch.no_grad():
        ref = sir(p, lat)
    assert torch.equal(ns["skip_conn_mlp_forward"](sir, p, lat, 1), ref)
