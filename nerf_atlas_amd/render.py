"""Frame rendering with the reference's tile order (runner.py:490-509 render(), :879-892 test() tile loop)."""
import math
from typing import Optional

import torch

from .utils import mse2psnr


def render(model, cam, crop, size, times=None, with_noise=0.1):
    """runner.py:490-509.  crop = (top, left, h, w) in pixels of a size x size image.  Returns (out, rays)."""
    rays = cam.sample_positions(tuple(crop), size=size, with_noise=with_noise)
    if times is not None:
        return model((rays, times)), rays
    return model(rays), rays


def tile_list(render_size: int, crop_size: int):
    """Tiles in the reference's order: x over rows, then y over columns; ragged last tiles are clipped
    (runner.py:879-892)."""
    if crop_size <= 0:
        crop_size = render_size
    n = math.ceil(render_size / crop_size)
    tiles = []
    for x in range(n):
        for y in range(n):
            c0, c1 = x * crop_size, y * crop_size
            tiles.append((c0, c1, min(crop_size, render_size - c0), min(crop_size, render_size - c1)))
    return tiles


def render_frame(model, cam, size: int, crop_size: int = 0, times=None, tiles=None, out: Optional[torch.Tensor] = None):
    """test()-style tiled frame: every tile is one render() with B=1 written into got[c0:c0+cs, c1:c1+cs].
    `tiles` restricts the work to a subset (multi-GPU sharding); untouched pixels stay zero."""
    device = next(model.parameters()).device
    got = out if out is not None else torch.zeros(size, size, 3, device=device)
    with torch.no_grad():  # runner.test() renders under no_grad: fused inference kernels
        for (c0, c1, h, w) in (tiles if tiles is not None else tile_list(size, crop_size)):
            o, _ = render(model, cam, (c0, c1, h, w), size=size, times=times, with_noise=False)
            got[c0:c0 + h, c1:c1 + w, :] = o.squeeze(0)
    return got


def depth_map(model) -> torch.Tensor:
    """runner.py:894-897: expected termination depth of the LAST forward, volumetric_integrate(weights, ts) -> [B,H,W,1]."""
    from . import ops
    nerf = model.nerf
    ts = nerf.ts[:, None, None, None, None].expand(nerf.weights.shape + (1,)).contiguous()
    return ops.integrate(nerf.weights, ts)


def alpha_map(model) -> torch.Tensor:
    """runner.py:523 acc_map: sum of weights[:-1] along the ray (the white-background complement)."""
    from . import ops
    w = model.nerf.weights
    ones = torch.ones(w.shape + (1,), device=w.device)
    ones[-1] = 0
    return ops.integrate(w, ones)


def flow_map(model) -> torch.Tensor:
    """runner.py:908-910: volumetric_integrate(weights, rigid_dp) of a dynamic model's last forward."""
    from . import ops
    return ops.integrate(model.nerf.weights, model.rigid_dp.contiguous())


def psnr(got: torch.Tensor, exp: torch.Tensor) -> float:
    """runner.py:923-924."""
    return float(mse2psnr(torch.nn.functional.mse_loss(got, exp)))
