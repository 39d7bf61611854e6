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
    w = nerf.weights
    ts_ray = getattr(nerf, "ts_ray", None)
    if ts_ray is not None and ts_ray.shape[-1] == w.shape[0] and tuple(ts_ray.shape[:-1]) == tuple(w.shape[1:]):
        # coarse -> fine (PlainNeRF.forward_coarse_fine): the weights' rows are the per-ray steps [*batch, T + N]
        return ops.integrate(w, ts_ray.movedim(-1, 0).unsqueeze(-1).contiguous())
    assert nerf.ts.dim() == 1 and nerf.ts.shape[0] == w.shape[0], "weights do not belong to the shared steps model.ts"
    ts = nerf.ts[:, None, None, None, None].expand(w.shape + (1,)).contiguous()
    return ops.integrate(w, ts)


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


def rigidity_map(model) -> torch.Tensor:
    """runner.py:911-913: volumetric_integrate(weights, rigidity) of a dynamic model's last forward."""
    from . import ops
    return ops.integrate(model.nerf.weights, model.rigidity.contiguous())


def depth_to_normals(depth_img: torch.Tensor) -> torch.Tensor:
    """src/utils.py:421-427: forward differences of a depth image [H,W,1] -> unit normals [H-1,W-1,3] (image-space
    post-processing of a finished frame, not a per-sample operator)."""
    dz_dx = depth_img[1:, 1:, ...] - depth_img[:-1, 1:, ...]
    dz_dy = depth_img[1:, 1:, ...] - depth_img[1:, :-1, ...]
    d = torch.cat([dz_dx / 2, dz_dy / 2, torch.ones_like(dz_dx)], dim=-1)
    return torch.nn.functional.normalize(d, dim=-1)


def depth_vis(model, near: float, far: float, normals_from_depth: bool = False):
    """runner.py:511-519 (`--visualize depth`): [normalised depth of batch item 0 (, its normal map)] of the last forward.
    The reference line reads `(raw_depth[0]-args.near)/(args.far - args.near).clamp(min=0, max=1)`: the clamp binds to the
    denominator and does not exist for the float arguments the CLI passes; this follows the evident intent
    ((raw - near) / (far - near), clamped to [0, 1]) -- DESIGN section 8."""
    depth = ((depth_map(model)[0] - near) / (far - near)).clamp(min=0, max=1)
    items = [depth]
    if normals_from_depth:
        items.append(((50 * depth_to_normals(depth) + 1) / 2).clamp(min=0, max=1))
    return items


def flow_vis(model):
    """runner.py:521-526 (`--visualize flow`): integrated rigid flow of batch item 0, normalised by its largest vector norm,
    signed square root, mapped to [0, 1]."""
    if not hasattr(model, "rigid_dp"): return []
    flow = flow_map(model)[0]
    flow = flow / flow.norm(dim=-1).max()
    flow = flow.abs().sqrt().copysign(flow)
    return [(flow + 1) / 2]


def rigidity_vis(model):
    """runner.py:528-531 (`--visualize rigidity`)."""
    if not hasattr(model, "rigidity"): return []
    return [rigidity_map(model)[0]]


# runner.py:534-538
visualizations = {"depth": depth_vis, "flow": flow_vis, "rigidity": rigidity_vis}


def render_over_time(model, cam, size: int, crop_size: int, times, with_alpha: bool = False, rank: int = 0,
                     world: int = 1):
    """runner.py:998-1017: one camera, a sweep of times through a dynamic model; frame i is rendered in test()-style
    tiles at times[i].  Frames are independent, so with world > 1 this rank renders frames rank, rank+world, ...
    (D-NeRF time sweeps shard by frame); returns [(index, frame [size,size,3(+1)])]."""
    device = next(model.parameters()).device
    frames = []
    with torch.no_grad():
        for i in range(rank, len(times), world):
            t = times[i].reshape(1).to(device)
            got = torch.zeros(size, size, 3 + int(with_alpha), device=device)
            for (c0, c1, h, w) in tile_list(size, crop_size):
                o, _ = render(model, cam, (c0, c1, h, w), size=size, times=t, with_noise=False)
                got[c0:c0 + h, c1:c1 + w, :3] = o.squeeze(0)
                if with_alpha:
                    got[c0:c0 + h, c1:c1 + w, 3] = alpha_map(model)[0, ..., 0]
            frames.append((i, got))
    return frames


def psnr(got: torch.Tensor, exp: torch.Tensor) -> float:
    """runner.py:923-924."""
    return float(mse2psnr(torch.nn.functional.mse_loss(got, exp)))
