"""SDF marching (src/march.py:10-203) with the reference's function protocol

    fn(sdf_model, r_o, r_d, iters, eps, near, far) -> (pts, hits, dist | best_pos, None | throughput)

`sdf_model` is any callable pts[..., 3] -> [..., >=1] whose column 0 is the signed distance (the reference passes
`SDF.underlying`).  Per-ray state stays on the device.  Two equivalent paths (identical hits / distances: the SDF network
evaluates every row independently of its neighbours):
  * dense (`compact=False`): every iteration is one SDF evaluation for ALL rays (the fused MLP kernel when `sdf_model` is one
    of this package's SDF models) plus one elementwise HIP update (csrc/march.hip) that leaves inactive rays untouched -- no
    gather, no host synchronisation;
  * compacted (`compact=True`, the default above `COMPACT_MIN_RAYS` rays): like the reference, which indexes its state with
    boolean masks (src/march.py:37-45, :164-179), only the rays that are still marching go through the network:
    `na_compact_rays` (ordered, deterministic) -> gather of their positions -> SDF on n rows -> indexed update.  One host
    read of the live count per iteration (where the reference's mask indexing synchronises too); the loop ends when no ray
    is left.  `last_stats` records how many network rows the last call evaluated against the dense count.
Inference only (the reference also runs these under no_grad).
"""
import random

import torch

from . import ops


def load_intersection_kind(kind):
    """src/march.py:10-20."""
    if kind == "sphere": return sphere_march
    if kind == "bisect": return bisect
    if kind == "secant":
        raise NotImplementedError("secant marching is marked broken in the reference (src/march.py:112-113); use bisect")
    raise NotImplementedError(f"unknown intersection kind {kind}")


COMPACT_MIN_RAYS = 4096   # below this a dense iteration is cheaper than the compaction's launches + host read
last_stats = {"mlp_rows": 0, "dense_rows": 0, "iters": 0}


def _sdf(model, pts):
    out = model(pts)
    return out if out.is_contiguous() else out.contiguous()


def _use_compact(compact, n_rays: int) -> bool:
    return n_rays >= COMPACT_MIN_RAYS if compact is None else bool(compact)


@torch.no_grad()
def sphere_march(self, r_o, r_d, iters: int = 32, eps: float = 1e-3, near: float = 0, far: float = 1, compact=None):
    """src/march.py:27-47 -> (pts, hits [...], dist [..., 1], None)."""
    r_o, r_d = r_o.contiguous(), r_d.contiguous()
    batch = r_o.shape[:-1]
    dist = torch.full(batch + (1,), float(near), device=r_o.device, dtype=torch.float32)
    hits = torch.zeros(batch, device=r_o.device, dtype=torch.uint8)
    rem = torch.ones(batch, device=r_o.device, dtype=torch.uint8)
    R = rem.numel()
    rows = done = 0
    if _use_compact(compact, R):
        idx = torch.empty(R + 256, device=r_o.device, dtype=torch.int32)
        cnt = torch.empty(1, device=r_o.device, dtype=torch.int32)
        for _ in range(iters):
            n = R if done == 0 else ops.compact_rays(rem, idx, cnt)   # (first iteration: every ray is live)
            if n == 0:
                break  # (the reference keeps iterating over empty index sets: no-ops)
            if n == R and done == 0:
                ops.sphere_march_update(_sdf(self, ops.ray_points(r_o, r_d, dist)), eps, far, dist, hits, rem)
            else:
                ops.sphere_march_update_indexed(_sdf(self, ops.ray_points_indexed(r_o, r_d, dist, idx, n)), idx, n, eps, far,
                                                dist, hits, rem)
            rows += n
            done += 1
    else:
        for _ in range(iters):
            ops.sphere_march_update(_sdf(self, ops.ray_points(r_o, r_d, dist)), eps, far, dist, hits, rem)
        rows, done = R * iters, iters
    last_stats.update(mlp_rows=rows, dense_rows=R * iters, iters=done)
    return ops.ray_points(r_o, r_d, dist), hits.bool(), dist, None


@torch.no_grad()
def throughput_with_sign_change(self, r_o, r_d, near: float, far: float, batch_size: int = 128, jitter=None):
    """src/march.py:78-110 -> (sdf at the closest sample, that position, last_pos, first_neg); last_pos / first_neg are
    step offsets times the step (the reference does not add `near` back).  `jitter` replaces the reference's
    random.random() draw; the very first probe is `r_o + near` (sic: the scalar is added to the origin)."""
    r_o, r_d = r_o.contiguous(), r_d.contiguous()
    batch = r_o.shape[:-1]
    dev = r_o.device
    j = random.random() if jitter is None else float(jitter)
    max_t = far - near + j * (2 / batch_size)
    step = max_t / batch_size
    curr_min = _sdf(self, ops.ray_points(r_o, torch.ones_like(r_o), float(near)))[..., 0].contiguous()
    idxs = torch.zeros(batch, device=dev, dtype=torch.int32)
    last_pos = torch.full(batch, -1, device=dev, dtype=torch.int32)
    first_neg = torch.full(batch, -1, device=dev, dtype=torch.int32)
    for i in range(batch_size):
        t = near + step * (i + 1)
        ops.sign_change_update(_sdf(self, ops.ray_points(r_o, r_d, t)), i, curr_min, idxs, last_pos, first_neg)
    best_t = near + idxs.unsqueeze(-1) * step
    best_pos = ops.ray_points(r_o, r_d, best_t.float())
    val = _sdf(self, best_pos)
    return val[..., 0], best_pos, last_pos.unsqueeze(-1) * step, first_neg.unsqueeze(-1) * step


@torch.no_grad()
def bisection(self, r_o, r_d, near, far, iters: int = 32, eps: float = 1e-6, compact=None):
    """src/march.py:147-180; near/far per-ray tensors [..., 1] (updated in place like the reference's)."""
    r_o, r_d = r_o.contiguous(), r_d.contiguous()
    low = near if near.dtype == torch.float32 and near.is_contiguous() else near.float().contiguous()
    high = far if far.dtype == torch.float32 and far.is_contiguous() else far.float().contiguous()
    assert bool((high >= low).all())
    sdf_low = _sdf(self, ops.ray_points(r_o, r_d, low))[..., 0, None].contiguous()
    sdf_high = _sdf(self, ops.ray_points(r_o, r_d, high))[..., 0, None].contiguous()
    z = torch.empty_like(low)
    todo = torch.empty(low.shape, device=low.device, dtype=torch.uint8)
    ops.bisection_update(None, eps, low, high, sdf_low, sdf_high, z, todo)
    R = todo.numel()
    rows = 2 * R
    done = 0
    if _use_compact(compact, R):
        idx = torch.empty(R + 256, device=low.device, dtype=torch.int32)
        cnt = torch.empty(1, device=low.device, dtype=torch.int32)
        for i in range(iters):
            n = ops.compact_rays(todo, idx, cnt)
            if n == 0:
                break  # (`if not todo.any(): break` of the reference)
            ops.bisection_update_indexed(_sdf(self, ops.ray_points_indexed(r_o, r_d, z, idx, n)), idx, n, eps, low, high, sdf_low,
                                         sdf_high, z, todo)
            rows += n
            done += 1
    else:
        for i in range(iters):
            if i % 8 == 0 and not bool(todo.any()):
                break  # the reference checks every iteration; converged rays are no-ops, so checking every 8th is equivalent
            ops.bisection_update(_sdf(self, ops.ray_points(r_o, r_d, z)), eps, low, high, sdf_low, sdf_high, z, todo)
            rows += R
            done += 1
    last_stats.update(mlp_rows=rows, dense_rows=(2 + iters) * R, iters=done)
    return ops.ray_points(r_o, r_d, z)


@torch.no_grad()
def bisect(self, r_o, r_d, iters: int = 128, eps: float = 0, near: float = 0, far: float = 1, jitter=None, compact=None):
    """src/march.py:63-75 -> (pts, hits, best_pos, throughput [..., 1])."""
    tput, best_pos, last_pos, first_neg = throughput_with_sign_change(self, r_o, r_d, near=near, far=far,
                                                                      batch_size=iters, jitter=jitter)
    pts = bisection(self, r_o, r_d, near=last_pos, far=first_neg, iters=min(32, iters), compact=compact)
    return pts, tput < 0, best_pos, tput.unsqueeze(-1)
