"""PyTorch-CPU fp32 restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Every function cites the reference (JulianKnodt/nerf_atlas) file:line it follows.
It is checked against fixtures produced by the real reference (tests/golden/, made by
tools/gen_golden.py) in tests/test_oracle_golden.py -> the oracle is *pinned*.

Nothing in nerf_atlas_amd/ imports this module.  It exists so that (a) the GPU parity
tests have a same-box checker (the reference's Python cannot travel to the GPU box) and
(b) bench.py can time a CPU baseline ("port") next to the HIP path.
"""
import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

__all__ = [
    "pixel_grid", "nerf_camera_rays", "dtu_camera_rays", "compute_ts", "compute_pts",
    "cumuprod_exclusive", "alpha_from_density", "volumetric_integrate", "sky_white",
    "hash_resolutions", "hash_corner_indices", "hash_encode", "fourier_encode",
    "positional_encode", "skip_mlp", "mlp_linear_shapes", "dir_to_elev_azim", "sigmoid",
    "expected_sin", "integrated_pos_enc_diag", "radii_x", "cylinder_moments",
    "cone_moments", "lift_gaussian_intended", "mip_latent_intended", "de_casteljau",
    "cubic_bezier", "laplace_cdf", "tiny_nerf", "plain_nerf", "plain_nerf_from_pts", "volsdf", "dynamic_nerf_spline",
    "view_refl", "positional_refl", "pos_linear_view_refl", "mse2psnr", "render_tiled",
    "HASH_PRIMES", "sphere_march", "throughput_with_sign_change", "bisection", "bisect",
    "point_light", "intersect_mask", "occlusion", "div_approx", "dnerf_rigid_dp", "ffjord_div",
    "sky_random", "depth_to_normals", "depth_vis", "flow_vis", "rigidity_vis",
    "sample_pdf_intended", "merge_ts_intended", "plain_nerf_rayts", "linspace01_f32",
]

# ----------------------------------------------------------------------------- A1 pixels


def pixel_grid(size: int, crop=None) -> torch.Tensor:
    """runner.py:490-503: positions[r, c] = (u=c, v=r) as float, then crop rows t:t+h, cols l:l+w."""
    ii, jj = torch.meshgrid(
        torch.arange(size, dtype=torch.float), torch.arange(size, dtype=torch.float), indexing="ij")
    positions = torch.stack([ii.transpose(-1, -2), jj.transpose(-1, -2)], dim=-1)
    if crop is not None:
        t, l, h, w = crop
        positions = positions[t:t + h, l:l + w, :]
    return positions


# ----------------------------------------------------------------------------- A2 cameras


def nerf_camera_rays(positions, c2w, focal: float, size: int, noise=None, with_noise: float = 0.0):
    """src/cameras.py:45-66 NeRFCamera.sample_positions.

    `noise` ([H,W,2] uniform [0,1) draws for (u, v)) replaces the reference's global-RNG
    rand_like so that jittered rays are reproducible.
    """
    u, v = positions.split([1, 1], dim=-1)
    if with_noise and noise is not None:
        u = u + (noise[..., 0:1] - 0.5) * with_noise
        v = v + (noise[..., 1:2] - 0.5) * with_noise
    d = torch.stack([(u - size * 0.5) / focal, -(v - size * 0.5) / focal, -torch.ones_like(u)], dim=-1)
    r_d = torch.sum(d[..., None, :] * c2w[..., :3, :3], dim=-1)
    r_d = r_d.permute(2, 0, 1, 3)
    r_o = c2w[..., :3, -1][:, None, None, :].expand_as(r_d)
    return torch.cat([r_o, r_d], dim=-1)


def _dtu_lift(x, y, z, intrinsics):
    """src/cameras.py:159-174 lift."""
    shape = x.shape
    fx = intrinsics[..., 0, 0, None].expand(shape)
    fy = intrinsics[..., 1, 1, None].expand(shape)
    cx = intrinsics[..., 0, 2, None].expand(shape)
    cy = intrinsics[..., 1, 2, None].expand(shape)
    sk = intrinsics[..., 0, 1, None].expand(shape)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack([x_lift, y_lift, z, torch.ones_like(z)], dim=-1)


def dtu_camera_rays(positions, pose, intrinsic, size: int):
    """src/cameras.py:190-223 DTUCamera.sample_positions (pose-matrix branch)."""
    r_o = pose[:, :3, 3]
    W, H, _ = positions.shape
    N = pose.shape[0]
    normalize = torch.tensor([1600, 1200], dtype=torch.float) / size
    u, v = (positions * normalize).reshape(-1, 2).split([1, 1], dim=-1)
    u = u.reshape(1, -1).expand(N, -1)
    v = v.reshape(1, -1).expand(N, -1)
    points = _dtu_lift(u, v, torch.ones_like(u), intrinsic)
    world = torch.bmm(pose, points.permute(0, 2, 1)).permute(0, 2, 1)[..., :3]
    r_o = r_o[:, None, :].expand_as(world)
    r_d = F.normalize(world - r_o, dim=-1)
    return torch.cat([r_o, r_d], dim=-1).reshape(N, W, H, 6)


# ----------------------------------------------------------------------------- A3 sampling


def compute_ts(near: float, far: float, steps: int, lindisp: bool = False, perturb: float = 0.0,
               rand: Optional[torch.Tensor] = None):
    """src/nerf.py:29-47 compute_ts.  `rand` [T] replaces torch.rand_like (Q7/Q13)."""
    if lindisp:
        t_vals = torch.linspace(0, 1, steps, dtype=torch.float)
        ts = 1 / (1 / max(near, 1e-10) * (1 - t_vals) + 1 / far * t_vals)
    else:
        ts = torch.linspace(near, far, steps=steps, dtype=torch.float)
    mids = None
    if perturb > 0:
        mids = 0.5 * (ts[:-1] + ts[1:])
        lower = torch.cat([mids, ts[-1:]])
        upper = torch.cat([ts[:1], mids])
        ts = lower + (upper - lower) * (rand * perturb)
    return ts, mids


def compute_pts(r_o, r_d, ts):
    """src/nerf.py:50-55: pts[T,...,3] = r_o[None] + ts (x) r_d."""
    return r_o.unsqueeze(0) + torch.tensordot(ts, r_d, dims=0)


# ----------------------------------------------------------------------------- A8 compositing


def cumuprod_exclusive(t):
    """src/nerf.py:22-27."""
    cp = torch.cumprod(t, dim=0)
    cp = torch.roll(cp, 1, dims=0)
    cp[0, ...] = 1.0
    return cp


def alpha_from_density(density, ts, r_d, softplus: bool = True):
    """src/nerf.py:60-73 (Q1,Q3,Q5,Q6)."""
    sigma_a = F.softplus(density - 1) if softplus else F.relu(density)
    end_val = torch.full_like(ts[..., :1], 1e10)
    dists = torch.cat([ts[..., 1:] - ts[..., :-1], end_val], dim=-1).clamp(min=1e-5)
    while len(dists.shape) < 4:
        dists = dists[..., None]
    dists = dists * torch.linalg.norm(r_d, dim=-1)
    alpha = 1 - torch.exp(-sigma_a * dists)
    weights = alpha * cumuprod_exclusive(1.0 - alpha + 1e-10)
    return alpha, weights


def volumetric_integrate(weights, other):
    """src/nerf.py:79-80."""
    return torch.sum(weights[..., None] * other, dim=0)


def sky_white(weights):
    """src/nerf.py:98 (Q4)."""
    return 1 - weights[:-1].sum(dim=0).unsqueeze(-1)


def sky_random(weights, rand):
    """src/nerf.py:101-103 random_color: one uniform draw per ray (`rand_like` of the [..., 1] remainder, broadcast over the
    colour channels) times the remainder.  `rand` is that draw, passed in (Q13: stochastic inputs are explicit)."""
    return rand * (1 - weights[:-1].sum(dim=0).unsqueeze(-1))


# ----------------------------------------------------------------------------- A5 hash encoder

HASH_PRIMES = (1, 2654435761, 805459861)


def hash_resolutions(levels: int = 8, low: int = 16, high: int = 1 << 14):
    """src/neural_blocks.py:126-128,146 (Q8): N_l = low * scale**i as a *Python double*,
    scale = exp((ln high - ln low)/levels - 1)."""
    scale = math.exp((math.log(high) - math.log(low)) / levels - 1)
    return [low * (scale ** i) for i in range(levels)]


def _hash_fn(v):
    """src/neural_blocks.py:135-139 (int64 multiply, xor)."""
    primes = torch.tensor(HASH_PRIMES, dtype=torch.long)
    vs = (v * primes).split(1, dim=-1)
    out = vs[0]
    for w in vs[1:]:
        out = out.bitwise_xor(w)
    return out


def _corners(l):
    """src/neural_blocks.py:149-165 corner order."""
    lx, ly, lz = l.split([1, 1, 1], dim=-1)
    h = l + 1
    hx, hy, hz = h.split([1, 1, 1], dim=-1)
    cat = lambda a, b, c: torch.cat([a, b, c], dim=-1)
    return [l, cat(lx, ly, hz), cat(lx, hy, lz), cat(lx, hy, hz),
            cat(hx, ly, lz), cat(hx, ly, hz), cat(hx, hy, lz), h]


def hash_corner_indices(x, levels: int = 8, emb_size: int = 1 << 16):
    """Table indices [levels, 8, N] (int64), src/neural_blocks.py:143-166."""
    out = []
    for N_l in hash_resolutions(levels):
        l = (x * N_l).floor().long()
        out.append(torch.stack([(_hash_fn(v) % emb_size).squeeze(-1) for v in _corners(l)], dim=0))
    return torch.stack(out, dim=0)


def hash_encode(x, tables: Sequence[torch.Tensor], include_input: bool = True, emb_size: int = 1 << 16):
    """src/neural_blocks.py:139-193 HashEncoder.forward.  x [N,3]; tables: 8 x [65536,4]."""
    out = []
    res = hash_resolutions(len(tables))
    for i, emb in enumerate(tables):
        v_l = x * res[i]
        l = v_l.floor().long()
        embs = torch.stack([emb[(_hash_fn(v) % emb_size).squeeze(-1)] for v in _corners(l)], dim=0)
        ws = v_l - l
        wx, wy, wz = ws.split([1, 1, 1], dim=-1)
        iwx, iwy, iwz = (1 - ws).split([1, 1, 1], dim=-1)
        weights = torch.stack([
            iwx * iwy * iwz, iwx * iwy * wz, iwx * wy * iwz, iwx * wy * wz,
            wx * iwy * iwz, wx * iwy * wz, wx * wy * iwz, wx * wy * wz], dim=0)
        out.append((embs * weights).sum(dim=0))
    out = torch.cat(out, dim=-1)
    if include_input:
        out = torch.cat([x, out], dim=-1)
    return out


# ----------------------------------------------------------------------------- A5' Fourier


def fourier_encode(x, basis, extra_scale: float = 1.0):
    """src/utils.py:14-17 fourier + src/neural_blocks.py:52; basis [D,F]."""
    mapped = x @ (extra_scale * basis)
    return torch.cat([mapped.sin(), mapped.cos()], dim=-1)


def positional_encode(x, bands):
    """src/neural_blocks.py:30-34 PositionalEncoder.forward."""
    raw = torch.tensordot(x, bands, dims=0).reshape(x.shape[:-1] + (-1,))
    return torch.cat([raw.sin(), raw.cos()], dim=-1)


# ----------------------------------------------------------------------------- A4 SkipConnMLP


def mlp_linear_shapes(dim_p: int, num_layers: int, hidden: int, out: int, skip: int = 3):
    """src/neural_blocks.py:234-249: (in,out) of init, layers[i], out."""
    shapes = [(dim_p, hidden)]
    for i in range(num_layers):
        shapes.append(((hidden + dim_p) if (i % skip) == 0 and i != num_layers - 1 else hidden, hidden))
    shapes.append((hidden, out))
    return shapes


def _act(kind):
    if kind == "leaky_relu":
        return lambda t: F.leaky_relu(t, 0.01)
    if kind == "sin":
        return torch.sin
    raise NotImplementedError(kind)


def skip_mlp(params: dict, prefix: str, p, latent=None, act: str = "leaky_relu", skip: int = 3,
             enc=None, collect=None):
    """src/neural_blocks.py:279-296 SkipConnMLP.forward.

    params: state-dict style {prefix+'init.weight', prefix+'layers.0.weight', ..., prefix+'out.bias'}.
    enc: callable on [N,in] -> [N,E] or None.  collect: optional list receiving per-layer outputs.
    """
    a = _act(act)
    batches = p.shape[:-1]
    init = p.reshape(-1, p.shape[-1])
    if enc is not None:
        init = torch.cat([init, enc(init)], dim=-1)
    if latent is not None and latent.shape[-1] != 0:
        init = torch.cat([init, latent.reshape(-1, latent.shape[-1])], dim=-1)
    n_layers = 0
    while f"{prefix}layers.{n_layers}.weight" in params:
        n_layers += 1
    x = F.linear(init, params[prefix + "init.weight"], params[prefix + "init.bias"])
    if collect is not None:
        collect.append(x)
    for i in range(n_layers):
        if i != n_layers - 1 and (i % skip) == 0:
            x = torch.cat([x, init], dim=-1)
        x = F.linear(a(x), params[f"{prefix}layers.{i}.weight"], params[f"{prefix}layers.{i}.bias"])
        if collect is not None:
            collect.append(x)
    y = F.linear(a(x), params[prefix + "out.weight"], params[prefix + "out.bias"])
    return y.reshape(batches + (y.shape[-1],))


# ----------------------------------------------------------------------------- A7 view / heads


def dir_to_elev_azim(direc):
    """src/utils.py:247-254."""
    lim = 1 - 1e-6
    x, y, z = F.normalize(direc, dim=-1).clamp(min=-lim, max=lim).split([1, 1, 1], dim=-1)
    return torch.cat([z.acos(), torch.atan2(y, x)], dim=-1)


def sigmoid(kind: str):
    """src/utils.py:484-518 sigmoid_kinds (subset that is reachable from the 5 configs + cheap ones)."""
    fat = lambda v, eps=1e-2: v.sigmoid() * (1 + 2 * eps) - eps
    table = {
        "normal": torch.sigmoid,
        "thin": lambda v: fat(v, -1e-2) + 1e-2,
        "fat": fat,
        "tanh": torch.tanh,
        "upshifted": lambda v: v.sigmoid() + 1e-2,
        "relu": F.relu,
        "sin": torch.sin,
        "leaky_relu": F.leaky_relu,
        "upshifted_softplus": lambda v: F.softplus(v) + 1e-2,
        "upshifted_relu": lambda v: F.relu(v) + 1e-2,
        "cyclic": lambda v: ((v / 5).sin() + 1) / 2 * (1 + 2 * -1e-2) - (-1e-2),
    }
    if kind not in table:
        raise NotImplementedError(kind)
    return table[kind]


def _hash_enc_from(params, prefix):
    tables = [params[f"{prefix}embs.{i}.weight"] for i in range(8)]
    return lambda x: hash_encode(x, tables)


def view_refl(params, prefix, x, view, latent, act="thin"):
    """src/refl.py:190-207 View: act(mlp([x | elaz(view)], latent)), sin activations."""
    v = dir_to_elev_azim(view)
    return sigmoid(act)(skip_mlp(params, prefix + "mlp.", torch.cat([x, v], dim=-1), latent, act="sin"))


def positional_refl(params, prefix, x, latent, act="thin"):
    """src/refl.py:230-245 Positional: act(mlp(x, latent)), hash encoder, 5x256 LeakyReLU."""
    enc = _hash_enc_from(params, prefix + "mlp.enc.")
    return sigmoid(act)(skip_mlp(params, prefix + "mlp.", x, latent, enc=enc))


def pos_linear_view_refl(params, prefix, x, view, latent, act="thin", out_features=3, im=64):
    """src/refl.py:248-290 PosLinearView.forward (view='raw')."""
    enc = _hash_enc_from(params, prefix + "pos.enc.")
    pos_all = sigmoid(act)(skip_mlp(params, prefix + "pos.", x, latent, enc=enc))
    pos, intermediate = pos_all.split([out_features, im], dim=-1)
    view_latent = intermediate if latent is None else torch.cat([latent, intermediate], dim=-1)
    vin = torch.cat([x, F.normalize(view, dim=-1)], dim=-1)
    linear = skip_mlp(params, prefix + "view.", vin, view_latent, act="sin").sigmoid()
    return (linear / 2 + 0.5) * pos


# ----------------------------------------------------------------------------- A6 mip IPE primitives


def expected_sin(x, x_var):
    """src/utils.py:23-27."""
    y = (-0.5 * x_var).exp() * x.sin()
    y_var = (0.5 * (1 - (-2 * x_var).exp() * (2 * x).cos()) - y.square()).clamp(min=0)
    return y, y_var


def integrated_pos_enc_diag(x, x_cov, min_deg: int, max_deg: int):
    """src/utils.py:39-48."""
    scales = torch.exp2(torch.arange(min_deg, max_deg, dtype=x.dtype))
    out_shape = x.shape[:-1] + (-1,)
    y = (x[..., None, :] * scales[..., None]).reshape(out_shape)
    y_var = (x_cov[..., None, :] * scales[..., None].square()).reshape(out_shape)
    return expected_sin(torch.cat([y, y + 0.5 * math.pi], dim=-1), torch.cat([y_var, y_var], dim=-1))[0]


def radii_x(r_d):
    """src/utils.py:77-81 (r_d [B,H,W,3] -> [B,H,W,1])."""
    dx = (r_d[..., :-1, :, :] - r_d[..., 1:, :, :]).square().sum(dim=-1).sqrt()
    dx = torch.cat([dx, dx[:, -2:-1, :]], dim=-2)
    return dx[..., None] * 2 / math.sqrt(12)


def cylinder_moments(t0, t1, rad):
    """src/utils.py:95-101 scalar moments (t_mean, t_var, r_var)."""
    return (t1 + t0) / 2, (t1 - t0).square() / 12, rad * rad / 4


def cone_moments(t0, t1, rad):
    """src/utils.py:83-91 scalar moments (note t_var uses hw/3, as written in the reference)."""
    mu = (t1 + t0) / 2
    hw = (t1 - t0) / 2
    mu2, hw2 = mu * mu, hw * hw
    hw4 = hw2 * hw2
    t_mean = mu + (2 * mu * hw2) / (3 * mu2 + hw2)
    t_var = hw / 3 - (4 / 15) * ((hw4 * (12 * mu2 - hw2)) / (3 * mu2 + hw2).square())
    r_var = rad * rad * (mu2 / 4 + (5 / 12) * hw2 - 4 / 15 * hw4 / (3 * mu2 + hw2))
    return t_mean, t_var, r_var


def lift_gaussian_intended(r_d, t_mean, t_var, r_var):
    """Intended layout of src/utils.py:60-73 (the reference returns cov as [3,B,H,W,T];
    here mean and cov are both [T,B,H,W,3]).  t_* are [T]; r_var is [B,H,W,1] (or [T,B,H,W,1])."""
    T = t_mean.shape[0]
    tm = t_mean.reshape(T, 1, 1, 1, 1)
    tv = t_var.reshape(T, 1, 1, 1, 1)
    rv = r_var if r_var.dim() == 5 else r_var[None]
    mean = r_d[None] * tm
    magn_sq = r_d.square().sum(dim=-1, keepdim=True).clamp(min=1e-10)
    outer_diag = r_d.square()
    null_outer_diag = 1 - outer_diag / magn_sq
    cov = tv * outer_diag[None] + rv * null_outer_diag[None]
    return mean, cov


def mip_latent_intended(r_o, r_d, ts, kind: str = "cylinder", min_deg=0, max_deg=16, end: float = 1e10):
    """Intended composed mip latent (src/nerf.py:256-261 + src/utils.py:103-140): [T,B,H,W,96].
    Deliberate divergence from HEAD; for the cone the last interval is clamped by the
    caller via `end`."""
    t0 = ts
    t1 = torch.cat([ts[1:], torch.tensor([end], dtype=ts.dtype)])
    rad = radii_x(r_d)
    if kind == "cylinder":
        t_mean, t_var, r_var = cylinder_moments(t0, t1, rad)
    else:
        t_mean, t_var, r_var = cone_moments(t0.reshape(-1, 1, 1, 1, 1), t1.reshape(-1, 1, 1, 1, 1), rad[None])
        t_mean, t_var = t_mean.reshape(-1), t_var.reshape(-1)
    mean, cov = lift_gaussian_intended(r_d, t_mean, t_var, r_var)
    mean = mean + r_o[None]
    return integrated_pos_enc_diag(mean, cov, min_deg, max_deg)


# ----------------------------------------------------------------------------- A11 Bezier


def de_casteljau(coeffs, t, N: int):
    """src/nerf.py:1173-1178."""
    betas = coeffs
    m1t = 1 - t
    for _ in range(1, N):
        betas = betas[:-1] * m1t + betas[1:] * t
    return betas.squeeze(0)


def cubic_bezier(coeffs, t, N: int = 4):
    """src/nerf.py:1201-1206."""
    m1t = 1 - t
    m1t_sq, t_sq = m1t * m1t, t * t
    k = torch.stack([m1t_sq * m1t, 3 * m1t_sq * t, 3 * t_sq * m1t, t_sq * t], dim=0)
    return (k * coeffs).sum(dim=0)


# ----------------------------------------------------------------------------- A12 VolSDF density


def laplace_cdf(sdf_vals, scale):
    """src/utils.py:50-58."""
    scaled = sdf_vals / scale
    return torch.where(scaled <= 0, scaled.clamp(max=0).exp() / 2, 1 - scaled.clamp(min=0).neg().exp() / 2)


# ----------------------------------------------------------------------------- model forwards


def _sky(bg, weights):
    if bg == "black":
        return 0
    if bg == "white":
        return sky_white(weights)
    if isinstance(bg, tuple) and bg[0] == "random":  # ("random", rand [..., 1])
        return sky_random(weights, bg[1])
    raise NotImplementedError(bg)


def tiny_nerf(params, rays, near, far, steps, act="upshifted", bg="black", aux=None):
    """Intended TinyNeRF (src/nerf.py:278-305 composed as SURVEY 8(c).5: density = estim[...,0])."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    ts, _ = compute_ts(near, far, steps)
    pts = compute_pts(r_o, r_d, ts)
    out = skip_mlp(params, "estim.", pts)
    density, feats = out[..., 0], out[..., 1:]
    alpha, weights = alpha_from_density(density, ts, r_d)
    if aux is not None:
        aux.update(ts=ts, alpha=alpha, weights=weights)
    return volumetric_integrate(weights, sigmoid(act)(feats)) + _sky(bg, weights)


def _refl_dispatch(params, refl_kind, x, view, latent, act):
    if refl_kind == "view":
        return view_refl(params, "refl.", x, view, latent, act)
    if refl_kind == "pos":
        return positional_refl(params, "refl.", x, latent, act)
    if refl_kind == "pos-linear-view":
        return pos_linear_view_refl(params, "refl.", x, view, latent, act)
    raise NotImplementedError(refl_kind)


def plain_nerf_from_pts(params, pts, ts, r_o, r_d, refl_kind="view", act="thin", bg="black",
                        mip_latent=None, refl_latent=None, aux=None, prefix=""):
    """src/nerf.py:337-361 PlainNeRF.from_pts (eval mode)."""
    p = {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)} if prefix else params
    latent = mip_latent
    first_out = skip_mlp(p, "first.", pts, latent, enc=_hash_enc_from(p, "first.enc."))
    density = first_out[..., 0]
    intermediate = first_out[..., 1:]
    view = r_d.unsqueeze(0).expand_as(pts)
    rl = intermediate if refl_latent is None else torch.cat([intermediate, refl_latent], dim=-1)
    if latent is not None:
        rl = torch.cat([latent, rl], dim=-1)
    rgb = _refl_dispatch(p, refl_kind, pts, view, rl, act)
    alpha, weights = alpha_from_density(density, ts, r_d)
    if aux is not None:
        aux.update(ts=ts, alpha=alpha, weights=weights, density=density, rgb=rgb)
    return volumetric_integrate(weights, rgb) + _sky(bg, weights)


def plain_nerf(params, rays, near, far, steps, refl_kind="view", act="thin", bg="black",
               mip: Optional[str] = None, aux=None):
    """src/nerf.py:326-361 PlainNeRF.forward (eval mode: no perturb, no density noise)."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    ts, _ = compute_ts(near, far, steps)
    pts = compute_pts(r_o, r_d, ts)
    mip_latent = None if mip is None else mip_latent_intended(r_o, r_d, ts, mip, end=(2 * ts[-1] - ts[-2]).item())
    return plain_nerf_from_pts(params, pts, ts, r_o, r_d, refl_kind, act, bg, mip_latent, aux=aux)


# ----------------------------------------------------------------------------- coarse -> fine (parity unpinned: intended reading)
def linspace01_f32(N: int):
    """torch.linspace(0, 1, N) as ATen's scalar CPU kernel evaluates it in fp32 -- step = 1 / (N - 1); the lower half counts up
    from 0, the upper half down from 1 -- with every operation rounded separately (numpy float32; torch's vectorised kernels may
    fuse the multiply-add, a last-bit difference that an almost empty interval of the cdf amplifies to 1e-4 of a step: the
    deterministic draw is therefore DEFINED by this formula, here and in csrc/basic_ops.hip)."""
    import numpy as np
    step = np.float32(1.0) / np.float32(max(N - 1, 1))
    j = np.arange(N)
    lo = step * j.astype(np.float32)
    hi = np.float32(1.0) - step * (N - 1 - j).astype(np.float32)
    return torch.from_numpy(np.where((j < N // 2) | (N == 1), lo, hi).astype(np.float32))  # (N = 1: [0], like torch)


def sample_pdf_intended(ts, weights, N: int, u=None):
    """The reference's sample_pdf (src/nerf.py:1745-1779; call site :572-578 passes (mids, weights[:-1], steps_fine)) cannot run:
    it gathers its bins with cdf indices (T entries against T - 1 mids) and calls exit().  INTENDED reading, in fp64: weight i of
    weights[:-1] is the mass of [ts[i], ts[i + 1]] (alpha_from_density's own definition), the bins are `ts`.  PARITY UNPINNED --
    there is no reference output to pin this against; csrc/basic_ops.hip `resample_ts_kernel` restates the same arithmetic.
    ts [T]; weights [T, *batch] (the last row is dropped here); u None (linspace(0, 1, N), fp32 like the reference) or
    [N, *batch].  Returns float64 [N, *batch]."""
    w = weights[:-1].double() + 1e-5
    pdf = w / w.sum(dim=0, keepdim=True)
    cdf = torch.cumsum(pdf, dim=0)
    cdf = torch.cat([torch.zeros_like(cdf[:1]), cdf], dim=0)                       # [T, *batch]
    T = cdf.shape[0]
    batch = cdf.shape[1:]
    if u is None:
        u = linspace01_f32(N).reshape((N,) + (1,) * len(batch)).expand((N,) + tuple(batch))
    u = u.double().contiguous()
    c2 = cdf.reshape(T, -1).t().contiguous()                                        # [rays, T]
    u2 = u.reshape(N, -1).t().contiguous()                                          # [rays, N]
    inds = torch.searchsorted(c2, u2, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=T - 1)
    c0, c1 = torch.gather(c2, 1, below), torch.gather(c2, 1, above)
    tsd = ts.double()
    b0, b1 = tsd[below], tsd[above]
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u2 - c0) / denom
    samples = b0 + t * (b1 - b0)
    return samples.t().reshape((N,) + tuple(batch))


def merge_ts_intended(ts, fine):
    """z_vals of the fine pass: the coarse steps ts [T] and the new positions fine [N, *batch] of every ray in increasing order
    (stable, coarse first) -> [T + N, *batch]."""
    batch = fine.shape[1:]
    allv = torch.cat([ts.reshape((-1,) + (1,) * len(batch)).expand((ts.shape[0],) + tuple(batch)).to(fine.dtype), fine], dim=0)
    return torch.sort(allv, dim=0, stable=True).values


def plain_nerf_rayts(params, rays, ts_ray, refl_kind="view", act="thin", bg="black", aux=None):
    """PlainNeRF.from_pts with PER-RAY steps ts_ray [T, *batch] (the fine pass): positions o + t d, interval lengths along the
    step axis (the reference's alpha_from_density differences ts along its LAST axis: shared steps only), last interval 1e10."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    pts = r_o.unsqueeze(0) + ts_ray.unsqueeze(-1) * r_d.unsqueeze(0)
    first_out = skip_mlp(params, "first.", pts, None, enc=_hash_enc_from(params, "first.enc."))
    density, intermediate = first_out[..., 0], first_out[..., 1:]
    view = r_d.unsqueeze(0).expand_as(pts)
    rgb = _refl_dispatch(params, refl_kind, pts, view, intermediate, act)
    sigma_a = F.softplus(density - 1)
    dists = torch.cat([ts_ray[1:] - ts_ray[:-1], torch.full_like(ts_ray[:1], 1e10)], dim=0).clamp(min=1e-5)
    dists = dists * torch.linalg.norm(r_d, dim=-1)
    alpha = 1 - torch.exp(-sigma_a * dists)
    weights = alpha * cumuprod_exclusive(1.0 - alpha + 1e-10)
    if aux is not None:
        aux.update(alpha=alpha, weights=weights)
    return volumetric_integrate(weights, rgb) + _sky(bg, weights)


def volsdf(params, rays, near, far, steps, sdf_kind="mlp", refl_kind="view", act="thin", aux=None):
    """src/nerf.py:981-1013 VolSDF.forward/from_pts (no normals, no secondary), src/sdf.py:109-112,250-287."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    ts, _ = compute_ts(near, far, steps)
    pts = compute_pts(r_o, r_d, ts)
    if sdf_kind == "mlp":
        basis = params["sdf.underlying.mlp.enc.basis"]
        raw = skip_mlp(params, "sdf.underlying.mlp.", pts, enc=lambda x: fourier_encode(x, basis))
    elif sdf_kind == "siren":
        raw = skip_mlp(params, "sdf.underlying.siren.", pts, act="sin")
    else:
        raise NotImplementedError(sdf_kind)
    sdf_vals, latent = raw[..., 0], raw[..., 1:]
    scale = params["scale"]
    density = 1 / scale * laplace_cdf(-sdf_vals, scale)
    alpha, weights = alpha_from_density(density, ts, r_d, softplus=False)
    view = r_d.unsqueeze(0).expand_as(pts)
    rgb = _refl_dispatch({k[len("sdf."):]: v for k, v in params.items() if k.startswith("sdf.")},
                         refl_kind, pts, view, latent, act)
    if aux is not None:
        aux.update(ts=ts, alpha=alpha, weights=weights, sdf=sdf_vals)
    return volumetric_integrate(weights, rgb)


def dynamic_nerf_spline(params, rays, times, near, far, steps, spline: int, refl_kind="view", act="thin",
                        bg="black", aux=None, refl_latent: int = 0):
    """src/nerf.py:1241-1303 DynamicNeRF (spline>1) over a canonical PlainNeRF; refl_latent > 0 (:1246-1248, 1272-1278): the
    network's extra enc_rigidity | spline * refl_latent columns go through the same spline, scaled by sigmoid(enc_rigidity),
    and reach the canonical model's reflectance as `refl_latent` (:1303)."""
    r_o, r_d = rays.split([3, 3], dim=-1)
    ts, _ = compute_ts(near, far, steps)
    pts = compute_pts(r_o, r_d, ts)
    t = times[None, :, None, None, None].expand(*pts.shape[:-1], 1)
    est = skip_mlp(params, "delta_estim.", pts, enc=_hash_enc_from(params, "delta_estim.enc."))
    layout = [1, 3 * spline] + ([1, refl_latent * spline] if refl_latent > 0 else [0, 0])
    rigidity, ps, enc_rigidity, enc = est.split(layout, dim=-1)
    rigidity = (rigidity / 2).sigmoid()
    ps = torch.stack(ps.split([3] * spline, dim=-1), dim=0)
    fn = cubic_bezier if spline == 4 else de_casteljau
    if refl_latent > 0:
        enc = torch.stack(enc.split([refl_latent] * spline, dim=-1), dim=0)
        dp, enc = fn(torch.cat([ps, enc], dim=-1), t, spline).split([3, refl_latent], dim=-1)
        enc = enc * enc_rigidity.sigmoid()
    else:
        dp, enc = fn(ps, t, spline), None
    rigid_dp = dp * rigidity
    if aux is not None:
        aux.update(dp=dp, rigidity=rigidity, rigid_dp=rigid_dp, pts=pts, refl_latent=enc)
    return plain_nerf_from_pts(params, pts + rigid_dp, ts, r_o, r_d, refl_kind, act, bg, refl_latent=enc, aux=aux,
                               prefix="canonical.")


def dnerf_rigid_dp(params, pts, t, spline: int):
    """src/nerf.py:1267-1278 spline_interpolate: dp * rigidity at points pts [...,3] and times t [...,1]."""
    est = skip_mlp(params, "delta_estim.", pts, enc=_hash_enc_from(params, "delta_estim.enc."))
    rigidity = (est[..., :1] / 2).sigmoid()
    ps = torch.stack(est[..., 1:1 + 3 * spline].split([3] * spline, dim=-1), dim=0)
    dp = (cubic_bezier if spline == 4 else de_casteljau)(ps, t, spline)
    return dp * rigidity


def div_approx(x, fn_x, e):
    """src/utils.py:467-478 with the reference's randn_like draw passed in as `e`: <e, (d fn_x / d x)^T e>.  Like the
    reference it builds no graph for the result (no create_graph): the estimate is a constant for the optimiser."""
    e_dydx, = torch.autograd.grad(inputs=x, outputs=fn_x, grad_outputs=e, retain_graph=True, only_inputs=True)
    return (e_dydx * e).sum(dim=-1)


def ffjord_div(params, pts, times, spline: int, e):
    """runner.py:697-699 for a DynamicNeRF: div_approx(model.pts, model.rigid_dp); pts [T,B,H,W,3], times [B]."""
    with torch.enable_grad():
        x = pts.detach().clone().requires_grad_()
        t = times[None, :, None, None, None].expand(*x.shape[:-1], 1)
        return div_approx(x, dnerf_rigid_dp(params, x, t, spline), e).detach()


# ----------------------------------------------------------------------------- test()-style frame


def mse2psnr(mse):
    """src/utils.py:184."""
    return -10 * torch.log10(mse)


def render_tiled(model_fn, c2w, focal, size: int, crop_size: int):
    """runner.py:879-892: tile loop, x over rows then y over cols, ragged last tile via slicing."""
    got = torch.zeros(size, size, 3)
    n = math.ceil(size / crop_size)
    for x in range(n):
        c0 = x * crop_size
        for y in range(n):
            c1 = y * crop_size
            pos = pixel_grid(size, (c0, c1, crop_size, crop_size))
            rays = nerf_camera_rays(pos, c2w, focal, size)
            got[c0:c0 + crop_size, c1:c1 + crop_size, :] = model_fn(rays).squeeze(0)
    return got


# ----------------------------------------------------------------------------- N3 auxiliary maps (runner.py:511-538, 894-913)


def depth_to_normals(depth_img):
    """src/utils.py:421-427: forward differences of a depth image [H,W,1] -> unit normals [H-1,W-1,3]."""
    dz_dx = depth_img[1:, 1:, ...] - depth_img[:-1, 1:, ...]
    dz_dy = depth_img[1:, 1:, ...] - depth_img[1:, :-1, ...]
    d = torch.cat([dz_dx / 2, dz_dy / 2, torch.ones_like(dz_dx)], dim=-1)
    return F.normalize(d, dim=-1)


def depth_vis(weights, ts, near: float, far: float, normals_from_depth: bool = False):
    """runner.py:511-519 with the evident intent of its second line, ((raw - near) / (far - near)).clamp(0, 1): as written the
    clamp binds to the denominator `(args.far - args.near)` and exists only for tensor arguments (tools/gen_golden.py g18 calls
    it that way, with far - near = 1, where both readings agree on [0, 1]).  Returns [depth(, normal map)] of batch item 0."""
    raw = volumetric_integrate(weights, ts[:, None, None, None, None])
    depth = ((raw[0] - near) / (far - near)).clamp(min=0, max=1)
    items = [depth]
    if normals_from_depth:
        items.append(((50 * depth_to_normals(depth) + 1) / 2).clamp(min=0, max=1))
    return items


def flow_vis(weights, rigid_dp):
    """runner.py:521-526: integrated flow of batch item 0, normalised by its largest vector norm, signed square root, -> [0,1]."""
    flow = volumetric_integrate(weights, rigid_dp)[0]
    flow = flow / flow.norm(dim=-1).max()
    flow = flow.abs().sqrt().copysign(flow)
    return (flow + 1) / 2


def rigidity_vis(weights, rigidity):
    """runner.py:528-531."""
    return volumetric_integrate(weights, rigidity)[0]


# ----------------------------------------------------------------------------- N4 SDF marching (src/march.py)
# Dense restatements: the reference gathers the active rays with boolean masks; updating only where the mask is set
# is the same computation ray by ray.  `sdf_fn(pts[..., 3]) -> [..., >=1]`, column 0 = signed distance.


def sphere_march(sdf_fn, r_o, r_d, iters: int = 32, eps: float = 1e-3, near: float = 0, far: float = 1):
    """src/march.py:27-47."""
    hits = torch.zeros(r_o.shape[:-1] + (1,), dtype=torch.bool)
    rem = torch.ones(r_o.shape[:-1], dtype=torch.bool)
    curr_dist = torch.full(r_o.shape[:-1] + (1,), float(near))
    for _ in range(iters):
        dist = sdf_fn(r_o + r_d * curr_dist)[..., 0:1]
        act = rem.unsqueeze(-1)
        hits = hits | (act & (dist < eps) & (curr_dist <= far))
        curr_dist = torch.where(act, curr_dist + dist, curr_dist)
        rem = rem & ~(hits.squeeze(-1) | (curr_dist > far).squeeze(-1))
    return r_o + r_d * curr_dist, hits.squeeze(-1), curr_dist, None


def throughput_with_sign_change(sdf_fn, r_o, r_d, near: float, far: float, batch_size: int = 128, jitter: float = 0.0):
    """src/march.py:78-110; `jitter` stands for the reference's random.random() draw."""
    max_t = far - near + jitter * (2 / batch_size)
    step = max_t / batch_size
    curr_min = sdf_fn(r_o + near)[..., 0]  # sic (src/march.py:90): the scalar is added to the origin
    idxs = torch.zeros_like(curr_min, dtype=torch.long)
    last_pos = torch.full_like(idxs, -1)
    first_neg = torch.full_like(idxs, -1)
    for i in range(batch_size):
        t = near + step * (i + 1)
        sd = sdf_fn(r_o + t * r_d)[..., 0]
        idxs = torch.where(sd < curr_min, i + 1, idxs)
        curr_min = torch.minimum(curr_min, sd)
        mask = (first_neg == -1) & (sd < 0)
        last_pos = torch.where(mask, i, last_pos)
        first_neg = torch.where(mask, i + 1, first_neg)
    best_pos = r_o + (near + idxs.unsqueeze(-1) * step) * r_d
    return sdf_fn(best_pos)[..., 0], best_pos, last_pos.unsqueeze(-1) * step, first_neg.unsqueeze(-1) * step


def bisection(sdf_fn, r_o, r_d, near, far, iters: int = 32, eps: float = 1e-6):
    """src/march.py:147-180 (near/far: per-ray [..., 1] tensors; not modified here)."""
    low, high = near.clone(), far.clone()
    sdf_low = sdf_fn(r_o + low * r_d)[..., 0, None]
    sdf_high = sdf_fn(r_o + high * r_d)[..., 0, None]
    todo = ((high - low) > eps) & (sdf_low > 0) & (sdf_high < 0) & (high > low)
    z = (low + high) / 2
    for _ in range(iters):
        if not todo.any():
            break
        sdf_mid = sdf_fn(r_o + z * r_d)[..., 0, None]
        lm = (sdf_mid > 0) & todo
        low = torch.where(lm, z, low)
        sdf_low = torch.where(lm, sdf_mid, sdf_low)
        hm = (sdf_mid < 0) & todo
        high = torch.where(hm, z, high)
        sdf_high = torch.where(hm, sdf_mid, sdf_high)
        z = (low + high) / 2
        todo = todo & ((high - low) > eps) & (sdf_low > 0) & (sdf_high < 0) & (high > low)
    return r_o + z * r_d


def bisect(sdf_fn, r_o, r_d, iters: int = 128, near: float = 0, far: float = 1, jitter: float = 0.0):
    """src/march.py:63-75."""
    tput, best_pos, last_pos, first_neg = throughput_with_sign_change(sdf_fn, r_o, r_d, near, far, iters, jitter)
    pts = bisection(sdf_fn, r_o, r_d, last_pos, first_neg, iters=min(32, iters))
    return pts, tput < 0, best_pos, tput.unsqueeze(-1)


# ----------------------------------------------------------------------------- N4 lights + occlusion


def point_light(x, center, intensity, distance_decay: bool = True):
    """src/lights.py:118-132 Point.forward for one light: center/intensity [3] (what `Point.iter()` hands out, curr_idx 0);
    x [...,3] (already masked by the caller) -> (unit direction to the light, distance [...,1], spectrum [...,3])."""
    d = center - x
    dist = torch.linalg.norm(d, ord=2, dim=-1, keepdim=True)
    d = F.normalize(d, eps=1e-6, dim=-1)
    spectrum = (intensity / (4 * math.pi * dist.square())) if distance_decay else intensity.expand_as(x)
    return d, dist, spectrum


def intersect_mask(sdf_fn, r_o, r_d, near: float, far: float, eps: float = 1e-3, batch_size: int = 196, jitter: float = 0.0):
    """src/sdf.py:123-135 (eval mode: 196 uniform probes): (visible = ~(throughput < eps), throughput)."""
    tput, _, _, _ = throughput_with_sign_change(sdf_fn, r_o, r_d, near, far, batch_size, jitter=jitter)
    return ~(tput < eps), tput


def occlusion(kind, params, pts, center, intensity, sdf_fn, jitter: float = 0.0, mask=None, alpha=None,
              component: str = "pos-elaz", aux=None):
    """src/renderers.py:29-163: (direction to the light, attenuated spectrum) for one point light.
    kind: None | hard | learned | learned-const | all-learned | joint-all-const.  params: the occlusion module's
    state dict ("attenuation." / "alo.attenuation." SkipConnMLP with a FourierEncoder, "alpha" / "lcsl.alpha")."""
    x = pts if mask is None else pts[mask]
    d, dist, spectrum = point_light(x, center, intensity)
    aux = {} if aux is None else aux

    def visible(near, far, eps=1e-3):
        vis, tput = intersect_mask(sdf_fn, x, d, near, far, eps, jitter=jitter)
        aux["tput"], aux["far"] = tput, far
        return vis

    def att_mlp(prefix, inp):
        basis = params[prefix + "enc.basis"]
        return skip_mlp(params, prefix, inp, enc=lambda v: fourier_encode(v, basis))

    def far_of(default=6):  # :37,62 / :81 (no mask -> 6) / :140
        return float(dist.max())

    if kind is None:                                   # :29-31
        return d, spectrum
    if kind == "hard":                                 # :34-46
        vis = visible(0.1, far_of())
        return d, torch.where(vis[..., None], spectrum, torch.zeros_like(spectrum))
    if kind == "learned":                              # :48-68
        vis = visible(2e-3, far_of())
        att = att_mlp("attenuation.", torch.cat([x, dir_to_elev_azim(d)], dim=-1)).sigmoid()
        return d, torch.where(vis.reshape_as(att), spectrum, spectrum * att)
    if kind == "learned-const":                        # :70-84 (`mask and mask.any()` -> far 6 without a mask)
        vis = visible(1e-2, far_of() if mask is not None else 6)
        hit_att = vis + (~vis) * torch.as_tensor(alpha).sigmoid()
        return d, spectrum * hit_att.unsqueeze(-1)
    comp = (lambda a, b: a) if component == "pos" else (lambda a, b: torch.cat([a, dir_to_elev_azim(b)], dim=-1))
    if kind == "all-learned":                          # :96-121
        raw = att_mlp("attenuation.", comp(x, d))
        aux["raw_att"] = raw
        return d, spectrum * (raw.sigmoid() + 1e-2)
    if kind == "joint-all-const":                      # :123-147
        assert mask is None, "src/renderers.py:138"
        raw = att_mlp("alo.attenuation.", comp(x, d))
        aux["raw_att"] = raw
        vis = visible(1e-1, far_of())
        hit_att = vis + (~vis) * torch.as_tensor(alpha).sigmoid()
        return d, spectrum * (raw.sigmoid() + 1e-2) * hit_att.unsqueeze(-1)
    raise NotImplementedError(kind)
