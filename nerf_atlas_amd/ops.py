"""Torch-tensor wrappers over the C ABI (include/nerf_atlas_amd.h).

PyTorch is only the memory/stream plumbing here: every function checks its tensors (cuda, fp32,
contiguous), passes raw device pointers + the current HIP stream to the HIP library and returns the
output tensor.  There is no eager/CPU fallback.
"""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib, config
from ._lib import NaMlpDesc, check

ACT = {"none": 0, "leaky_relu": 1, "sin": 2}
ENC = {"none": 0, "hash": 1, "fourier": 2}
# (f16: everything but the register-engine renderer na_render_plain_view; f16x: render_ls_pack / render_plain_view_ls only)
PREC = {"bf16": 0, "bf16x3": 1, "f16": 2, "f16x": 3}
LAYOUT = {"generic": 0, "plain_first": 1, "plain_view": 2}
BG = {"black": 0, "white": 1}
SIGMOID = {"normal": 0, "thin": 1, "fat": 2, "tanh": 3, "upshifted": 4, "relu": 5, "sin": 6, "leaky_relu": 7,
           "upshifted_softplus": 8, "upshifted_relu": 9, "cyclic": 10, "identity": 11}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class F16xSaturated(_lib.NaError):
    """An f16x launch met an activation beyond the IEEE-half range (its output is the NaN frame): config.set_f16x_on_saturation."""


def _f16x_guard(precision: str, out: torch.Tensor, what: str):
    """config.f16x_on_saturation "raise" / "rerender_bf16x3": read one element of the launch's output back (the range guard poisons
    the whole of it) and raise when it is the flagged frame.  "nan" (default): nothing, no synchronisation."""
    if precision == "f16x" and config.f16x_on_saturation != "nan" and out.numel() > 0:
        if bool(torch.isnan(out.reshape(-1)[0])):
            raise F16xSaturated(f"{what}: an activation left the half range (f16x output poisoned); render these weights in bf16x3")


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA(HIP) tensor: the hot path has no CPU implementation")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------------------------- rays / samples
def raygen(c2w: torch.Tensor, focal: float, size: int, crop, noise: Optional[torch.Tensor] = None,
           with_noise: float = 0.0) -> torch.Tensor:
    """rays[B,h,w,6] for crop (t,l,h,w) clipped to the image like the reference's slicing (runner.py:490-503,
    src/cameras.py:45-66)."""
    lib = _lib.load()
    c2w = _f32(c2w, "c2w")
    B = c2w.shape[0]
    t, l, h, w = crop
    h = max(0, min(h, size - t))
    w = max(0, min(w, size - l))
    rays = torch.empty(B, h, w, 6, device=c2w.device, dtype=torch.float32)
    nz = None
    if noise is not None and with_noise:
        nz = _f32(noise, "noise")
        assert nz.shape == (h, w, 2), nz.shape
    check(lib.na_raygen(_ptr(c2w), B, float(focal), int(size), t, l, h, w, _ptr(nz), float(with_noise or 0.0),
                        _ptr(rays), _stream()))
    return rays


def raygen_dtu(pose: torch.Tensor, intrinsic: torch.Tensor, size: int, crop) -> torch.Tensor:
    lib = _lib.load()
    pose, intrinsic = _f32(pose, "pose"), _f32(intrinsic, "intrinsic")
    B = pose.shape[0]
    t, l, h, w = crop
    h = max(0, min(h, size - t))
    w = max(0, min(w, size - l))
    rays = torch.empty(B, h, w, 6, device=pose.device, dtype=torch.float32)
    check(lib.na_raygen_dtu(_ptr(pose), _ptr(intrinsic), B, int(size), t, l, h, w, _ptr(rays), _stream()))
    return rays


def compute_ts(near: float, far: float, steps: int, device, lindisp: bool = False, perturb: float = 0.0,
               rand: Optional[torch.Tensor] = None, want_mids: bool = False):
    lib = _lib.load()
    ts = torch.empty(steps, device=device, dtype=torch.float32)
    mids = torch.empty(max(steps - 1, 0), device=device, dtype=torch.float32) if (want_mids or perturb > 0) else None
    if perturb > 0:
        rand = _f32(rand, "rand")
        assert rand.shape == (steps,)
    check(lib.na_compute_ts(float(near), float(far), int(steps), int(lindisp), float(perturb), _ptr(rand), _ptr(ts),
                            _ptr(mids), _stream()))
    return ts, (mids if perturb > 0 else None)


def compute_pts(rays: torch.Tensor, ts: torch.Tensor) -> torch.Tensor:
    """pts[T, *rays.shape[:-1], 3] (src/nerf.py:50-55)."""
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    R = rays.numel() // 6
    T = ts.shape[0]
    pts = torch.empty((T,) + tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    check(lib.na_compute_pts(_ptr(rays), _ptr(ts), T, R, _ptr(pts), _stream()))
    return pts


# ------------------------------------------------------------------------------------------------- encoders
def hash_encode(x: torch.Tensor, tables: torch.Tensor, include_input: bool = True, want_indices: bool = False):
    lib = _lib.load()
    x, tables = _f32(x, "x"), _f32(tables, "tables")
    assert x.shape[-1] == 3 and tables.shape == (8, 65536, 4), (x.shape, tables.shape)
    N = x.numel() // 3
    out = torch.empty(tuple(x.shape[:-1]) + (32 + 3 * int(include_input),), device=x.device, dtype=torch.float32)
    idx = torch.empty(8, 8, N, device=x.device, dtype=torch.int64) if want_indices else None
    check(lib.na_hash_encode(_ptr(x), N, _ptr(tables), int(include_input), _ptr(out), _ptr(idx), _stream()))
    return (out, idx) if want_indices else out


def hash_encode_rows(x: torch.Tensor, tables: torch.Tensor, include_input: bool = True, lead: int = 1) -> torch.Tensor:
    """[N, 32 + 3 (include_input + lead)] rows [x (lead) | x (include_input) | features]: with lead = 1 the init rows cat([p, enc(p)])
    of a hash-encoded SkipConnMLP, written by the encoder (na_hash_encode_rows)."""
    lib = _lib.load()
    x, tables = _f32(x, "x"), _f32(tables, "tables")
    assert x.dim() == 2 and x.shape[-1] == 3 and tables.shape == (8, 65536, 4), (x.shape, tables.shape)
    N = x.shape[0]
    out = torch.empty(N, 32 + 3 * (int(include_input) + lead), device=x.device, dtype=torch.float32)
    check(lib.na_hash_encode_rows(_ptr(x), N, _ptr(tables), int(include_input), lead, _ptr(out), _stream()))
    return out


def plain_head_rows(first_out: torch.Tensor, pts: torch.Tensor, dirs: torch.Tensor):
    """first_out [N, 1 + C], pts [N = T x R, 3], dirs [R, 3] -> (density [N], rows [N, 5 + C] = [x | elev, azim | first_out[:, 1:]])."""
    lib = _lib.load()
    first_out, pts, dirs = _f32(first_out, "first_out"), _f32(pts, "pts"), _f32(dirs, "dirs")
    N, C = first_out.shape[0], first_out.shape[1] - 1
    R = dirs.numel() // 3
    assert pts.numel() == 3 * N and N % R == 0, (pts.shape, dirs.shape, N)
    density = torch.empty(N, device=pts.device, dtype=torch.float32)
    rows = torch.empty(N, 5 + C, device=pts.device, dtype=torch.float32)
    check(lib.na_plain_head_rows(_ptr(first_out), _ptr(pts), _ptr(dirs), N, R, C, _ptr(density), _ptr(rows), _stream()))
    return density, rows


def plain_head_rows_backward(g_density, g_rows: torch.Tensor, want_pts: bool):
    lib = _lib.load()
    g_rows = _f32(g_rows, "g_rows")
    N, C = g_rows.shape[0], g_rows.shape[1] - 5
    g_density = None if g_density is None else _f32(g_density, "g_density")
    g_first = torch.empty(N, 1 + C, device=g_rows.device, dtype=torch.float32)
    g_pts = torch.empty(N, 3, device=g_rows.device, dtype=torch.float32) if want_pts else None
    check(lib.na_plain_head_rows_backward(None if g_density is None else _ptr(g_density), _ptr(g_rows), N, C, _ptr(g_first),
                                          None if g_pts is None else _ptr(g_pts), _stream()))
    return g_first, g_pts


def hash_encode_backward_rows(x: torch.Tensor, g_rows: torch.Tensor, col0: int) -> torch.Tensor:
    """tables gradient from the 32 feature columns col0.. of wider gradient rows, read in place"""
    lib = _lib.load()
    x, g_rows = _f32(x, "x"), _f32(g_rows, "g_rows")
    N = x.numel() // 3
    assert g_rows.dim() == 2 and g_rows.shape[0] == N
    tg = torch.zeros(8, 65536, 4, device=x.device, dtype=torch.float32)
    check(lib.na_hash_encode_backward_rows(_ptr(x), N, _ptr(g_rows), g_rows.shape[1], col0, _ptr(tg), _stream()))
    return tg


def hash_encode_backward_input_rows(x: torch.Tensor, tables: torch.Tensor, g_rows: torch.Tensor, include_input: bool, lead: int) -> torch.Tensor:
    lib = _lib.load()
    x, tables, g_rows = _f32(x, "x"), _f32(tables, "tables"), _f32(g_rows, "g_rows")
    N = x.numel() // 3
    assert g_rows.dim() == 2 and g_rows.shape[0] == N
    gx = torch.empty_like(x)
    check(lib.na_hash_encode_backward_input_rows(_ptr(x), N, _ptr(tables), _ptr(g_rows), g_rows.shape[1], int(include_input), lead,
                                                 _ptr(gx), _stream()))
    return gx


def fourier_encode(x: torch.Tensor, basis: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    lib = _lib.load()
    x, basis = _f32(x, "x"), _f32(basis, "basis")
    D, F = basis.shape
    assert x.shape[-1] == D
    N = x.numel() // D
    out = torch.empty(tuple(x.shape[:-1]) + (2 * F,), device=x.device, dtype=torch.float32)
    check(lib.na_fourier_encode(_ptr(x), N, D, _ptr(basis), F, float(scale), _ptr(out), _stream()))
    return out


def positional_encode(x: torch.Tensor, bands: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x, bands = _f32(x, "x"), _f32(bands, "bands")
    D, NB = x.shape[-1], bands.shape[0]
    N = x.numel() // D
    out = torch.empty(tuple(x.shape[:-1]) + (2 * D * NB,), device=x.device, dtype=torch.float32)
    check(lib.na_positional_encode(_ptr(x), N, D, _ptr(bands), NB, _ptr(out), _stream()))
    return out


def view_elaz(dirs: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    dirs = _f32(dirs, "dirs")
    N = dirs.numel() // 3
    out = torch.empty(tuple(dirs.shape[:-1]) + (2,), device=dirs.device, dtype=torch.float32)
    check(lib.na_view_elaz(_ptr(dirs), N, _ptr(out), _stream()))
    return out


def view_rows(pts: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """[..., R, 5] rows [x, y, z, elev, azim]: pts [..., R, 3] with one direction per ray dirs [R, 3] (na_view_rows)."""
    lib = _lib.load()
    pts, dirs = _f32(pts, "pts"), _f32(dirs, "dirs")
    R = dirs.shape[0]
    assert pts.shape[-2] == R and pts.shape[-1] == 3 and dirs.shape == (R, 3)
    out = torch.empty(tuple(pts.shape[:-1]) + (5,), device=pts.device, dtype=torch.float32)
    check(lib.na_view_rows(_ptr(pts), _ptr(dirs), pts.numel() // 3, R, _ptr(out), _stream()))
    return out


def sigmoid(x: torch.Tensor, kind: str) -> torch.Tensor:
    lib = _lib.load()
    if kind not in SIGMOID:
        raise NotImplementedError(f"Unknown sigmoid kind({kind})")
    x = _f32(x, "x")
    out = torch.empty_like(x)
    check(lib.na_sigmoid(_ptr(x), x.numel(), SIGMOID[kind], _ptr(out), _stream()))
    return out


MIP_KIND = {"cylinder": 0, "cone": 1}


def mip_encode(rays: torch.Tensor, ts: torch.Tensor, kind: str, t_end: float, min_deg: int = 0, max_deg: int = 16):
    """rays [B,H,W,6] of ONE crop -> [T,B,H,W,6*(max_deg-min_deg)] (intended layout)."""
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    B, H, W, _ = rays.shape
    T = ts.shape[0]
    out = torch.empty(T, B, H, W, 6 * (max_deg - min_deg), device=rays.device, dtype=torch.float32)
    check(lib.na_mip_encode(_ptr(rays), B, H, W, _ptr(ts), T, MIP_KIND[kind], float(t_end), min_deg,
                            max_deg, _ptr(out), _stream()))
    return out


# ------------------------------------------------------------------------------------------------- compositing
def composite(density: torch.Tensor, feat: torch.Tensor, ts: torch.Tensor, rays: torch.Tensor, softplus: bool = True,
              bg: str = "black", want_weights: bool = True, rand: Optional[torch.Tensor] = None):
    """density [T,...], feat [T,...,C], rays [...,6] -> (out [...,C], alpha [T,...], weights [T,...]).
    bg "random" (src/nerf.py:99-103) takes the per-ray uniform draw `rand` [..., 1]."""
    lib = _lib.load()
    density, feat, ts, rays = _f32(density, "density"), _f32(feat, "feat"), _f32(ts, "ts"), _f32(rays, "rays")
    T = ts.shape[0]
    Cn = feat.shape[-1]
    R = rays.numel() // 6
    assert density.numel() == T * R and feat.numel() == T * R * Cn, (density.shape, feat.shape, rays.shape)
    if bg not in BG and bg != "random":
        raise NotImplementedError(bg)
    out = torch.empty(tuple(rays.shape[:-1]) + (Cn,), device=rays.device, dtype=torch.float32)
    alpha = torch.empty_like(density) if want_weights else None
    weights = torch.empty_like(density) if want_weights else None
    if bg == "random":
        rand = _f32(rand, "rand")
        assert rand.numel() == R, (rand.shape, rays.shape)
        check(lib.na_composite_random_bg(_ptr(density), _ptr(feat), _ptr(ts), _ptr(rays), T, R, Cn, 0 if softplus else 1,
                                         _ptr(rand), _ptr(alpha), _ptr(weights), _ptr(out), _stream()))
        return out, alpha, weights
    check(lib.na_composite(_ptr(density), _ptr(feat), _ptr(ts), _ptr(rays), T, R, Cn, 0 if softplus else 1, BG[bg],
                           _ptr(alpha), _ptr(weights), _ptr(out), _stream()))
    return out, alpha, weights


def sky_random(weights: torch.Tensor, rand: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[..., :] += rand[..., 0] * (1 - sum(weights[:-1])) in place (src/nerf.py:101-103); weights [T,...], rand [...,1]."""
    lib = _lib.load()
    weights, rand = _f32(weights, "weights"), _f32(rand, "rand")
    assert out.is_contiguous() and out.dtype == torch.float32 and out.is_cuda
    T = weights.shape[0]
    R = weights.numel() // T
    assert rand.numel() == R and out.numel() % R == 0, (weights.shape, rand.shape, out.shape)
    check(lib.na_sky_random(_ptr(weights), _ptr(rand), T, R, out.numel() // R, _ptr(out), _stream()))
    return out


def integrate(weights: torch.Tensor, other: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    weights, other = _f32(weights, "weights"), _f32(other, "other")
    T = weights.shape[0]
    R = weights.numel() // T
    Cn = other.shape[-1]
    out = torch.empty(tuple(weights.shape[1:]) + (Cn,), device=weights.device, dtype=torch.float32)
    check(lib.na_integrate(_ptr(weights), _ptr(other), T, R, Cn, _ptr(out), _stream()))
    return out


def normalize3(v: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    v = _f32(v, "v")
    out = torch.empty_like(v)
    check(lib.na_normalize3(_ptr(v), v.numel() // 3, _ptr(out), _stream()))
    return out


def pos_linear_combine(lin: torch.Tensor, pos: torch.Tensor, C_out: int) -> torch.Tensor:
    """(sigmoid(lin)/2 + 0.5) * pos[..., :C_out]; pos may be wider (its first C_out columns are used, by row pitch)."""
    if torch.is_grad_enabled() and (lin.requires_grad or pos.requires_grad):
        raise RuntimeError("ops.pos_linear_combine is not differentiable: go through autograd.PosLinearCombineFn")
    lib = _lib.load()
    lin = _f32(lin, "lin")
    pos2, ld = _rows(pos, pos.shape[-1], "pos")
    N = lin.numel()
    assert pos2.shape[0] == N and C_out <= pos.shape[-1]
    out = torch.empty(tuple(lin.shape[:-1]) + (C_out,), device=lin.device, dtype=torch.float32)
    check(lib.na_pos_linear_combine(_ptr(lin), _ptr(pos2), ld, N, C_out, _ptr(out), _stream()))
    return out


def pos_linear_combine_backward(lin: torch.Tensor, pos: torch.Tensor, g: torch.Tensor, C_out: int, want_lin=True,
                                want_pos=True):
    """Gradients of pos_linear_combine w.r.t. lin [...,1] and pos [...,W] (columns >= C_out get zeros)."""
    lib = _lib.load()
    lin, g = _f32(lin, "lin"), _f32(g, "g")
    pos2, ld = _rows(pos, pos.shape[-1], "pos")
    N = lin.numel()
    g_lin = torch.empty_like(lin) if want_lin else None
    g_pos = torch.empty(pos.shape, device=pos.device, dtype=torch.float32) if want_pos else None
    check(lib.na_pos_linear_combine_backward(_ptr(lin), _ptr(pos2), ld, _ptr(g), N, C_out,
                                             _ptr(g_lin) if want_lin else None, _ptr(g_pos) if want_pos else None,
                                             pos.shape[-1], _stream()))
    return g_lin, g_pos


def laplace_density(sdf: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    sdf = _f32(sdf, "sdf")
    beta = _f32(beta.reshape(1), "beta")
    out = torch.empty_like(sdf)
    check(lib.na_laplace_density(_ptr(sdf), sdf.numel(), _ptr(beta), _ptr(out), _stream()))
    return out


def bezier_warp(est: torch.Tensor, pts: torch.Tensor, t: torch.Tensor, n_ctrl: int, n_rl: int = 0):
    """est [...,>=1+3n] (rigidity | control points), pts [...,3], t [...] -> (pts', dp, rigidity); with n_rl > 0 (DynamicNeRF's
    refl_latent: est [..., >= 2 + (3 + n_rl) n] = ... | enc_rigidity | latent control rows) -> (pts', dp, rigidity, refl_latent [..., n_rl])."""
    lib = _lib.load()
    est, pts, t = _f32(est, "est"), _f32(pts, "pts"), _f32(t, "t")
    N = pts.numel() // 3
    assert t.numel() == N and est.numel() // est.shape[-1] == N
    out = torch.empty_like(pts)
    dp = torch.empty_like(pts)
    rig = torch.empty(tuple(pts.shape[:-1]) + (1,), device=pts.device, dtype=torch.float32)
    if n_rl > 0:
        enc = torch.empty(tuple(pts.shape[:-1]) + (n_rl,), device=pts.device, dtype=torch.float32)
        check(lib.na_bezier_warp_latent(_ptr(est), est.shape[-1], _ptr(pts), _ptr(t), N, n_ctrl, n_rl, _ptr(out), _ptr(dp), _ptr(rig),
                                        _ptr(enc), _stream()))
        return out, dp, rig, enc
    check(lib.na_bezier_warp(_ptr(est), est.shape[-1], _ptr(pts), _ptr(t), N, n_ctrl, _ptr(out), _ptr(dp), _ptr(rig),
                             _stream()))
    return out, dp, rig


# ------------------------------------------------------------------------------------------------- SDF marching
def ray_points(r_o: torch.Tensor, r_d: torch.Tensor, t) -> torch.Tensor:
    """r_o + r_d * t; t a float or a per-ray tensor [..., 1] / [...]."""
    lib = _lib.load()
    r_o, r_d = _f32(r_o, "r_o"), _f32(r_d, "r_d")
    R = r_o.numel() // 3
    pts = torch.empty_like(r_o)
    if torch.is_tensor(t):
        t = _f32(t, "t")
        assert t.numel() == R
        check(lib.na_ray_points(_ptr(r_o), _ptr(r_d), _ptr(t), 0.0, R, _ptr(pts), _stream()))
    else:
        check(lib.na_ray_points(_ptr(r_o), _ptr(r_d), None, float(t), R, _ptr(pts), _stream()))
    return pts


def compact_rays(live: torch.Tensor, idx: torch.Tensor, count: torch.Tensor) -> int:
    """idx[:n] = ascending indices of the rays with live != 0 (uint8 mask of any shape); returns n -- one host read of the
    device counter (the reference's boolean-mask indexing synchronises at the same point).  idx: int32 [R + 256], count: int32 [1]."""
    lib = _lib.load()
    assert live.dtype == torch.uint8 and live.is_contiguous() and idx.dtype == torch.int32 and idx.numel() >= live.numel() + 256
    check(lib.na_compact_rays(_ptr(live), live.numel(), _ptr(idx), _ptr(count), _stream()))
    return int(count.item())


def ray_points_indexed(r_o, r_d, t_ray, idx, n: int) -> torch.Tensor:
    """[n,3] positions r_o + r_d * t of the compacted rays idx[:n]"""
    lib = _lib.load()
    pts = torch.empty(n, 3, device=r_o.device, dtype=torch.float32)
    check(lib.na_ray_points_indexed(_ptr(r_o), _ptr(r_d), _ptr(t_ray), _ptr(idx), n, _ptr(pts), _stream()))
    return pts


def sphere_march_update_indexed(sdf, idx, n: int, eps: float, far: float, dist, hits, rem):
    lib = _lib.load()
    sdf, stride = _sdf_col(sdf)
    check(lib.na_sphere_march_update_indexed(_ptr(sdf), stride, _ptr(idx), n, float(eps), float(far), _ptr(dist), _ptr(hits),
                                             _ptr(rem), _stream()))


def bisection_update_indexed(sdf_mid, idx, n: int, eps: float, low, high, sdf_low, sdf_high, z, todo):
    lib = _lib.load()
    sdf_mid, stride = _sdf_col(sdf_mid)
    check(lib.na_bisection_update_indexed(_ptr(sdf_mid), stride, _ptr(idx), n, float(eps), _ptr(low), _ptr(high), _ptr(sdf_low),
                                          _ptr(sdf_high), _ptr(z), _ptr(todo), _stream()))


def _sdf_col(sdf: torch.Tensor):
    sdf = _f32(sdf, "sdf")
    return sdf, (sdf.shape[-1] if sdf.dim() > 1 else 1)


def sphere_march_update(sdf, eps: float, far: float, dist, hits, rem):
    lib = _lib.load()
    sdf, stride = _sdf_col(sdf)
    check(lib.na_sphere_march_update(_ptr(sdf), stride, dist.numel(), float(eps), float(far), _ptr(dist), _ptr(hits),
                                     _ptr(rem), _stream()))


def sign_change_update(sdf, step: int, curr_min, idxs, last_pos, first_neg):
    lib = _lib.load()
    sdf, stride = _sdf_col(sdf)
    check(lib.na_sign_change_update(_ptr(sdf), stride, curr_min.numel(), int(step), _ptr(curr_min), _ptr(idxs),
                                    _ptr(last_pos), _ptr(first_neg), _stream()))


def bisection_update(sdf_mid, eps: float, low, high, sdf_low, sdf_high, z, todo):
    lib = _lib.load()
    stride = 1
    if sdf_mid is not None:
        sdf_mid, stride = _sdf_col(sdf_mid)
    check(lib.na_bisection_update(_ptr(sdf_mid), stride, low.numel(), float(eps), _ptr(low), _ptr(high), _ptr(sdf_low),
                                  _ptr(sdf_high), _ptr(z), _ptr(todo), _stream()))


# ------------------------------------------------------------------------------------------------- MLP


# ---------------------------------------------------------------------------------------- N4 lights / occlusion
def point_light(x: torch.Tensor, center: torch.Tensor, intensity: torch.Tensor, distance_decay: bool = True):
    """src/lights.py:118-132: (unit direction to the light, distance [..., 1], spectrum [..., 3]) for points x [..., 3];
    center / intensity hold one light ([3]) or one per point (x.shape)."""
    lib = _lib.load()
    x, center, intensity = _f32(x, "x"), _f32(center, "center"), _f32(intensity, "intensity")
    N = x.numel() // 3
    strides = []
    for t, name in ((center, "center"), (intensity, "intensity")):
        assert t.numel() in (3, 3 * N), f"{name}: one light or one per point"
        strides.append(3 if (t.numel() == 3 * N and N > 1) else 0)
    d = torch.empty_like(x)
    dist = torch.empty(tuple(x.shape[:-1]) + (1,), device=x.device, dtype=torch.float32)
    spectrum = torch.empty_like(x)
    check(lib.na_point_light(_ptr(x), _ptr(center), strides[0], _ptr(intensity), strides[1], int(bool(distance_decay)), N,
                             _ptr(d), _ptr(dist), _ptr(spectrum), _stream()))
    return d, dist, spectrum


def occlusion_apply(spectrum: torch.Tensor, visible: Optional[torch.Tensor] = None, raw_att: Optional[torch.Tensor] = None,
                    att_mode: int = 0, hidden_value: float = 0.0) -> torch.Tensor:
    """spectrum * a(raw_att) * v(visible): see include/nerf_atlas_amd.h (src/renderers.py:40-45,65-67,82-83,118-121)."""
    lib = _lib.load()
    spectrum = _f32(spectrum, "spectrum")
    N = spectrum.numel() // 3
    vis = None
    if visible is not None:
        assert visible.dtype == torch.bool and visible.numel() == N and visible.is_cuda
        vis = visible.contiguous().view(torch.uint8)
    if raw_att is not None:
        raw_att = _f32(raw_att, "raw_att")
        assert raw_att.numel() == N
    out = torch.empty_like(spectrum)
    check(lib.na_occlusion_apply(_ptr(spectrum), _ptr(vis), _ptr(raw_att), int(att_mode), float(hidden_value), N, _ptr(out),
                                 _stream()))
    return out


def linear_f32(x0: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], pre_act: str = "none",
               x1: Optional[torch.Tensor] = None, split_bf16: bool = False, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = W . act([x0 | x1]) + b: exact fp32 (f32 MFMA), or with split_bf16 the 3-product bf16 split used by the
    training step (relative error ~2^-16, 5x the matrix-core rate).  packed: W as train_pack_many left it (split_bf16 only)."""
    lib = _lib.load()
    x0, W = _f32(x0, "x0"), _f32(W, "W")
    N = x0.shape[0]
    in0 = x0.shape[1]
    in1 = 0
    if x1 is not None:
        x1 = _f32(x1, "x1")
        in1 = x1.shape[1]
    assert W.shape[1] == in0 + in1, (W.shape, in0, in1)
    if b is not None:
        b = _f32(b, "b")
    y = torch.empty(N, W.shape[0], device=x0.device, dtype=torch.float32)
    if split_bf16 and packed is not None:
        check(lib.na_linear_bf16x3_pk(_ptr(x0), in0, _ptr(x1), in1, N, _ptr(packed), _ptr(b), W.shape[0], ACT[pre_act], _ptr(y),
                                      _stream()))
        return y
    fn = lib.na_linear_bf16x3 if split_bf16 else lib.na_linear_f32
    check(fn(_ptr(x0), in0, _ptr(x1), in1, N, _ptr(W), _ptr(b), W.shape[0], ACT[pre_act], _ptr(y), _stream()))
    return y


def train_gemm_packed_ok(N: int, M: int) -> bool:
    """Does a batch of N rows with M output columns run the training GEMMs that take packed operands (na_train_gemm_packed_ok)?"""
    return bool(_lib.load().na_train_gemm_packed_ok(int(N), int(M)))


def train_pack_many(mats):
    """ONE launch packs every B operand of a training step (na_train_pack_many).  mats: [(W [rows, cols] fp32 contiguous,
    transposed)] -- the operand is W itself (a forward's [out, in]) or, with transposed, W^T (an input gradient's [in, out], read
    straight from W).  Returns the packed operands as uint8 views of one buffer (None where the shape has no packed form)."""
    lib = _lib.load()
    n = len(mats)
    if n == 0:
        return []
    Ws = [_f32(w.detach(), "W") for w, _ in mats]
    Ms = [int(w.shape[1] if t else w.shape[0]) for (w, t) in mats]
    Ks = [int(w.shape[0] if t else w.shape[1]) for (w, t) in mats]
    sizes = [int(lib.na_train_packed_bytes(m, k)) for m, k in zip(Ms, Ks)]
    offs, total = [], 0
    for sz in sizes:
        offs.append(total)
        total += (sz + 255) & ~255
    buf = torch.empty(max(total, 16), device=Ws[0].device, dtype=torch.uint8)
    views = [buf[o:o + sz] if sz else None for o, sz in zip(offs, sizes)]
    idx = [i for i, sz in enumerate(sizes) if sz]
    if idx:
        k = len(idx)
        check(lib.na_train_pack_many(k, (C.c_void_p * k)(*[Ws[i].data_ptr() for i in idx]), (C.c_int * k)(*[Ms[i] for i in idx]),
                                     (C.c_int * k)(*[Ks[i] for i in idx]), (C.c_int * k)(*[int(Ws[i].shape[1]) for i in idx]),
                                     (C.c_int * k)(*[int(bool(mats[i][1])) for i in idx]),
                                     (C.c_void_p * k)(*[views[i].data_ptr() for i in idx]), _stream()))
    return views


def linear_dgrad(dY: torch.Tensor, W: torch.Tensor, x0: torch.Tensor, pre_act: str = "none",
                 x1: Optional[torch.Tensor] = None, want0: bool = True, want1: bool = True, packed_t: Optional[torch.Tensor] = None):
    """(g_x0 [N,in0] | None, g_x1 [N,in1] | None) = (dY . W) * act'([x0|x1]) for y = W . act([x0|x1]) + b.
    packed_t: W^T as train_pack_many left it (no transposing copy, no pack launch)."""
    lib = _lib.load()
    dY, W, x0 = _f32(dY, "dY"), _f32(W, "W"), _f32(x0, "x0")
    N, in0 = x0.shape
    in1 = 0
    if x1 is not None:
        x1 = _f32(x1, "x1")
        in1 = x1.shape[1]
    out = dY.shape[1]
    assert W.shape == (out, in0 + in1) and dY.shape[0] == N
    g0 = torch.empty_like(x0) if want0 else None
    g1 = torch.empty_like(x1) if (want1 and x1 is not None) else None
    if g0 is None and g1 is None:
        return None, None
    if packed_t is not None:
        check(lib.na_linear_dgrad_bf16x3_pk(_ptr(dY), out, N, _ptr(packed_t), _ptr(x0), in0, _ptr(x1), in1, ACT[pre_act], _ptr(g0),
                                            _ptr(g1), _stream()))
        return g0, g1
    Wt = W.t().contiguous()  # [in, out]: the K-contiguous operand of the input-gradient GEMM (<= 0.6 MB)
    check(lib.na_linear_dgrad_bf16x3(_ptr(dY), out, N, _ptr(Wt), _ptr(x0), in0, _ptr(x1), in1, ACT[pre_act], _ptr(g0), _ptr(g1),
                              _stream()))
    return g0, g1


def linear_bwd_fused_ok(N: int, out: int, in0: int) -> bool:
    """Does (N, out, in0) run the one-pass input-gradient + weight-gradient kernel (na_linear_bwd_fused_ok)?"""
    return bool(_lib.load().na_linear_bwd_fused_ok(int(N), int(out), int(in0)))


def linear_bwd_fused(dY: torch.Tensor, x0: torch.Tensor, pre_act: str, packed_t: torch.Tensor, in1: int = 0, want_bias: bool = True,
                     dW: Optional[torch.Tensor] = None, col0: int = 0):
    """(g_x [N,in0], dW, db | None) of one source x0 of y = W . act([.. x0 ..]) + b in one pass over dY and x0
    (na_linear_bwd_bf16x3_pk): in0 = 256, or a narrow source of <= 128 columns.  packed_t: W^T as train_pack_many left it.
    dW None: a fresh [out, in0 + in1] buffer whose columns 0..in0-1 are WRITTEN (in1 columns of a second source left to the caller);
    dW given: its columns col0..col0+in0-1 are written (the second source of a skip layer: col0 = the first source's width)."""
    lib = _lib.load()
    dY, x0 = _f32(dY, "dY"), _f32(x0, "x0")
    N, in0 = x0.shape
    out = dY.shape[1]
    db = None
    if dW is None:
        assert col0 == 0
        ld = in0 + in1
        nW = out * ld
        pad = (-nW) % 4
        acc = torch.empty(nW + pad + (out if want_bias else 0), device=x0.device, dtype=torch.float32)
        dW = acc[:nW].view(out, ld)
        db = acc[nW + pad:] if want_bias else None
    else:
        assert dW.is_contiguous() and dW.shape[0] == out and col0 % 64 == 0 and col0 + in0 <= dW.shape[1]
    g0 = torch.empty_like(x0)
    wp = packed_t.data_ptr() + int(lib.na_train_packed_row_offset(col0, out))
    # (the partial gradients' workspace from torch's allocator: stream-ordered reuse for microseconds; hipMallocAsync inside the
    # call cost 230 us of host time)
    ws = torch.empty(int(lib.na_linear_bwd_workspace_bytes(N, in0)), device=x0.device, dtype=torch.uint8)
    check(lib.na_linear_bwd_bf16x3_pk(_ptr(dY), out, N, wp, _ptr(x0), in0, ACT[pre_act], _ptr(g0), dW.data_ptr() + 4 * col0, dW.shape[1],
                                      _ptr(db), _ptr(ws), _stream()))
    return g0, dW, db


def linear_bwd_partials(dY: torch.Tensor, x0: torch.Tensor, pre_act: str, packed_t: torch.Tensor, col0: int = 0, want_bias: bool = True,
                        add: Optional[torch.Tensor] = None):
    """The one-pass backward of a source WITHOUT its reduction (na_linear_bwd_partials_bf16x3_pk): (g_x [N, in0], workspace holding
    the partial gradients, number of partials).  train_reduce_many sums the partials of many Linears in one launch."""
    lib = _lib.load()
    dY, x0 = _f32(dY, "dY"), _f32(x0, "x0")
    N, in0 = x0.shape
    out = dY.shape[1]
    nbytes = int(lib.na_linear_bwd_workspace_bytes(N, in0))
    ws = torch.empty(nbytes, device=x0.device, dtype=torch.uint8)
    g0 = torch.empty_like(x0)
    wp = packed_t.data_ptr() + int(lib.na_train_packed_row_offset(col0, out))
    if add is not None:  # (another consumer's gradient of x0: summed into g_x inside the kernel; narrow sources / outputs only)
        add = _f32(add, "add")
        assert add.shape == x0.shape
    check(lib.na_linear_bwd_partials_bf16x3_pk(_ptr(dY), out, N, wp, _ptr(x0), in0, ACT[pre_act], _ptr(g0), _ptr(add), int(want_bias),
                                               _ptr(ws), _stream()))
    return g0, ws, int(lib.na_linear_bwd_partial_count(N, in0))


def train_reduce_many(entries):
    """entries: [(workspace, partials, out, in, dW [out, ld] contiguous, col0, db | None)]: dW[:, col0:col0+in] and db WRITTEN, all
    entries by one launch (na_train_reduce_many)."""
    lib = _lib.load()
    n = len(entries)
    if n == 0:
        return
    vp, ip = C.c_void_p * n, C.c_int * n
    check(lib.na_train_reduce_many(
        n, vp(*[e[0].data_ptr() for e in entries]), ip(*[int(e[1]) for e in entries]), ip(*[int(e[2]) for e in entries]),
        ip(*[int(e[3]) for e in entries]), ip(*[int(e[4].shape[1]) for e in entries]),
        vp(*[e[4].data_ptr() + 4 * int(e[5]) for e in entries]), vp(*[(e[6].data_ptr() if e[6] is not None else None) for e in entries]),
        _stream()))


def adam_step(params, grads, exp_avgs, exp_avg_sqs, one_minus_beta1: float, beta2: float, one_minus_beta2: float,
              bias_correction2_sqrt: float, eps: float, neg_step_size: float, fma_mask: int):
    """torch's foreach Adam update of every tensor by ONE launch (na_adam_step): lists of contiguous fp32 device tensors."""
    lib = _lib.load()
    n = len(params)
    if n == 0:
        return
    for t in (*params, *grads, *exp_avgs, *exp_avg_sqs):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("adam_step: contiguous fp32 device tensors only")
    vp, ip = C.c_void_p * n, C.c_int64 * n
    check(lib.na_adam_step(n, vp(*[t.data_ptr() for t in params]), vp(*[t.data_ptr() for t in grads]), vp(*[t.data_ptr() for t in exp_avgs]),
                           vp(*[t.data_ptr() for t in exp_avg_sqs]), ip(*[t.numel() for t in params]), float(one_minus_beta1), float(beta2),
                           float(one_minus_beta2), float(bias_correction2_sqrt), float(eps), float(neg_step_size), int(fma_mask), _stream()))


def linear_wgrad_cols(x: torch.Tensor, dY: torch.Tensor, pre_act: str, dW: torch.Tensor, col0: int):
    """dW[:, col0:col0 + x.shape[1]] = dY^T . act(x) WRITTEN (na_linear_wgrad_bf16x3_cols): one source of a concatenated input."""
    lib = _lib.load()
    x, dY = _f32(x, "x"), _f32(dY, "dY")
    N, k = x.shape
    out = dY.shape[1]
    assert dW.is_contiguous() and dW.shape[0] == out and col0 + k <= dW.shape[1]
    check(lib.na_linear_wgrad_bf16x3_cols(_ptr(x), k, N, _ptr(dY), out, ACT[pre_act], dW.data_ptr() + 4 * col0, dW.shape[1], None,
                                          _stream()))
    return dW


# ------------------------------------------------------------------------------------------------- backward
def act_backward(x: torch.Tensor, g: torch.Tensor, act: str) -> torch.Tensor:
    lib = _lib.load()
    x, g = _f32(x, "x"), _f32(g, "g")
    assert x.shape == g.shape
    out = torch.empty_like(x)
    check(lib.na_act_backward(_ptr(x), _ptr(g), x.numel(), ACT[act], _ptr(out), _stream()))
    return out


def sigmoid_backward(x: torch.Tensor, g: torch.Tensor, kind: str) -> torch.Tensor:
    lib = _lib.load()
    x, g = _f32(x, "x"), _f32(g, "g")
    out = torch.empty_like(x)
    check(lib.na_sigmoid_backward(_ptr(x), _ptr(g), x.numel(), SIGMOID[kind], _ptr(out), _stream()))
    return out


def linear_wgrad(x0: torch.Tensor, dY: torch.Tensor, pre_act: str = "none", x1: Optional[torch.Tensor] = None,
                 want_bias: bool = True, split_bf16: bool = False):
    """(dW [out, in0+in1], db [out]) for y = W . act([x0|x1]) + b; exact fp32 or the split-bf16 GEMM."""
    lib = _lib.load()
    x0, dY = _f32(x0, "x0"), _f32(dY, "dY")
    N, in0 = x0.shape
    in1 = 0
    if x1 is not None:
        x1 = _f32(x1, "x1")
        in1 = x1.shape[1]
    out = dY.shape[1]
    # (one buffer for both results: db sits behind dW, on a 16-byte boundary; the split-bf16 entry point WRITES them -- round 5:
    # no zero fill --, the exact-fp32 one accumulates)
    nW = out * (in0 + in1)
    pad = (-nW) % 4
    alloc = torch.empty if split_bf16 else torch.zeros
    acc = alloc(nW + pad + (out if want_bias else 0), device=x0.device, dtype=torch.float32)
    dW = acc[:nW].view(out, in0 + in1)
    db = acc[nW + pad:] if want_bias else None
    fn = lib.na_linear_wgrad_bf16x3_ow if split_bf16 else lib.na_linear_wgrad
    check(fn(_ptr(x0), in0, _ptr(x1), in1, N, _ptr(dY), out, ACT[pre_act], _ptr(dW), _ptr(db), _stream()))
    return dW, db


def hash_encode_backward(x: torch.Tensor, g_out: torch.Tensor, include_input: bool = True) -> torch.Tensor:
    lib = _lib.load()
    x, g_out = _f32(x, "x"), _f32(g_out, "g_out")
    N = x.numel() // 3
    tg = torch.zeros(8, 65536, 4, device=x.device, dtype=torch.float32)
    check(lib.na_hash_encode_backward(_ptr(x), N, _ptr(g_out), int(include_input), _ptr(tg), _stream()))
    return tg


def hash_encode_backward_input(x: torch.Tensor, tables: torch.Tensor, g_out: torch.Tensor,
                               include_input: bool = True) -> torch.Tensor:
    lib = _lib.load()
    x, tables, g_out = _f32(x, "x"), _f32(tables, "tables"), _f32(g_out, "g_out")
    N = x.numel() // 3
    gx = torch.empty_like(x)
    check(lib.na_hash_encode_backward_input(_ptr(x), N, _ptr(tables), _ptr(g_out), int(include_input), _ptr(gx),
                                            _stream()))
    return gx


def hash_encode_jvp(x: torch.Tensor, tables: torch.Tensor, tangent: torch.Tensor, include_input: bool = True) -> torch.Tensor:
    """d hash_encode(x)/dx . tangent, rows [tangent | per-level features] like hash_encode's output."""
    lib = _lib.load()
    x, tables, tangent = _f32(x, "x"), _f32(tables, "tables"), _f32(tangent, "tangent")
    assert tangent.shape == x.shape and x.shape[-1] == 3 and tuple(tables.shape) == (8, 65536, 4)
    N = x.numel() // 3
    out = torch.empty(tuple(x.shape[:-1]) + (32 + 3 * int(include_input),), device=x.device, dtype=torch.float32)
    check(lib.na_hash_encode_jvp(_ptr(x), N, _ptr(tables), _ptr(tangent), int(include_input), _ptr(out), _stream()))
    return out


def hash_encode_jvp_backward(x: torch.Tensor, tangent: torch.Tensor, g_t: torch.Tensor, include_input: bool = True) -> torch.Tensor:
    """d <g_t, J(x).tangent> / d tables -> [8, 65536, 4]."""
    lib = _lib.load()
    x, tangent, g_t = _f32(x, "x"), _f32(tangent, "tangent"), _f32(g_t, "g_t")
    N = x.numel() // 3
    assert g_t.numel() == N * (32 + 3 * int(include_input))
    grad = torch.zeros(8, 65536, 4, device=x.device, dtype=torch.float32)
    check(lib.na_hash_encode_jvp_backward(_ptr(x), _ptr(tangent), N, _ptr(g_t), int(include_input), _ptr(grad), _stream()))
    return grad


def ffjord_div(est: torch.Tensor, est_tangent: torch.Tensor, t: torch.Tensor, e: torch.Tensor, n_ctrl: int) -> torch.Tensor:
    """<e, d(rigid_dp)/dx . e> per point (runner.py:697-700): est / est_tangent [..., S], t [...], e [..., 3]."""
    lib = _lib.load()
    est, est_tangent, t, e = _f32(est, "est"), _f32(est_tangent, "est_tangent"), _f32(t, "t"), _f32(e, "e")
    S = est.shape[-1]
    N = est.numel() // S
    assert est_tangent.shape == est.shape and t.numel() == N and e.numel() == 3 * N
    out = torch.empty(est.shape[:-1], device=est.device, dtype=torch.float32)
    check(lib.na_ffjord_div(_ptr(est), _ptr(est_tangent), S, _ptr(t), _ptr(e), N, int(n_ctrl), _ptr(out), _stream()))
    return out


def laplace_density_backward(sdf: torch.Tensor, beta: torch.Tensor, g: torch.Tensor, want_beta: bool = True):
    """-> (g_sdf, g_beta [1] or None)."""
    lib = _lib.load()
    sdf, g = _f32(sdf, "sdf"), _f32(g, "g")
    beta = _f32(beta.reshape(1), "beta")
    g_sdf = torch.empty_like(sdf)
    g_beta = torch.zeros(1, device=sdf.device, dtype=torch.float32) if want_beta else None
    check(lib.na_laplace_density_backward(_ptr(sdf), sdf.numel(), _ptr(beta), _ptr(g), _ptr(g_sdf),
                                          _ptr(g_beta) if want_beta else None, _stream()))
    return g_sdf, g_beta


def bezier_warp_backward(est: torch.Tensor, t: torch.Tensor, n_ctrl: int, g_pts=None, g_dp=None, g_rig=None, n_rl: int = 0, g_enc=None):
    lib = _lib.load()
    est, t = _f32(est, "est"), _f32(t, "t")
    N = t.numel()
    g_pts = None if g_pts is None else _f32(g_pts, "g_pts")
    g_dp = None if g_dp is None else _f32(g_dp, "g_dp")
    g_rig = None if g_rig is None else _f32(g_rig, "g_rig")
    g_est = torch.empty_like(est)
    opt = lambda g: None if g is None else _ptr(g)
    if n_rl > 0:
        g_enc = None if g_enc is None else _f32(g_enc, "g_enc")
        assert g_enc is None or g_enc.numel() == N * n_rl
        check(lib.na_bezier_warp_latent_backward(_ptr(est), est.shape[-1], _ptr(t), N, n_ctrl, n_rl, opt(g_pts), opt(g_dp), opt(g_rig),
                                                 opt(g_enc), _ptr(g_est), _stream()))
        return g_est
    check(lib.na_bezier_warp_backward(_ptr(est), est.shape[-1], _ptr(t), N, n_ctrl, opt(g_pts), opt(g_dp), opt(g_rig), _ptr(g_est),
                                      _stream()))
    return g_est


def composite_backward(density, feat, ts, rays, g_out, softplus: bool = True, bg: str = "black", rand=None):
    lib = _lib.load()
    density, feat, ts, rays, g_out = (_f32(density, "density"), _f32(feat, "feat"), _f32(ts, "ts"), _f32(rays, "rays"),
                                      _f32(g_out, "g_out"))
    T = ts.shape[0]
    Cn = feat.shape[-1]
    R = rays.numel() // 6
    gd = torch.empty_like(density)
    gf = torch.empty_like(feat)
    if bg == "random":
        rand = _f32(rand, "rand")
        check(lib.na_composite_random_bg_backward(_ptr(density), _ptr(feat), _ptr(ts), _ptr(rays), T, R, Cn, 0 if softplus else 1,
                                                  _ptr(rand), _ptr(g_out), _ptr(gd), _ptr(gf), _stream()))
        return gd, gf
    check(lib.na_composite_backward(_ptr(density), _ptr(feat), _ptr(ts), _ptr(rays), T, R, Cn, 0 if softplus else 1, BG[bg],
                                    _ptr(g_out), _ptr(gd), _ptr(gf), _stream()))
    return gd, gf


def make_desc(in_size, enc_kind, enc_dims, latent_size, num_layers, hidden, out_size, skip, activation,
              layout="generic") -> NaMlpDesc:
    return NaMlpDesc(in_size, ENC[enc_kind], enc_dims, latent_size, num_layers, hidden, out_size, skip,
                     ACT[activation], LAYOUT[layout])


def mlp_packed_bytes(desc: NaMlpDesc, precision: str) -> int:
    return int(_lib.load().na_mlp_packed_bytes(C.byref(desc), PREC[precision]))


def mlp_pack(desc: NaMlpDesc, precision: str, weights: Sequence[torch.Tensor],
             biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """weights/biases in the order init, layers[0..L-1], out (nn.Linear layout).  Returns the packed stream."""
    lib = _lib.load()
    nbytes = mlp_packed_bytes(desc, precision)
    if nbytes == 0:
        raise _lib.NaError(-3, "this SkipConnMLP shape has no MFMA kernel (use linear_f32 per layer)")
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [_f32(b.detach(), "bias") for b in biases]
    n = desc.num_layers + 2
    assert len(ws) == n and len(bs) == n
    packed = torch.empty(nbytes, device=ws[0].device, dtype=torch.uint8)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
    check(lib.na_mlp_pack(C.byref(desc), PREC[precision], wp, bp, _ptr(packed), _stream()))
    return packed


def _rows(t: torch.Tensor, width: int, name: str):
    """[N, width] rows of a float32 CUDA tensor WITHOUT copying when it is a column slice of a wider row-major buffer
    (unit stride along the last dim, one constant row pitch); returns (tensor, pitch in floats)."""
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA(HIP) tensor: the hot path has no CPU implementation")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    v = t.reshape(-1, width)  # a view whenever the strides allow it
    if v.numel() == 0 or (v.stride(1) == 1 and v.stride(0) >= width):
        return v, (v.stride(0) if v.shape[0] > 1 else max(v.stride(0), width))
    v = v.contiguous()
    return v, width


def mlp_forward(desc: NaMlpDesc, precision: str, packed: torch.Tensor, p: torch.Tensor,
                latent: Optional[torch.Tensor] = None, enc_params: Optional[torch.Tensor] = None, mip=None) -> torch.Tensor:
    """Fused SkipConnMLP forward.  p [..., in_size] and latent [..., latent_size] may be column slices of wider
    buffers (e.g. `first_out[..., 1:]`): their row pitch is passed down instead of making them contiguous.
    mip = (rays [B,H,W,6], ts [T], kind, t_end, min_deg, max_deg): the leading 6*(max_deg-min_deg) latent columns are the
    IPE of the samples, generated in the kernel's prologue; `latent` then holds only the remaining columns (or is None)."""
    lib = _lib.load()
    p2, p_ld = _rows(p, desc.in_size, "p")
    N = p2.shape[0]
    gen = 0
    mdesc = None
    if mip is not None:
        rays, ts, kind, t_end, min_deg, max_deg = mip
        rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
        assert rays.dim() == 4 and rays.shape[-1] == 6
        B, H, W = rays.shape[:3]
        gen = 6 * (max_deg - min_deg)
        mdesc = _lib.NaMipDesc(rays.data_ptr(), ts.data_ptr(), B, H, W, ts.shape[0], MIP_KIND[kind], min_deg, max_deg,
                               float(t_end))
    lat2, lat_ld = (None, 0)
    if latent is not None:
        lat2, lat_ld = _rows(latent, desc.latent_size - gen, "latent")
        assert lat2.shape[0] == N
    if enc_params is not None:
        enc_params = _f32(enc_params, "enc_params")
    y = torch.empty(tuple(p.shape[:-1]) + (desc.out_size,), device=p.device, dtype=torch.float32)
    check(lib.na_mlp_forward_mip(C.byref(desc), PREC[precision], _ptr(packed), _ptr(p2), p_ld, _ptr(lat2), lat_ld,
                                 _ptr(enc_params), None if mdesc is None else C.cast(C.byref(mdesc), C.c_void_p), N, _ptr(y),
                                 _stream()))
    return y


# ------------------------------------------------------------------------------------------------- fused renderer
def render_plain_view(rays: torch.Tensor, ts: torch.Tensor, hash_tables: torch.Tensor, packed_first: torch.Tensor,
                      packed_view: torch.Tensor, precision: str, sigmoid_kind: str = "thin", bg: str = "black",
                      want_weights: bool = False, workspace: Optional[torch.Tensor] = None,
                      pts: Optional[torch.Tensor] = None):
    """PlainNeRF(view) forward, fully fused (src/nerf.py:326-361).  rays [...,6] -> (rgb [...,3], alpha, weights).
    pts [T,...,3]: explicit sample positions (from_pts with deformed points) instead of o + t d."""
    lib = _lib.load()
    rays, ts, hash_tables = _f32(rays, "rays"), _f32(ts, "ts"), _f32(hash_tables, "hash_tables")
    R = rays.numel() // 6
    T = ts.shape[0]
    if bg not in BG:
        raise NotImplementedError(bg)
    nbytes = int(lib.na_render_workspace_bytes(T, R))
    if workspace is None:
        workspace = torch.empty(nbytes, device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
        check(lib.na_render_plain_view_pts(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(hash_tables), _ptr(packed_first),
                                           _ptr(packed_view), PREC[precision], SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha),
                                           _ptr(weights), _ptr(out), _ptr(workspace), workspace.numel(), _stream()))
        return out, alpha, weights
    check(lib.na_render_plain_view(_ptr(rays), R, _ptr(ts), T, _ptr(hash_tables), _ptr(packed_first), _ptr(packed_view),
                                   PREC[precision], SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights), _ptr(out),
                                   _ptr(workspace), workspace.numel(), _stream()))
    return out, alpha, weights


def render_ls_pack(precision: str, first_wb, view_wb) -> torch.Tensor:
    """Pack PlainNeRF.first and View.mlp ({init, layers.0..3, out} weights and biases each) into the weight stream of
    the layer-synchronous renderer (na_render_ls_pack)."""
    lib = _lib.load()
    (w1, b1), (w2, b2) = first_wb, view_wb
    assert len(w1) == 6 and len(w2) == 6, "the LS renderer is specialised for 4 hidden layers per MLP"
    dev = w1[0].device
    keep = [[_f32(w.detach(), "weight") for w in w1], [None if b is None else _f32(b.detach(), "bias") for b in b1],
            [_f32(w.detach(), "weight") for w in w2], [None if b is None else _f32(b.detach(), "bias") for b in b2]]
    shapes1 = [(256, 38), (256, 294), (256, 256), (256, 256), (256, 256), (65, 256)]
    shapes2 = [(256, 69), (256, 325), (256, 256), (256, 256), (256, 256), (3, 256)]
    for w, shp in zip(keep[0] + keep[2], shapes1 + shapes2):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS renderer: weight shape {tuple(w.shape)} != {shp}")
    arrs = [(C.c_void_p * 6)(*[0 if t is None else t.data_ptr() for t in lst]) for lst in keep]
    packed = torch.empty(int(lib.na_render_ls_packed_bytes(PREC[precision])), device=dev, dtype=torch.uint8)
    check(lib.na_render_ls_pack(PREC[precision], arrs[0], arrs[1], arrs[2], arrs[3], _ptr(packed), _stream()))
    return packed


def render_plain_view_ls(rays: torch.Tensor, ts: torch.Tensor, hash_tables: torch.Tensor, packed: torch.Tensor,
                         precision: str, sigmoid_kind: str = "thin", bg: str = "black", want_weights: bool = False,
                         workspace: Optional[torch.Tensor] = None, pts: Optional[torch.Tensor] = None):
    """PlainNeRF(view) forward on the layer-synchronous engine (same contract as render_plain_view)."""
    lib = _lib.load()
    rays, ts, hash_tables = _f32(rays, "rays"), _f32(ts, "ts"), _f32(hash_tables, "hash_tables")
    R = rays.numel() // 6
    T = ts.shape[0]
    if bg not in BG:
        raise NotImplementedError(bg)
    nbytes = int(lib.na_render_ls_workspace_bytes(T, R))
    if workspace is None:
        workspace = torch.empty(nbytes, device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
    check(lib.na_render_plain_view_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(hash_tables), _ptr(packed),
                                      PREC[precision], SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights),
                                      _ptr(out), _ptr(workspace), workspace.numel(), _stream()))
    _f16x_guard(precision, out, "na_render_plain_view_ls")
    return out, alpha, weights


TRAIN_LS_MAX_ROWS = (1 << 22) - 1  # na_train_plain_view_ls: 32-bit byte offsets into a [N, 256] fp32 plane


def train_plain_view_ls(rays: torch.Tensor, ts: torch.Tensor, pts: torch.Tensor, hash_tables: torch.Tensor, packed: torch.Tensor,
                        sigmoid_kind: str = "thin"):
    """The training forward of PlainNeRF(view) as one launch (na_train_plain_view_ls): returns (planes [10, N, 256], view_rows [N, 69],
    density [N], rgb_pre [N, 3], out [*batch, 3]) with N = T * R, rows t * R + ray.  `packed`: render_ls_pack("bf16x3", ...) of the
    current weights."""
    lib = _lib.load()
    rays, ts, pts, hash_tables = _f32(rays, "rays"), _f32(ts, "ts"), _f32(pts, "pts"), _f32(hash_tables, "hash_tables")
    R, T = rays.numel() // 6, ts.numel()
    N = T * R
    assert pts.numel() == N * 3, (pts.shape, T, R)
    dev = rays.device
    planes = torch.empty((10, N, 256), device=dev, dtype=torch.float32)
    view_rows = torch.empty((N, 69), device=dev, dtype=torch.float32)
    density = torch.empty((N,), device=dev, dtype=torch.float32)
    rgb_pre = torch.empty((N, 3), device=dev, dtype=torch.float32)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=dev, dtype=torch.float32)
    workspace = torch.empty(int(lib.na_render_ls_workspace_bytes(T, R)), device=dev, dtype=torch.uint8)
    check(lib.na_train_plain_view_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(hash_tables), _ptr(packed), SIGMOID[sigmoid_kind],
                                     _ptr(planes), _ptr(view_rows), _ptr(density), _ptr(rgb_pre), _ptr(out), _ptr(workspace),
                                     workspace.numel(), _stream()))
    return planes, view_rows, density, rgb_pre, out


def resample_ts(ts: torch.Tensor, weights: torch.Tensor, n_fine: int, u: Optional[torch.Tensor] = None, want_fine: bool = False):
    """Inverse-cdf resampling of a coarse pass (na_resample_ts): ts [T], weights [T, *batch] -> merged [*batch, T + n_fine] (the
    coarse and the new positions of every ray in increasing order) and, with want_fine, fine [*batch, n_fine].  u: None
    (linspace(0, 1, n_fine)) or [n_fine, *batch] draws."""
    lib = _lib.load()
    ts, weights = _f32(ts, "ts"), _f32(weights, "weights")
    T = ts.shape[0]
    assert weights.shape[0] == T, (weights.shape, T)
    batch = tuple(weights.shape[1:])
    R = weights.numel() // T
    if u is not None:
        u = _f32(u, "u")
        assert tuple(u.shape) == (n_fine,) + batch, (u.shape, n_fine, batch)
    merged = torch.empty(batch + (T + n_fine,), device=ts.device, dtype=torch.float32)
    fine = torch.empty(batch + (n_fine,), device=ts.device, dtype=torch.float32) if want_fine else None
    check(lib.na_resample_ts(_ptr(ts), _ptr(weights), R, T, _ptr(u), n_fine, _ptr(fine), _ptr(merged), _stream()))
    return (merged, fine) if want_fine else merged


def render_plain_view_ls_rayts(rays: torch.Tensor, ts_ray: torch.Tensor, hash_tables: torch.Tensor, packed: torch.Tensor,
                               precision: str, sigmoid_kind: str = "thin", bg: str = "black", want_weights: bool = False,
                               workspace: Optional[torch.Tensor] = None):
    """render_plain_view_ls with per-ray steps ts_ray [*batch, T] (increasing along T)."""
    lib = _lib.load()
    rays, ts_ray, hash_tables = _f32(rays, "rays"), _f32(ts_ray, "ts_ray"), _f32(hash_tables, "hash_tables")
    R = rays.numel() // 6
    T = ts_ray.shape[-1]
    assert ts_ray.numel() == R * T, (ts_ray.shape, R)
    if bg not in BG:
        raise NotImplementedError(bg)
    nbytes = int(lib.na_render_ls_workspace_bytes(T, R))
    if workspace is None:
        workspace = torch.empty(nbytes, device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    check(lib.na_render_plain_view_ls_rayts(_ptr(rays), R, _ptr(ts_ray), T, _ptr(hash_tables), _ptr(packed), PREC[precision],
                                            SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights), _ptr(out), _ptr(workspace),
                                            workspace.numel(), _stream()))
    _f16x_guard(precision, out, "na_render_plain_view_ls_rayts")
    return out, alpha, weights


def render_plain_mip_ls_pack(precision: str, first_wb, view_wb) -> torch.Tensor:
    """PlainNeRF.first and View.mlp of a mip model (96 IPE latent columns in both) as one weight stream of the
    layer-synchronous renderer (na_render_plain_mip_ls_pack; f16x only)."""
    lib = _lib.load()
    (w1, b1), (w2, b2) = first_wb, view_wb
    assert len(w1) == 6 and len(w2) == 6, "the LS renderer is specialised for 4 hidden layers per MLP"
    dev = w1[0].device
    keep = [[_f32(w.detach(), "weight") for w in w1], [_f32(b.detach(), "bias") for b in b1],
            [_f32(w.detach(), "weight") for w in w2], [_f32(b.detach(), "bias") for b in b2]]
    shapes1 = [(256, 134), (256, 390), (256, 256), (256, 256), (256, 256), (65, 256)]
    shapes2 = [(256, 165), (256, 421), (256, 256), (256, 256), (256, 256), (3, 256)]
    for w, shp in zip(keep[0] + keep[2], shapes1 + shapes2):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS mip renderer: weight shape {tuple(w.shape)} != {shp}")
    arrs = [(C.c_void_p * 6)(*[t.data_ptr() for t in lst]) for lst in keep]
    packed = torch.empty(int(lib.na_render_plain_mip_ls_packed_bytes(PREC[precision])), device=dev, dtype=torch.uint8)
    check(lib.na_render_plain_mip_ls_pack(PREC[precision], arrs[0], arrs[1], arrs[2], arrs[3], _ptr(packed), _stream()))
    return packed


def render_plain_mip_ls(rays: torch.Tensor, ts: torch.Tensor, hash_tables: torch.Tensor, packed: torch.Tensor, precision: str,
                        kind: str, t_end: float, min_deg: int, max_deg: int, sigmoid_kind: str = "thin", bg: str = "black",
                        want_weights: bool = False, workspace: Optional[torch.Tensor] = None):
    """PlainNeRF(view) + mip forward as one launch (rays [B,H,W,6] of whole crops; same outputs as render_plain_view_ls)."""
    lib = _lib.load()
    rays, ts, hash_tables = _f32(rays, "rays"), _f32(ts, "ts"), _f32(hash_tables, "hash_tables")
    assert rays.dim() == 4 and rays.shape[-1] == 6, rays.shape
    B, H, W = rays.shape[:3]
    R, T = B * H * W, ts.shape[0]
    if bg not in BG:
        raise NotImplementedError(bg)
    nbytes = int(lib.na_render_ls_workspace_bytes(T, R))
    if workspace is None:
        workspace = torch.empty(nbytes, device=rays.device, dtype=torch.uint8)
    out = torch.empty((B, H, W, 3), device=rays.device, dtype=torch.float32)
    alpha = torch.empty((T, B, H, W), device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty((T, B, H, W), device=rays.device, dtype=torch.float32) if want_weights else None
    check(lib.na_render_plain_mip_ls(_ptr(rays), B, H, W, _ptr(ts), T, _ptr(hash_tables), _ptr(packed), PREC[precision],
                                     MIP_KIND[kind], min_deg, max_deg, float(t_end), SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha),
                                     _ptr(weights), _ptr(out), _ptr(workspace), workspace.numel(), _stream()))
    _f16x_guard(precision, out, "na_render_plain_mip_ls")
    return out, alpha, weights


def _wb_arrays(lists):
    return [(C.c_void_p * len(lst))(*[0 if t is None else t.data_ptr() for t in lst]) for lst in lists]


_FIRST_SHAPES = [(256, 38), (256, 294), (256, 256), (256, 256), (256, 256), (65, 256)]


def render_plain_pos_ls_pack(precision: str, first_wb, pos_wb) -> torch.Tensor:
    """PlainNeRF.first and refl.Positional.mlp ({init, layers.0..4, out}) as one weight stream of the layer-synchronous renderer
    (na_render_plain_pos_ls_pack; f16x only)."""
    lib = _lib.load()
    (w1, b1), (w2, b2) = first_wb, pos_wb
    assert len(w1) == 6 and len(w2) == 7
    keep = [[_f32(w.detach(), "weight") for w in w1], [_f32(b.detach(), "bias") for b in b1],
            [_f32(w.detach(), "weight") for w in w2], [_f32(b.detach(), "bias") for b in b2]]
    shapes2 = [(256, 102), (256, 358), (256, 256), (256, 256), (256, 358), (256, 256), (3, 256)]
    for w, shp in zip(keep[0] + keep[2], _FIRST_SHAPES + shapes2):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS pos renderer: weight shape {tuple(w.shape)} != {shp}")
    nbytes = int(lib.na_render_plain_pos_ls_packed_bytes(PREC[precision]))
    if nbytes == 0:
        raise _lib.NaError(f"LS pos renderer: precision {precision} not supported (f16x)")
    arrs = _wb_arrays(keep)
    packed = torch.empty(nbytes, device=keep[0][0].device, dtype=torch.uint8)
    check(lib.na_render_plain_pos_ls_pack(PREC[precision], arrs[0], arrs[1], arrs[2], arrs[3], _ptr(packed), _stream()))
    return packed


def render_plain_plv_ls_pack(precision: str, first_wb, head_wb, n_rl: int = 0) -> torch.Tensor:
    """PlainNeRF.first and refl.PosLinearView ({pos.init, pos.layers.0..1, pos.out, view.init, view.layers.0..1, view.out}) as one
    weight stream (na_render_plain_plv_ls_pack; f16x only); n_rl = DynamicNeRF's refl_latent columns (0..3)."""
    lib = _lib.load()
    (w1, b1), (w2, b2) = first_wb, head_wb
    assert len(w1) == 6 and len(w2) == 8
    keep = [[_f32(w.detach(), "weight") for w in w1], [_f32(b.detach(), "bias") for b in b1],
            [_f32(w.detach(), "weight") for w in w2], [_f32(b.detach(), "bias") for b in b2]]
    n = int(n_rl)
    shapes2 = [(256, 102 + n), (256, 358 + n), (256, 256), (67, 256), (128, 134 + n), (128, 262 + n), (128, 128), (1, 128)]
    for w, shp in zip(keep[0] + keep[2], _FIRST_SHAPES + shapes2):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS pos-linear-view renderer: weight shape {tuple(w.shape)} != {shp}")
    nbytes = int(lib.na_render_plain_plv_ls_packed_bytes(PREC[precision]))
    if nbytes == 0 or not 0 <= n <= 3:
        raise _lib.NaError(f"LS pos-linear-view renderer: precision {precision} / {n} refl_latent columns not supported (f16x, 0..3)")
    arrs = _wb_arrays(keep)
    packed = torch.empty(nbytes, device=keep[0][0].device, dtype=torch.uint8)
    check(lib.na_render_plain_plv_ls_pack(PREC[precision], arrs[0], arrs[1], arrs[2], arrs[3], n, _ptr(packed), _stream()))
    return packed


def _render_head_ls(which: str, rays, ts, hash_tables, hash_tables_refl, packed, precision, sigmoid_kind, bg, want_weights, workspace,
                    pts, refl_latent=None):
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    hash_tables, hash_tables_refl = _f32(hash_tables, "hash_tables"), _f32(hash_tables_refl, "hash_tables_refl")
    R = rays.numel() // 6
    T = ts.shape[0]
    if bg not in BG:
        raise NotImplementedError(bg)
    nbytes = int(lib.na_render_head_ls_workspace_bytes(T, R))
    if workspace is None:
        workspace = torch.empty(nbytes, device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
    if which == "pos":
        assert refl_latent is None
        check(lib.na_render_plain_pos_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(hash_tables), _ptr(hash_tables_refl), _ptr(packed),
                                         PREC[precision], SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights), _ptr(out),
                                         _ptr(workspace), workspace.numel(), _stream()))
    else:
        n_rl, ld = 0, 0
        if refl_latent is not None:
            n_rl = refl_latent.shape[-1]
            refl_latent, ld = _rows(refl_latent, n_rl, "refl_latent")
            assert refl_latent.shape[0] == T * R, (refl_latent.shape, T, R)
        check(lib.na_render_plain_plv_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(hash_tables), _ptr(hash_tables_refl),
                                         _ptr(refl_latent), ld, n_rl, _ptr(packed), PREC[precision], SIGMOID[sigmoid_kind], BG[bg],
                                         _ptr(alpha), _ptr(weights), _ptr(out), _ptr(workspace), workspace.numel(), _stream()))
    _f16x_guard(precision, out, f"na_render_plain_{which}_ls")
    return out, alpha, weights


def render_plain_pos_ls(rays, ts, hash_tables, hash_tables_refl, packed, precision: str, sigmoid_kind: str = "thin", bg: str = "black",
                        want_weights: bool = False, workspace: Optional[torch.Tensor] = None, pts: Optional[torch.Tensor] = None):
    """PlainNeRF + refl.Positional (`--refl-kind pos`) forward as one launch; same contract as render_plain_view_ls."""
    return _render_head_ls("pos", rays, ts, hash_tables, hash_tables_refl, packed, precision, sigmoid_kind, bg, want_weights, workspace, pts)


def render_plain_plv_ls(rays, ts, hash_tables, hash_tables_refl, packed, precision: str, sigmoid_kind: str = "thin", bg: str = "black",
                        want_weights: bool = False, workspace: Optional[torch.Tensor] = None, pts: Optional[torch.Tensor] = None,
                        refl_latent: Optional[torch.Tensor] = None):
    """PlainNeRF + refl.PosLinearView (`--refl-kind pos-linear-view`) forward as one launch; refl_latent [T, ..., n <= 3] =
    DynamicNeRF's per-sample reflectance latent (rows may be a column slice: passed by pitch)."""
    return _render_head_ls("plv", rays, ts, hash_tables, hash_tables_refl, packed, precision, sigmoid_kind, bg, want_weights, workspace, pts,
                           refl_latent)


def mlp_hash_ls_pack(precision: str, weights, biases) -> torch.Tensor:
    """Pack a hash-encoded SkipConnMLP (in 3, HashEncoder, 5 x 256, skip 3: {init, layers.0..4, out}) into the weight stream of the
    layer-synchronous engine (D-NeRF's deformation network): "f16x" (out <= 32) or, round 6, "bf16x3" (the three-product split, out <= 64)."""
    lib = _lib.load()
    assert len(weights) == 7 and len(biases) == 7
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [None if b is None else _f32(b.detach(), "bias") for b in biases]
    n_out = ws[-1].shape[0]
    shapes = [(256, 38), (256, 294), (256, 256), (256, 256), (256, 294), (256, 256), (n_out, 256)]
    for w, shp in zip(ws, shapes):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS hash MLP: weight shape {tuple(w.shape)} != {shp}")
    nbytes = int(lib.na_mlp_hash_ls_packed_bytes(PREC[precision]))
    if nbytes == 0 or not 1 <= n_out <= (64 if precision == "bf16x3" else 32):
        raise _lib.NaError(f"LS hash MLP: precision {precision} / {n_out} output rows not supported (f16x: <= 32 rows, bf16x3: <= 64)")
    wp = (C.c_void_p * 7)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 7)(*[0 if b is None else b.data_ptr() for b in bs])
    packed = torch.empty(nbytes, device=ws[0].device, dtype=torch.uint8)
    check(lib.na_mlp_hash_ls_pack(PREC[precision], wp, bp, n_out, _ptr(packed), _stream()))
    return packed


def mlp_hash_ls(rays: torch.Tensor, ts: torch.Tensor, tables: torch.Tensor, packed: torch.Tensor, precision: str, n_out: int,
                pts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The packed hash-encoded MLP at the samples of rays [..., 6] x ts [T] (or explicit pts [T, ..., 3]) -> [T, ..., n_out]."""
    lib = _lib.load()
    rays, ts, tables = _f32(rays, "rays"), _f32(ts, "ts"), _f32(tables, "tables")
    T = ts.shape[0]
    R = rays.numel() // 6
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3
    y = torch.empty((T,) + tuple(rays.shape[:-1]) + (n_out,), device=rays.device, dtype=torch.float32)
    check(lib.na_mlp_hash_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(tables), _ptr(packed), PREC[precision], n_out, _ptr(y),
                             n_out, _stream()))
    _f16x_guard(precision, y, "na_mlp_hash_ls")
    return y


def mlp_fourier_ls_pack(precision: str, weights, biases) -> torch.Tensor:
    """Pack a Fourier-encoded SkipConnMLP (in 3, 128 frequencies, 6 x 256, skip 3, out 65: {init, layers.0..5, out}) into the weight
    stream of the layer-synchronous engine (f16x only; VolSDF's MLP SDF network)."""
    lib = _lib.load()
    assert len(weights) == 8 and len(biases) == 8
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [None if b is None else _f32(b.detach(), "bias") for b in biases]
    shapes = [(256, 259), (256, 515), (256, 256), (256, 256), (256, 515), (256, 256), (256, 256), (65, 256)]
    for w, shp in zip(ws, shapes):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS Fourier MLP: weight shape {tuple(w.shape)} != {shp}")
    nbytes = int(lib.na_mlp_fourier_ls_packed_bytes(PREC[precision]))
    if nbytes == 0:
        raise _lib.NaError(f"LS Fourier MLP: precision {precision} not supported (f16x only)")
    wp = (C.c_void_p * 8)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 8)(*[0 if b is None else b.data_ptr() for b in bs])
    packed = torch.empty(nbytes, device=ws[0].device, dtype=torch.uint8)
    check(lib.na_mlp_fourier_ls_pack(PREC[precision], wp, bp, _ptr(packed), _stream()))
    return packed


def mlp_fourier_ls(rays: torch.Tensor, ts: torch.Tensor, basis: torch.Tensor, packed: torch.Tensor, precision: str,
                   pts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The packed Fourier-encoded MLP at the samples of rays [..., 6] x ts [T] (or explicit pts [T, ..., 3]) -> [T, ..., 65];
    basis [3, 128] (FourierEncoder.basis x extra_scale)."""
    lib = _lib.load()
    rays, ts, basis = _f32(rays, "rays"), _f32(ts, "ts"), _f32(basis, "basis")
    assert tuple(basis.shape) == (3, 128), basis.shape
    T = ts.shape[0]
    R = rays.numel() // 6
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3
    y = torch.empty((T,) + tuple(rays.shape[:-1]) + (65,), device=rays.device, dtype=torch.float32)
    check(lib.na_mlp_fourier_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(basis), _ptr(packed), PREC[precision], _ptr(y), 65,
                                _stream()))
    _f16x_guard(precision, y, "na_mlp_fourier_ls")
    return y


def render_tiny_ls_pack(precision: str, weights, biases) -> torch.Tensor:
    """Pack TinyNeRF.estim ({init, layers.0..5, out}) into the weight stream of the layer-synchronous renderer."""
    lib = _lib.load()
    assert len(weights) == 8 and len(biases) == 8, "TinyNeRF.estim has 6 hidden layers"
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [None if b is None else _f32(b.detach(), "bias") for b in biases]
    shapes = [(256, 3), (256, 259), (256, 256), (256, 256), (256, 259), (256, 256), (256, 256), (4, 256)]
    for w, shp in zip(ws, shapes):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS TinyNeRF renderer: weight shape {tuple(w.shape)} != {shp}")
    wp = (C.c_void_p * 8)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 8)(*[0 if b is None else b.data_ptr() for b in bs])
    packed = torch.empty(int(lib.na_render_tiny_ls_packed_bytes(PREC[precision])), device=ws[0].device, dtype=torch.uint8)
    check(lib.na_render_tiny_ls_pack(PREC[precision], wp, bp, _ptr(packed), _stream()))
    return packed


def render_tiny_ls(rays: torch.Tensor, ts: torch.Tensor, packed: torch.Tensor, precision: str, sigmoid_kind: str = "thin",
                   bg: str = "black", want_weights: bool = False, pts: Optional[torch.Tensor] = None):
    """TinyNeRF forward in one kernel (layer-synchronous engine): rays [..., 6], ts [T] -> (rgb [..., 3], alpha, weights)."""
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    R = rays.numel() // 6
    T = ts.shape[0]
    if bg not in BG:
        raise NotImplementedError(bg)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
    check(lib.na_render_tiny_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(packed), PREC[precision], SIGMOID[sigmoid_kind],
                                BG[bg], _ptr(alpha), _ptr(weights), _ptr(out), _stream()))
    _f16x_guard(precision, out, "na_render_tiny_ls")
    return out, alpha, weights


def render_view_ls_pack(precision: str, weights, biases) -> torch.Tensor:
    """Pack refl.View's MLP ({init, layers.0..3, out}; 5 + 64 inputs) into the weight stream of na_render_view_ls."""
    lib = _lib.load()
    assert len(weights) == 6 and len(biases) == 6
    ws = [_f32(w.detach(), "weight") for w in weights]
    bs = [None if b is None else _f32(b.detach(), "bias") for b in biases]
    shapes = [(256, 69), (256, 325), (256, 256), (256, 256), (256, 256), (3, 256)]
    for w, shp in zip(ws, shapes):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS View renderer: weight shape {tuple(w.shape)} != {shp}")
    wp = (C.c_void_p * 6)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * 6)(*[0 if b is None else b.data_ptr() for b in bs])
    packed = torch.empty(int(lib.na_render_view_ls_packed_bytes(PREC[precision])), device=ws[0].device, dtype=torch.uint8)
    check(lib.na_render_view_ls_pack(PREC[precision], wp, bp, _ptr(packed), _stream()))
    return packed


def render_view_ls(rays: torch.Tensor, ts: torch.Tensor, feat: torch.Tensor, beta: torch.Tensor, packed: torch.Tensor,
                   precision: str, sigmoid_kind: str = "thin", bg: str = "black", want_weights: bool = False,
                   pts: Optional[torch.Tensor] = None):
    """View head + compositing in one kernel.  feat [T, *rays.shape[:-1], >= 65]: column 0 signed distance, 1..64 latent
    (the SDF network's output, rows may be wider); beta: the Laplace scale (0-dim or 1-element device tensor)."""
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    R = rays.numel() // 6
    T = ts.shape[0]
    feat, ld = _rows(feat, feat.shape[-1], "feat")
    assert feat.shape[0] == T * R and ld >= 65, (feat.shape, T, R, ld)
    beta = _f32(beta.reshape(1), "beta")
    if bg not in BG:
        raise NotImplementedError(bg)
    workspace = torch.empty(int(lib.na_render_ls_workspace_bytes(T, R)), device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
    check(lib.na_render_view_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(feat), ld, _ptr(beta), _ptr(packed), PREC[precision],
                                SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights), _ptr(out), _ptr(workspace),
                                workspace.numel(), _stream()))
    _f16x_guard(precision, out, "na_render_view_ls")
    return out, alpha, weights


def render_volsdf_siren_ls_pack(precision: str, sdf_wb, view_wb) -> torch.Tensor:
    """Pack the SIREN SDF network (7 Linears) and refl.View's MLP (6) into the stream of na_render_volsdf_siren_ls."""
    lib = _lib.load()
    (w1, b1), (w2, b2) = sdf_wb, view_wb
    assert len(w1) == 7 and len(w2) == 6
    keep = [[_f32(w.detach(), "weight") for w in w1], [None if b is None else _f32(b.detach(), "bias") for b in b1],
            [_f32(w.detach(), "weight") for w in w2], [None if b is None else _f32(b.detach(), "bias") for b in b2]]
    shapes = [(256, 3), (256, 259), (256, 256), (256, 256), (256, 259), (256, 256), (65, 256),
              (256, 69), (256, 325), (256, 256), (256, 256), (256, 256), (3, 256)]
    for w, shp in zip(keep[0] + keep[2], shapes):
        if tuple(w.shape) != shp:
            raise ValueError(f"LS SIREN-VolSDF renderer: weight shape {tuple(w.shape)} != {shp}")
    arrs = [(C.c_void_p * len(lst))(*[0 if t is None else t.data_ptr() for t in lst]) for lst in keep]
    packed = torch.empty(int(lib.na_render_volsdf_siren_ls_packed_bytes(PREC[precision])), device=keep[0][0].device, dtype=torch.uint8)
    check(lib.na_render_volsdf_siren_ls_pack(PREC[precision], arrs[0], arrs[1], arrs[2], arrs[3], _ptr(packed), _stream()))
    return packed


def render_volsdf_siren_ls(rays: torch.Tensor, ts: torch.Tensor, beta: torch.Tensor, packed: torch.Tensor, precision: str,
                           sigmoid_kind: str = "thin", bg: str = "black", want_weights: bool = False,
                           pts: Optional[torch.Tensor] = None):
    """VolSDF (SIREN SDF network + View head) forward in one kernel: rays [..., 6], ts [T] -> (rgb, alpha, weights)."""
    lib = _lib.load()
    rays, ts = _f32(rays, "rays"), _f32(ts, "ts")
    R = rays.numel() // 6
    T = ts.shape[0]
    beta = _f32(beta.reshape(1), "beta")
    if bg not in BG:
        raise NotImplementedError(bg)
    workspace = torch.empty(int(lib.na_render_ls_workspace_bytes(T, R)), device=rays.device, dtype=torch.uint8)
    out = torch.empty(tuple(rays.shape[:-1]) + (3,), device=rays.device, dtype=torch.float32)
    shape_t = (T,) + tuple(rays.shape[:-1])
    alpha = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    weights = torch.empty(shape_t, device=rays.device, dtype=torch.float32) if want_weights else None
    if pts is not None:
        pts = _f32(pts, "pts")
        assert pts.numel() == T * R * 3, (pts.shape, T, R)
    check(lib.na_render_volsdf_siren_ls(_ptr(rays), _ptr(pts), R, _ptr(ts), T, _ptr(beta), _ptr(packed), PREC[precision],
                                        SIGMOID[sigmoid_kind], BG[bg], _ptr(alpha), _ptr(weights), _ptr(out), _ptr(workspace),
                                        workspace.numel(), _stream()))
    _f16x_guard(precision, out, "na_render_volsdf_siren_ls")
    return out, alpha, weights


# ------------------------------------------------------------------------------------------------- forward-mode tangents
def act_deriv(x: torch.Tensor, act: str, order: int = 1) -> torch.Tensor:
    lib = _lib.load()
    x = _f32(x, "x")
    out = torch.empty_like(x)
    check(lib.na_act_deriv(_ptr(x), x.numel(), ACT[act], order, _ptr(out), _stream()))
    return out


def mul_bcast(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [n...] shared by the J leading rows of b [J, n...]."""
    lib = _lib.load()
    a, b = _f32(a, "a"), _f32(b, "b")
    J = b.shape[0]
    assert b.numel() == J * a.numel(), (a.shape, b.shape)
    out = torch.empty_like(b)
    check(lib.na_mul_bcast(_ptr(a), _ptr(b), a.numel(), J, _ptr(out), _stream()))
    return out


def mul_reduce(g: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    g, b = _f32(g, "g"), _f32(b, "b")
    J = b.shape[0]
    out = torch.empty(b.shape[1:], device=b.device, dtype=torch.float32)
    check(lib.na_mul_reduce(_ptr(g), _ptr(b), out.numel(), J, _ptr(out), _stream()))
    return out


def eikonal_loss(normals: torch.Tensor) -> torch.Tensor:
    """normals [3, N] (tangent-major) -> scalar mean((|n| - 1)^2)."""
    lib = _lib.load()
    normals = _f32(normals, "normals")
    loss = torch.zeros(1, device=normals.device, dtype=torch.float32)
    check(lib.na_eikonal_loss(_ptr(normals), normals.shape[1], _ptr(loss), _stream()))
    return loss.reshape(())


def eikonal_loss_backward(normals: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    normals, g = _f32(normals, "normals"), _f32(g.reshape(1), "g")
    out = torch.empty_like(normals)
    check(lib.na_eikonal_loss_backward(_ptr(normals), normals.shape[1], _ptr(g), _ptr(out), _stream()))
    return out
