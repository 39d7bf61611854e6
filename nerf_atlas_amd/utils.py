"""Small helpers of the reference's src/utils.py that the hot path touches."""
import torch

from . import ops

# src/utils.py:500-514 sigmoid_kinds: same string keys; values are callables running the HIP kernel
_SIGMOID_NAMES = ["normal", "thin", "tanh", "cyclic", "upshifted", "fat", "leaky_relu", "relu", "sin",
                  "upshifted_softplus", "upshifted_relu"]


class _Sigmoid:
    def __init__(self, kind):
        self.kind = kind

    def __call__(self, v):
        if torch.is_grad_enabled() and v.requires_grad:
            from .autograd import SigmoidFn
            return SigmoidFn.apply(v.contiguous(), self.kind)
        return ops.sigmoid(v, self.kind)

    def __repr__(self):
        return f"sigmoid[{self.kind}]"


sigmoid_kinds = {k: _Sigmoid(k) for k in _SIGMOID_NAMES}
# "softmax" is a registry key of the reference (nn.Softmax) that no config on the path reaches
sigmoid_kinds["softmax"] = None


def load_sigmoid(kind="thin"):
    """src/utils.py:515-518."""
    s = sigmoid_kinds.get(kind, None)
    if s is None:
        raise NotImplementedError(f"Unknown sigmoid kind({kind})")
    return s


def mse2psnr(x):
    """src/utils.py:184."""
    return -10 * torch.log10(x)


def dir_to_elev_azim(direc):
    """src/utils.py:247-254."""
    return ops.view_elaz(direc)


class MipLatent:
    """The IPE latent of one crop, not materialised: the fused MLP kernels generate its 6*nd columns in their prologue
    (ops.mlp_forward(..., mip=...)), every other consumer gets the [T,B,H,W,6 nd] tensor from .tensor() (na_mip_encode).
    `rest` is an optional tensor of further latent columns that follow it (cat_not_none(mip, intermediate))."""

    def __init__(self, rays, ts, kind, t_end, min_deg, max_deg, rest=None):
        self.rays, self.ts, self.kind, self.t_end, self.min_deg, self.max_deg, self.rest = rays, ts, kind, t_end, min_deg, max_deg, rest
        self._t = None

    @property
    def width(self): return 6 * (self.max_deg - self.min_deg)

    def args(self): return (self.rays, self.ts, self.kind, self.t_end, self.min_deg, self.max_deg)

    def with_rest(self, rest):
        assert self.rest is None
        m = MipLatent(self.rays, self.ts, self.kind, self.t_end, self.min_deg, self.max_deg, rest)
        m._t = self._t
        return m

    def mip_tensor(self):
        if self._t is None:
            self._t = ops.mip_encode(self.rays, self.ts, self.kind, self.t_end, self.min_deg, self.max_deg)
        return self._t

    def tensor(self):
        t = self.mip_tensor()
        return t if self.rest is None else torch.cat([t, self.rest], dim=-1)


class _Mip:
    def __init__(self, kind, min_deg=0, max_deg=16):
        self.kind, self.min_deg, self.max_deg = kind, min_deg, max_deg

    def size(self):
        return self.max_deg - self.min_deg

    def __call__(self, rays, ts):
        """rays [B,H,W,6] of one crop, ts [T] -> [T,B,H,W,96] (intended layout; the last interval is closed
        at ts[-1] + (ts[-1]-ts[-2]) instead of 1e10)."""
        return self.lazy(rays, ts).tensor()

    def lazy(self, rays, ts) -> MipLatent:
        # t_end = NaN: the kernels close the last interval at 2 ts[-1] - ts[-2] themselves (reading it here would be a
        # device synchronisation per forward)
        return MipLatent(rays.contiguous(), ts, self.kind, float("nan"), self.min_deg, self.max_deg)


def CylinderGaussian(min_deg=0, max_deg=16):
    """src/utils.py:103-117."""
    return _Mip("cylinder", min_deg, max_deg)


def ConicGaussian(min_deg=0, max_deg=16):
    """src/utils.py:126-140."""
    return _Mip("cone", min_deg, max_deg)


def load_mip(args):
    """src/utils.py:119-124."""
    if args.mip is None:
        return None
    elif args.mip == "cone":
        return ConicGaussian()
    elif args.mip == "cylinder":
        return CylinderGaussian()
    raise NotImplementedError(f"Unknown mip kind {args.mip}")


# ------------------------------------------------------------------------------------------------- random source
class DeviceRandom:
    """Stochastic training terms (pixel jitter, stratified offsets, density noise) drawn on the tensor's device."""

    def rand(self, shape, device): return torch.rand(shape, device=device)
    def randn(self, shape, device): return torch.randn(shape, device=device)


class ReferenceStreamRandom:
    """The same draws taken from torch's global CPU generator in the reference's call order and copied to the
    device: after `torch.manual_seed(s)` a training run consumes exactly the stream the reference consumes on CPU
    (runner.py:609-850 -> cameras.py:55-58, nerf.py:42-46, nerf.py:347-348), which makes trajectories comparable
    iteration by iteration (tools/ref_train_fixture.py)."""

    def rand(self, shape, device): return torch.rand(shape).to(device)
    def randn(self, shape, device): return torch.randn(shape).to(device)


random_source = DeviceRandom()


def set_random_source(src):
    global random_source
    random_source = src if src is not None else DeviceRandom()
    return random_source


def rand(shape, device): return random_source.rand(tuple(shape), device)
def randn(shape, device): return random_source.randn(tuple(shape), device)


PACK_CACHE_ATTRS = ("_packed", "_packed_ls", "_packed_mip_ls", "_packed_view_ls", "_packed_siren_ls", "_packed_fourier_ls")


def invalidate_packed(module) -> int:
    """Drop every cached packed weight stream under `module` (SkipConnMLP._packed, the layer-synchronous streams of the
    models, the stacked hash tables of HashEncoder).  The caches are keyed on (Parameter._version, data_ptr): in-place ops on
    the Parameter under no_grad (`p.copy_`, `p.mul_`, optimizer steps) bump the version and re-pack on their own, and
    `load_state_dict` / `.to()` / `.float()` ... invalidate through the hooks of `PackedCacheMixin`; writes THROUGH `.data`
    (`p.data.copy_(...)`, `p.data *= ...`) or an out-of-band hipMemcpy into the parameter do neither, and the fused kernels
    would keep rendering the old weights.  Call this (or `model.invalidate_packed()`) after such writes, or run with
    `config.set_repack_always(True)`.  Returns the number of caches cleared."""
    n = 0
    for m in module.modules():
        for a in PACK_CACHE_ATTRS:
            c = m.__dict__.get(a)
            if isinstance(c, dict) and c:
                c.clear()
                n += 1
        if m.__dict__.get("_stacked") is not None and "_stamp" in m.__dict__:  # HashEncoder.tables()
            m._stacked = None
            m._stamp = None
            n += 1
    return n


class PackedCacheMixin:
    """Modules that cache packed weight streams (CommonNeRF, SkipConnMLP, HashEncoder): `invalidate_packed()` as a method,
    called automatically whenever torch replaces or rewrites the parameters wholesale -- `_apply` (`.to`, `.cuda`, `.float`,
    `.half` ...) and `load_state_dict` (a post hook, which torch runs on every submodule of the module being loaded)."""

    def _init_packed_hooks(self):
        def _hook(mod, _incompatible_keys):
            mod.invalidate_packed()  # (a post hook must return None)
        self.register_load_state_dict_post_hook(_hook)

    def invalidate_packed(self) -> int:
        return invalidate_packed(self)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packed()
        return out


def pack_stamp(linears):
    """cache key of a packed stream: version counter + address of every weight / bias; None = never reuse
    (config.repack_always: for callers that write parameters behind torch's back)"""
    from . import config
    if config.repack_always:
        return None
    return tuple((l.weight._version, l.weight.data_ptr(), l.bias._version, l.bias.data_ptr()) for l in linears)


# ------------------------------------------------------------------------------------------------- fallback notices
_noted = set()


def note_fallback(tag: str, msg: str):
    """ONE line on stderr per process and tag when an inference call leaves the fused kernels for a slower path because of its
    shape (the schedules of the layer-synchronous engine serve exactly the reference's default shapes; any
    other --hidden / layer count used to drop to the generic kernels, or to the exact-fp32 any-shape Linears, silently).
    NA_QUIET_FALLBACK=1 silences it."""
    import os
    import sys
    if tag in _noted or os.environ.get("NA_QUIET_FALLBACK") == "1":
        return
    _noted.add(tag)
    print(f"[nerf_atlas_amd] note: {msg}", file=sys.stderr, flush=True)
