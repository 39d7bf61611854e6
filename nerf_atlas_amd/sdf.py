"""SDF models used by VolSDF's volume path (src/sdf.py:15-32,83-112,250-258,278-287,308-316)."""
import torch
import torch.nn as nn

from . import refl as _refl
from .neural_blocks import FourierEncoder, SkipConnMLP


class SDFModel(nn.Module):
    def __init__(self, intermediate_size: int = 32):
        super().__init__()
        self.intermediate_size = intermediate_size

    def _net(self):
        raise NotImplementedError()

    def normals(self, pts, values=None):
        """src/sdf.py:43-48.  With `values=None` -- every call site of the reference -- the reference differentiates
        `self(pts)`, the WHOLE output row (signed distance + the intermediate features), with `grad_outputs = ones`
        (`utils.autograd`, src/utils.py:266-277): what it calls normals is d(sum_j out_j)/d pts, not d sdf/d pts.  The
        regularisers built on it (`--sdf-eikonal`, `--smooth-normals`: runner.py:683-727) are reproduced as the reference
        computes them (pinned by tests/golden/train_parity_volsdf_smooth.json, the reference's own training run).
        Forward-mode tangents through the MLP (one value row and three tangent rows per point;
        SkipConnMLP.forward_with_input_tangents) instead of autograd(create_graph=True): the result is differentiable
        w.r.t. the weights with first-order autograd.  Always runs the fp32 / split-bf16 training GEMMs, also under no_grad.
        `values="sdf"` selects the gradient of the signed distance alone (column 0)."""
        flat = pts.reshape(-1, 3)
        return self.normals_tangent_major(flat, values).t().reshape(pts.shape)

    def normals_tangent_major(self, pts, values=None):
        """[3, N] layout of the same normals (what ops.eikonal_loss consumes: no transposition in the graph)."""
        if values is not None and not (isinstance(values, str) and values == "sdf"):
            # the reference's `values` is a TENSOR to differentiate (src/sdf.py:43-48); forward-mode tangents cannot take an
            # arbitrary graph output, and silently treating it as None would return a different quantity
            raise NotImplementedError("SDFModel.normals: values must be None (the reference's call sites: gradient of the whole "
                                      "output row) or \"sdf\" (gradient of the signed distance); a tensor to differentiate is "
                                      "not supported by the forward-mode implementation")
        _, t = self._net().forward_with_input_tangents(pts.reshape(-1, 3))
        return t[..., 0] if values == "sdf" else t.sum(-1)


class MLP(SDFModel):
    """src/sdf.py:250-258."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.mlp = SkipConnMLP(in_size=3, out=1 + self.intermediate_size,
                               enc=FourierEncoder(input_dims=3, sigma=1 << 4), num_layers=6, hidden_size=256,
                               init="xavier")

    def forward(self, x): return self.mlp(x)
    def _net(self): return self.mlp


class SIREN(SDFModel):
    """src/sdf.py:278-287."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.siren = SkipConnMLP(in_size=3, out=1 + self.intermediate_size, num_layers=5, hidden_size=256,
                                 activation=torch.sin, skip=3, init="siren")

    def forward(self, x): return self.siren(x)
    def _net(self): return self.siren


class SDF(nn.Module):
    """src/sdf.py:83-112 (the surface-intersection methods are out of scope: SURVEY 2 row 9/10)."""

    def __init__(self, underlying: SDFModel, reflectance, isect=None, t_near: float = 0, t_far: float = 1, alpha: int = 1000):
        super().__init__()
        assert isinstance(underlying, SDFModel)
        self.underlying = underlying
        self.refl = reflectance
        self.far, self.near, self.alpha, self.isect = t_far, t_near, alpha, isect

    @property
    def sdf(self): return self

    @property
    def intermediate_size(self): return self.underlying.intermediate_size

    def from_pts(self, pts):
        raw = self.underlying(pts)
        latent = raw[..., 1:]
        return raw[..., 0], latent if latent.shape[-1] != 0 else None

    def normals(self, pts, values=None): return self.underlying.normals(pts, values)

    def intersect_mask(self, r_o, r_d, near=None, far=None, eps=1e-3):
        """src/sdf.py:123-135: rays whose closest uniform sample along [near, far] stays outside the surface
        (throughput >= eps) are masked out; (~hits, throughput, None)."""
        from . import march
        with torch.no_grad():
            throughput, _, _, _ = march.throughput_with_sign_change(
                self.underlying, r_o, r_d, near=self.near if near is None else near, far=self.far if far is None else far,
                batch_size=32 if self.training else 196)
            hits = throughput < eps
            return ~hits, throughput, None


def _out_of_scope(name):
    def cons(*a, **k):
        raise NotImplementedError(f"sdf kind '{name}' belongs to the surface-rendering path (out of scope)")
    return cons


# src/sdf.py:308-316
sdf_kinds = {"mlp": MLP, "siren": SIREN, **{k: _out_of_scope(k) for k in ["spheres", "triangles", "local", "curl-mlp"]}}


def load(args, with_integrator: bool = False):
    """src/sdf.py:15-32."""
    cons = sdf_kinds.get(args.sdf_kind, None)
    if cons is None:
        raise NotImplementedError(f"Unknown SDF kind: {args.sdf_kind}")
    model = cons(intermediate_size=args.shape_to_refl_size)
    refl_inst = _refl.load(args, args.refl_kind, getattr(args, "space_kind", "identity"), model.intermediate_size)
    return SDF(model, refl_inst, isect=None, t_near=args.near, t_far=args.far)
