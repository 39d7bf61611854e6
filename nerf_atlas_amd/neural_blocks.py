"""Encoders and SkipConnMLP with the reference's constructor/attribute protocol (src/neural_blocks.py:14-311).

Parameters live in ordinary nn.Parameters under the reference's names (init / layers.N / out / enc.embs.N /
enc.basis), so a reference state_dict loads unchanged.  forward() runs HIP kernels only: the fused MFMA engine
when the shape has one (hidden 256), otherwise a chain of exact-fp32 MFMA Linears.  With gradients enabled the
same operators run through nerf_atlas_amd/autograd.py (HIP forward + backward kernels).
"""
import math
from typing import Optional

import os

import torch
import torch.nn as nn

from . import autograd as ag
from . import config, ops, utils



class PositionalEncoder(nn.Module):
    """src/neural_blocks.py:14-34."""

    def __init__(self, input_dims: int = 3, max_freq: float = 6., N: int = 64, log_sampling: bool = False):
        super().__init__()
        if log_sampling:
            bands = 2 ** torch.linspace(1, max_freq, steps=N, dtype=torch.float)
        else:
            bands = torch.linspace(1, 2 ** max_freq, steps=N, dtype=torch.float)
        self.bands = nn.Parameter(bands, requires_grad=False)
        self.input_dims = input_dims

    def output_dims(self):
        return self.input_dims * 2 * len(self.bands)

    def forward(self, x):
        assert x.shape[-1] == self.input_dims
        if ag.needs_grad(x):
            raise NotImplementedError("d(%s)/d(position) has no HIP backward yet (DESIGN.md 9a)" % type(self).__name__)
        return ops.positional_encode(x, self.bands.data)


class FourierEncoder(nn.Module):
    """src/neural_blocks.py:36-55; basis = sigma * randn(freqs, D).T drawn on CPU like the reference."""

    def __init__(self, input_dims: int = 3, freqs: int = 128, sigma: int = 1 << 5, device="cpu"):
        super().__init__()
        self.input_dims = input_dims
        self.freqs = freqs
        self.basis = nn.Parameter(sigma * torch.randn(freqs, input_dims).T.contiguous(), requires_grad=False)
        self.extra_scale = 1

    def output_dims(self):
        return self.freqs * 2

    def forward(self, x):
        if ag.needs_grad(x):
            raise NotImplementedError("d(%s)/d(position) has no HIP backward yet (DESIGN.md 9a)" % type(self).__name__)
        return ops.fourier_encode(x, self.basis.data, float(self.extra_scale))

    def scale_freqs(self, amt: 1 + 1e-5, cap=2):
        self.extra_scale *= amt
        self.extra_scale = min(self.extra_scale, cap)


class LearnedFourierEncoder(nn.Module):
    """src/neural_blocks.py:57-72: Fourier features with a learnable scalar frequency scale."""

    def __init__(self, input_dims: int = 3, num_freqs: int = 16, sigma: int = 1 << 5, device="cpu"):
        super().__init__()
        self.input_dims = input_dims
        self.n_freqs = num_freqs
        self.basis = nn.Parameter(sigma * torch.randn(num_freqs, input_dims).T.contiguous(), requires_grad=False)
        self.extra_scale = nn.Parameter(torch.tensor(1.0), requires_grad=True)

    def output_dims(self):
        return self.n_freqs * 2

    def forward(self, x):
        if ag.needs_grad(x, self.extra_scale):
            raise NotImplementedError("LearnedFourierEncoder has no HIP backward yet (DESIGN.md 9a)")
        return ops.fourier_encode(x, self.basis.data, float(self.extra_scale))


class NNEncoder(nn.Module):
    """src/neural_blocks.py:75-87: sin(30 * Linear(x)) (the factor is folded into the exact-fp32 Linear)."""

    def __init__(self, input_dims: int = 3, out: int = 32, device=None):
        super().__init__()
        self.fwd = nn.Linear(input_dims, out)

    def output_dims(self):
        return self.fwd.out_features

    def forward(self, x):
        assert x.shape[-1] == self.fwd.in_features
        flat = x.reshape(-1, x.shape[-1]).contiguous()
        if ag.needs_grad(flat, *self.fwd.parameters()):
            # differentiable path: sin(30 (W x + b)) through the Linear and sigmoid autograd functions
            y = ag.LinearFn.apply(flat, None, self.fwd.weight * 30.0, self.fwd.bias * 30.0, "none")
            return ag.SigmoidFn.apply(y, "sin").reshape(x.shape[:-1] + (self.fwd.out_features,))
        y = ops.linear_f32(flat, (30.0 * self.fwd.weight.data).contiguous(), (30.0 * self.fwd.bias.data).contiguous())
        return ops.sigmoid(y, "sin").reshape(x.shape[:-1] + (self.fwd.out_features,))


class HashEncoder(utils.PackedCacheMixin, nn.Module):
    """src/neural_blocks.py:92-193 (8 levels x 65536 x 4, resolutions 16 * 0.87497^l, Q8/Q9)."""

    def __init__(self, input_dims: int = 3, emb_size: int = 1 << 16, feat_size: int = 4, levels: int = 8,
                 include_input=True):
        super().__init__()
        assert input_dims == 3, "Only supports 3 inputs currently"
        assert emb_size == 1 << 16 and feat_size == 4 and levels == 8, "the HIP hash encoder is specialised for the defaults"
        self.register_buffer("primes", torch.tensor([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437,
                                                     2165219737]), persistent=True)
        self.levels, self.in_features, self.include_input = levels, input_dims, include_input
        self.emb_size, self.feat_size = emb_size, feat_size
        self.embs = nn.ModuleList([nn.Embedding(emb_size, feat_size) for _ in range(levels)])
        self.low_reso, self.high_reso = 1 << 4, 1 << 14
        self.scale = math.exp((math.log(self.high_reso) - math.log(self.low_reso)) / levels - 1)
        self._stacked = None
        self._stamp = None
        self._init_packed_hooks()

    def output_dims(self):
        return self.levels * self.feat_size + self.include_input * self.in_features

    def tables(self) -> torch.Tensor:
        """[8,65536,4] contiguous copy of the embedding tables (rebuilt when a table changes)."""
        stamp = None if config.repack_always else tuple((e.weight._version, e.weight.data_ptr()) for e in self.embs)
        if self._stacked is None or stamp is None or stamp != self._stamp:
            self._stacked = torch.stack([e.weight.data for e in self.embs]).contiguous()
            self._stamp = stamp
        return self._stacked

    def forward(self, x):
        assert x.shape[-1] == self.in_features
        if ag.needs_grad(x, *[e.weight for e in self.embs]):
            flat = x.reshape(-1, 3).contiguous()
            tables = torch.stack([e.weight for e in self.embs])  # differentiable view of the 8 parameters
            return ag.HashEncodeFn.apply(flat, tables, self.include_input).reshape(x.shape[:-1] + (self.output_dims(),))
        return ops.hash_encode(x, self.tables(), self.include_input)


mlp_init_kinds = {None, "zero", "kaiming", "siren", "xavier"}


class _Act:
    def __init__(self, name):
        self.name = name


class SkipConnMLP(utils.PackedCacheMixin, nn.Module):
    """src/neural_blocks.py:204-311.  `activation`: nn.LeakyReLU (default) or torch.sin."""

    def __init__(self, num_layers=5, hidden_size=256, in_size=3, out=3, skip=3, activation=None, latent_size=0,
                 enc=None, last_layer_act=False, linear=nn.Linear, init=None):
        assert init in mlp_init_kinds, "Must use init kind"
        super().__init__()
        self.in_size = in_size
        self.enc = enc
        map_size = enc.output_dims() if enc is not None else 0
        self.dim_p = in_size + map_size + latent_size
        self.skip = skip
        self.latent_size = latent_size
        skip_size = hidden_size + self.dim_p
        self.init = nn.Linear(self.dim_p, hidden_size)
        self.layers = nn.ModuleList([
            linear(skip_size if (i % skip) == 0 and i != num_layers - 1 else hidden_size, hidden_size)
            for i in range(num_layers)])
        self.out = nn.Linear(hidden_size, out)
        weights = [self.init.weight, self.out.weight, *[l.weight for l in self.layers]]
        biases = [self.init.bias, self.out.bias, *[l.bias for l in self.layers]]
        if init == "zero":
            for t in weights + biases: nn.init.zeros_(t)
        elif init == "xavier":
            for t in weights: nn.init.xavier_uniform_(t)
            for t in biases: nn.init.zeros_(t)
        elif init == "siren":
            for t in weights:
                fan_in, _ = nn.init._calculate_fan_in_and_fan_out(t)
                a = math.sqrt(6 / fan_in)
                nn.init._no_grad_uniform_(t, -a, a)
            for t in biases: nn.init.zeros_(t)
        elif init == "kaiming":
            for t in weights: nn.init.kaiming_normal_(t, mode="fan_out")
            for t in biases: nn.init.zeros_(t)
        if activation is None or isinstance(activation, nn.LeakyReLU):
            self.act_name = "leaky_relu"
        elif activation is torch.sin:
            self.act_name = "sin"
        else:
            raise NotImplementedError(f"activation {activation}: the HIP path implements LeakyReLU and sin")
        self.activation = activation
        self.last_layer_act = last_layer_act
        self._packed = {}
        self._init_packed_hooks()

    # ---- HIP plumbing ------------------------------------------------------------------------
    def _linears(self):
        return [self.init, *self.layers, self.out]

    def enc_kind(self):
        if self.enc is None:
            return "none", 0
        if isinstance(self.enc, HashEncoder) and self.enc.include_input:
            return "hash", 35
        if isinstance(self.enc, FourierEncoder):
            return "fourier", self.enc.output_dims()
        return None, self.enc.output_dims()  # encoder without a fused prologue

    def desc(self, layout="generic"):
        kind, dims = self.enc_kind()
        if kind is None:
            return None
        return ops.make_desc(self.in_size, kind, dims, self.latent_size, len(self.layers), self.init.out_features,
                             self.out.out_features, self.skip, self.act_name, layout)

    def packed(self, precision: str, layout: str = "generic"):
        """Packed MFMA weight stream (cached; re-packed when any parameter changed), or None if this shape has
        no fused kernel."""
        desc = self.desc(layout)
        if desc is None or ops.mlp_packed_bytes(desc, precision) == 0:
            return None, None
        lin = self._linears()
        stamp = utils.pack_stamp(lin)
        key = (precision, layout)
        hit = self._packed.get(key)
        if hit is None or stamp is None or hit[0] != stamp:
            buf = ops.mlp_pack(desc, precision, [l.weight.data for l in lin], [l.bias.data for l in lin])
            self._packed[key] = (stamp, buf)
        return desc, self._packed[key][1]

    def _mip_prologue_shape(self):
        """the shapes csrc/mlp_fwd_inst.hip instantiates with the IPE prologue: PlainNeRF.first (hash, 38 + 96 inputs)
        and View.mlp (sin, 5 + 96 + 64 inputs)"""
        if self.init.out_features != 256 or self.packed(config.kernel_precision())[1] is None:
            return False
        if isinstance(self.enc, HashEncoder):
            return self.act_name == "leaky_relu" and self.dim_p == 38 + 96
        return self.enc is None and self.act_name == "sin" and self.dim_p == 5 + 96 + 64

    def enc_params(self):
        if isinstance(self.enc, HashEncoder):
            return self.enc.tables()
        if isinstance(self.enc, FourierEncoder):
            b = self.enc.basis.data
            return b if self.enc.extra_scale == 1 else (b * self.enc.extra_scale)
        return None

    def forward(self, p, latent: Optional[torch.Tensor] = None, pre=None):
        """pre (training only): (output rows of init and of every hidden layer but the last Linear, the network's output) of a forward that
        has already run (PlainNeRF's one-launch training forward); the node then only records what its backward needs."""
        batches = p.shape[:-1]
        mip = None
        if isinstance(latent, utils.MipLatent):
            # lazy IPE latent (config 3): the two MLP shapes of PlainNeRF(view) generate it in their prologue; every other
            # consumer (training, other shapes) sees the materialised tensor
            rest = latent.rest
            if (not self.last_layer_act and latent.width == 96 and self._mip_prologue_shape()
                    and not ag.needs_grad(p, rest, *self.parameters())):
                mip, latent = latent, rest
            else:
                latent = latent.tensor()
        if self.latent_size != 0:
            assert latent is not None or mip is not None, "Did not pass latent vector when some was expected"
        else:
            assert (latent is None) or (latent.shape[-1] == 0), "Passed latent vector when none was expected"
            latent = None
        out_size = self.out.out_features
        if mip is not None:
            desc, packed = self.packed(config.kernel_precision())
            assert packed is not None and self.latent_size == mip.width + (0 if latent is None else latent.shape[-1])
            y = ops.mlp_forward(desc, config.kernel_precision(), packed, p.reshape(-1, p.shape[-1]),
                                None if latent is None else latent.reshape(-1, latent.shape[-1]), self.enc_params(),
                                mip=mip.args())
            return y.reshape(batches + (out_size,))
        if not ag.needs_grad(p, latent, *self.parameters()):
            desc, packed = (None, None) if self.last_layer_act else self.packed(config.kernel_precision())
            if packed is not None:
                # column slices of wider buffers (`first_out[..., 1:]`) go down with their row pitch: no copy
                y = ops.mlp_forward(desc, config.kernel_precision(), packed, p.reshape(-1, p.shape[-1]),
                                    None if latent is None else latent.reshape(-1, self.latent_size), self.enc_params())
                return y.reshape(batches + (out_size,))
        flat = p.reshape(-1, p.shape[-1]).contiguous()
        lat = None if latent is None else latent.reshape(-1, self.latent_size).contiguous()
        if ag.needs_grad(flat, lat, *self.parameters()):
            return self._forward_train(flat, lat, pre=pre).reshape(batches + (out_size,))
        # any-shape path: exact-fp32 Linears (src/neural_blocks.py:288-296)
        if not self.last_layer_act and flat.is_cuda:
            utils.note_fallback(f"mlp-fp32-{self.in_size}-{self.init.out_features}-{len(self.layers)}-{self.out.out_features}",
                                f"SkipConnMLP(in {self.in_size}, hidden {self.init.out_features} x {len(self.layers)}, out {self.out.out_features}, "
                                f"enc {type(self.enc).__name__}) has no packed form for the fused MLP kernels: inference runs one exact-fp32 "
                                "Linear launch per layer")
        init = flat
        if self.enc is not None:
            init = torch.cat([init, self.enc(flat)], dim=-1)
        if lat is not None:
            init = torch.cat([init, lat], dim=-1)
        x = ops.linear_f32(init, self.init.weight.data, self.init.bias.data)
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            skip = i != n - 1 and (i % self.skip) == 0
            x = ops.linear_f32(x, layer.weight.data, layer.bias.data, pre_act=self.act_name, x1=init if skip else None)
        if self.last_layer_act:
            setattr(self, "last_layer_out", x.reshape(batches + (-1,)))
        y = ops.linear_f32(x, self.out.weight.data, self.out.bias.data, pre_act=self.act_name)
        return y.reshape(batches + (out_size,))

    def forward_rows(self, init, pre=None):
        """Differentiable forward from ready-made init rows [N, dim_p] = [p | enc(p) | latent] (a caller that assembles them with one
        kernel: PlainNeRF's training path, autograd.PlainHeadFn): src/neural_blocks.py:288-296 without the cats of :283-287."""
        assert init.dim() == 2 and init.shape[1] == self.init.in_features and self.enc is None, (init.shape, self.init.in_features)
        return self._forward_train(None, None, init=init, pre=pre)

    def _forward_train(self, flat, lat, init=None, pre=None):
        """Differentiable forward (fp32 Linears, HIP forward and backward kernels): src/neural_blocks.py:279-296."""
        if init is not None:
            pass
        elif (isinstance(self.enc, HashEncoder) and lat is None and flat.is_cuda and flat.shape[1] == 3
              and os.environ.get("NA_TRAIN_ROWS") != "0"):
            # [p | p | features] written by the encoder itself, its gradient read in place (autograd.HashInitFn: no cat, no slice copy)
            # (pre[2]: the stacked tables of this step, when the caller has built them already)
            tables = pre[2] if pre is not None and len(pre) > 2 else torch.stack([e.weight for e in self.enc.embs])
            init = ag.HashInitFn.apply(flat, tables, self.enc.include_input)
        else:
            init = flat
            if self.enc is not None:
                init = torch.cat([init, self.enc(flat)], dim=-1)
            if lat is not None:
                init = torch.cat([init, lat], dim=-1)
        init = init.contiguous()
        packs = self._train_packs(init)
        if self._mlp_fn_ok(init, packs):
            # every Linear takes the fused training kernels: the network is ONE autograd node (autograd.MlpTrainFn)
            n = len(self.layers)
            spec = {"act": self.act_name, "skips": [i != n - 1 and (i % self.skip) == 0 for i in range(n)], "packs": packs, "pre": None if pre is None else pre[:2]}
            params = []
            for lin in self._linears():
                params += [lin.weight, lin.bias]
            return ag.MlpTrainFn.apply(init, spec, *params)
        x = ag.LinearFn.apply(init, None, self.init.weight, self.init.bias, "none", packs[0])
        n = len(self.layers)
        for i, layer in enumerate(self.layers):
            skip = i != n - 1 and (i % self.skip) == 0
            x = ag.LinearFn.apply(x, init if skip else None, layer.weight, layer.bias, self.act_name, packs[1 + i])
        if self.last_layer_act:
            setattr(self, "last_layer_out", x)
        return ag.LinearFn.apply(x, None, self.out.weight, self.out.bias, self.act_name, packs[-1])

    def _mlp_fn_ok(self, init, packs) -> bool:
        """Can the whole network run as autograd.MlpTrainFn?  The split-bf16 arithmetic with packed operands for every Linear, the
        input needs its gradient, every source of every Linear runs the one-pass backward kernel (256 wide or <= 128 columns, at
        most 256 outputs, N >= 8 192), nobody asks for the last hidden layer's output."""
        if (config.train_precision != "bf16x3" or self.last_layer_act or not init.requires_grad or os.environ.get("NA_TRAIN_MLP_FN") == "0"
                or any(p is None or p[1] is None for p in packs)):
            return False
        N, w0 = init.shape
        n = len(self.layers)
        for i, lin in enumerate(self._linears()):
            out, k = lin.weight.shape
            skip = 1 <= i <= n and (i - 1) != n - 1 and ((i - 1) % self.skip) == 0
            in0 = k - (w0 if skip else 0)
            if not lin.weight.requires_grad or not ops.linear_bwd_fused_ok(N, out, in0) or (skip and not ops.linear_bwd_fused_ok(N, out, w0)):
                return False
        return True

    def _train_packs(self, init):
        """[(packed W, packed W^T | None) | None per Linear] for the split-bf16 training GEMMs: the bf16 hi / lo MFMA fragments of
        EVERY Linear of this network -- the forward's W and the input gradient's W^T, read straight from the Parameters -- built
        by ONE launch per training step (round 5; 24 pack launches + 12 transposing copies per PlainNeRF step before).  None where
        the batch or a shape takes the K-staged kernels (ops.train_gemm_packed_ok), or in the exact-fp32 arithmetic."""
        lins = self._linears()
        if config.train_precision != "bf16x3" or not init.is_cuda:
            return [None] * len(lins)
        N = init.shape[0]
        need_in = [init.requires_grad] + [True] * (len(lins) - 1)   # (a hidden layer's input always needs its gradient)
        want, mats = [], []
        for lin, nd in zip(lins, need_in):
            W = lin.weight
            ok = (W.is_contiguous() and W.dtype == torch.float32 and ops.train_gemm_packed_ok(N, W.shape[0])
                  and ops.train_gemm_packed_ok(N, W.shape[1]))
            want.append((ok, ok and nd))
            if ok:
                mats.append((W, False))
                if nd:
                    mats.append((W, True))
        views = iter(ops.train_pack_many(mats))
        out = []
        for ok, bw in want:
            if not ok:
                out.append(None)
                continue
            f = next(views)
            t = next(views) if bw else None
            out.append((f, t) if f is not None and (t is not None or not bw) else None)
        return out

    def forward_with_input_tangents(self, p):
        """(y [N,out], t [3,N,out]) with t[j] = d y / d p_j, propagated FORWARD through the network next to the values
        (src/sdf.py:43: the reference obtains the same Jacobian rows with torch.autograd.grad(create_graph=True)):
            t_0 = W_init . d[p | enc(p)]/dp_j,   t_{l+1} = W_l . (act'([z_l | init]) * [t_l | d init]),  no biases.
        Every node is a first-order autograd Function over HIP kernels (LinearFn, MulBcastFn, ActDerivFn), so a loss on
        the tangents -- the eikonal term, runner.py:685-692 -- back-propagates to the weights without double backward.
        p [N,3] does not require grad; encoders: none or FourierEncoder (its Jacobian is elementwise: cos/-sin times the
        fixed basis)."""
        assert self.latent_size == 0 and p.dim() == 2 and p.shape[1] == self.in_size == 3
        p = p.contiguous()
        N = p.shape[0]
        eye = torch.eye(3, device=p.device, dtype=torch.float32)[:, None, :].expand(3, N, 3)
        init, dinit = p, eye
        if self.enc is not None:
            if not isinstance(self.enc, FourierEncoder):
                raise NotImplementedError("input tangents through " + type(self.enc).__name__)
            enc = ops.fourier_encode(p, self.enc.basis.data, float(self.enc.extra_scale))  # [N, 2F] = [sin | cos]
            F_ = self.enc.freqs
            swapped = torch.cat([enc[:, F_:], enc[:, :F_]], dim=-1).contiguous()            # [cos | sin]
            B = (self.enc.basis.data * float(self.enc.extra_scale))                          # [3, F]
            signed = torch.cat([B, -B], dim=-1)[:, None, :].expand(3, N, 2 * F_).contiguous()  # d/dp_j: cos*B_j | -sin*B_j
            denc = ops.mul_bcast(swapped, signed)
            init = torch.cat([p, enc], dim=-1)
            dinit = torch.cat([eye, denc], dim=-1)
        init = init.contiguous()
        dinit = dinit.contiguous()
        K0 = init.shape[1]
        z = ag.LinearFn.apply(init, None, self.init.weight, self.init.bias, "none")
        t = ag.LinearFn.apply(dinit.reshape(3 * N, K0), None, self.init.weight, None, "none").reshape(3, N, -1)
        n = len(self.layers)
        lins = [(l, i != n - 1 and (i % self.skip) == 0) for i, l in enumerate(self.layers)] + [(self.out, False)]
        for lin, skip in lins:
            x_in = torch.cat([z, init], dim=-1).contiguous() if skip else z
            t_in = torch.cat([t, dinit], dim=-1).contiguous() if skip else t
            m = ag.ActDerivFn.apply(x_in, self.act_name)
            mt = ag.MulBcastFn.apply(m, t_in)
            z = ag.LinearFn.apply(z, init if skip else None, lin.weight, lin.bias, self.act_name)
            t = ag.LinearFn.apply(mt.reshape(3 * N, -1), None, lin.weight, None, "none").reshape(3, N, -1)
        return z, t

    @torch.no_grad()
    def forward_with_direction_tangent(self, p, e):
        """(y [N,out], dy [N,out]) with dy = (d y / d p) . e for one direction e per point -- the Jacobian-vector product
        the FFJORD divergence estimate contracts with e (runner.py:697-700, src/utils.py:467-478).  Values and tangent go
        through the network side by side in exact fp32 (the tangent sees no biases and act'(.) instead of act(.)).  No
        graph: the reference's estimate is a constant for the optimiser (torch.autograd.grad without create_graph).
        Encoders: none, HashEncoder (hash_jvp kernel), FourierEncoder."""
        assert self.latent_size == 0 and p.shape[-1] == self.in_size == 3 and e.shape == p.shape
        p, e = p.reshape(-1, 3).contiguous(), e.reshape(-1, 3).contiguous()
        init, dinit = p, e
        if isinstance(self.enc, HashEncoder):
            tables = self.enc.tables()
            init = torch.cat([p, ops.hash_encode(p, tables, self.enc.include_input)], dim=-1)
            dinit = torch.cat([e, ops.hash_encode_jvp(p, tables, e, self.enc.include_input)], dim=-1)
        elif isinstance(self.enc, FourierEncoder):
            enc = ops.fourier_encode(p, self.enc.basis.data, float(self.enc.extra_scale))
            F_ = self.enc.freqs
            proj = ops.linear_f32(e, (self.enc.basis.data * float(self.enc.extra_scale)).T.contiguous(), None)  # e . B
            swapped = torch.cat([enc[:, F_:], -enc[:, :F_]], dim=-1).contiguous()                              # cos | -sin
            denc = ops.mul_bcast(swapped, torch.cat([proj, proj], dim=-1)[None].contiguous())[0]
            init, dinit = torch.cat([p, enc], dim=-1), torch.cat([e, denc], dim=-1)
        elif self.enc is not None:
            raise NotImplementedError("direction tangents through " + type(self.enc).__name__)
        init, dinit = init.contiguous(), dinit.contiguous()
        z = ops.linear_f32(init, self.init.weight.data, self.init.bias.data)
        t = ops.linear_f32(dinit, self.init.weight.data, None)
        n = len(self.layers)
        lins = [(l, i != n - 1 and (i % self.skip) == 0) for i, l in enumerate(self.layers)] + [(self.out, False)]
        for lin, skip in lins:
            x_in = torch.cat([z, init], dim=-1).contiguous() if skip else z
            t_in = torch.cat([t, dinit], dim=-1).contiguous() if skip else t
            mt = ops.mul_bcast(ops.act_deriv(x_in, self.act_name), t_in[None])[0]
            z = ops.linear_f32(z, lin.weight.data, lin.bias.data, pre_act=self.act_name, x1=init if skip else None)
            t = ops.linear_f32(mt, lin.weight.data, None)
        return z, t

    def forward_with_direction_tangent_graph(self, p, e):
        """The same (y, dy = (d y / d p) . e) as forward_with_direction_tangent, but every node a first-order autograd Function
        (LinearFn in the configured training arithmetic, ActDerivFn, MulBcastFn, HashEncodeFn, HashJvpFn): a loss on dy
        back-propagates to the weights and the hash tables without double backward -- what `--dyn-diverge-decay`
        (runner.py:694-696: autograd of model.dp w.r.t. model.pts with create_graph) needs.  Encoders: none, HashEncoder."""
        assert self.latent_size == 0 and p.shape[-1] == self.in_size == 3 and e.shape == p.shape
        p, e = p.reshape(-1, 3).contiguous(), e.reshape(-1, 3).contiguous()
        N = p.shape[0]
        init, dinit = p, e
        if isinstance(self.enc, HashEncoder):
            tables = torch.stack([t.weight for t in self.enc.embs])  # differentiable view of the 8 parameters
            init = torch.cat([p, ag.HashEncodeFn.apply(p, tables, self.enc.include_input)], dim=-1)
            dinit = torch.cat([e, ag.HashJvpFn.apply(p, tables, e, self.enc.include_input)], dim=-1)
        elif self.enc is not None:
            raise NotImplementedError("differentiable direction tangents through " + type(self.enc).__name__)
        init, dinit = init.contiguous(), dinit.contiguous()
        z = ag.LinearFn.apply(init, None, self.init.weight, self.init.bias, "none")
        t = ag.LinearFn.apply(dinit, None, self.init.weight, None, "none")
        n = len(self.layers)
        lins = [(l, i != n - 1 and (i % self.skip) == 0) for i, l in enumerate(self.layers)] + [(self.out, False)]
        for lin, skip in lins:
            x_in = torch.cat([z, init], dim=-1).contiguous() if skip else z
            t_in = torch.cat([t, dinit], dim=-1).contiguous() if skip else t
            m = ag.ActDerivFn.apply(x_in, self.act_name)
            mt = ag.MulBcastFn.apply(m, t_in[None])[0]
            z = ag.LinearFn.apply(z, init if skip else None, lin.weight, lin.bias, self.act_name)
            t = ag.LinearFn.apply(mt.contiguous(), None, lin.weight, None, "none")
        return z, t

    def zero_last_layer(self):
        nn.init.zeros_(self.out.weight)
        nn.init.zeros_(self.out.bias)

    def uniform_last_layer(self, a=1e-4):
        nn.init.uniform_(self.out.weight, -a, a)
        nn.init.uniform_(self.out.bias, -a, a)
