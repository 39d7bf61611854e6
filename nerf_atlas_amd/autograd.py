"""torch.autograd glue for the training step.  Every forward AND backward below is a HIP kernel
behind the C ABI; torch only records the graph, slices/concatenates buffers and owns .grad accumulation.

Used by SkipConnMLP / the model classes whenever gradients are enabled and something requires them; inference
keeps using the fused MFMA kernels.
"""
import torch
from torch.autograd import Function

from . import config, ops


_NAN = {}


def _nan_scalar(device):
    t = _NAN.get(device)
    if t is None:
        t = _NAN[device] = torch.full((1, 1), float("nan"), device=device, dtype=torch.float32)
    return t


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class LinearFn(Function):
    """y = W . act([x0 | x1]) + b  (src/neural_blocks.py:288-296).  config.train_precision selects the three GEMMs:
    "bf16x3": split-bf16 MFMA kernels of csrc/train_gemm.hip (activation derivative fused into the input gradient);
    "fp32": exact f32-MFMA Linear, act_backward and weight-gradient kernels."""

    @staticmethod
    def forward(ctx, x0, x1, W, b, act, packs=None):
        """packs: None or (packed W, packed W^T | None) from ops.train_pack_many (SkipConnMLP._forward_train packs every Linear of
        the network with one launch per step) -- only meaningful in the "bf16x3" training arithmetic."""
        ctx.act = act
        ctx.fast = config.train_precision == "bf16x3"
        ctx.save_for_backward(x0, x1 if x1 is not None else torch.empty(0, device=x0.device), W)
        ctx.has_x1 = x1 is not None
        ctx.has_b = b is not None
        ctx.packed_t = packs[1] if (packs is not None and ctx.fast) else None
        return ops.linear_f32(x0, W, b, pre_act=act, x1=x1, split_bf16=ctx.fast,
                              packed=packs[0] if (packs is not None and ctx.fast) else None)

    @staticmethod
    def backward(ctx, gy):
        x0, x1, W = ctx.saved_tensors
        x1 = x1 if ctx.has_x1 else None
        gy = gy.contiguous()
        in0 = x0.shape[1]
        gx0 = gx1 = gW = gb = None
        want0, want1 = ctx.needs_input_grad[0], ctx.has_x1 and ctx.needs_input_grad[1]
        want_w = ctx.needs_input_grad[2] or (ctx.has_b and ctx.needs_input_grad[3])
        N, out = x0.shape[0], gy.shape[1]
        if (ctx.fast and want0 and want_w and ctx.packed_t is not None and (x1 is None or (in0 == 256 and x1.shape[1] <= 256))
                and ops.linear_bwd_fused_ok(N, out, in0)):
            # round 5: input gradient + weight gradient of a source in ONE pass over dY and that source (csrc/train_bwd.hip): the
            # 256 wide hidden input, an init Linear's narrow input, and the narrow second source of a skip layer as a call of its own
            in1 = x1.shape[1] if x1 is not None else 0
            gx0, gW, gb = ops.linear_bwd_fused(gy, x0, ctx.act, ctx.packed_t, in1=in1, want_bias=ctx.has_b)
            if x1 is not None:
                if want1 and ops.linear_bwd_fused_ok(N, out, in1):
                    gx1, _, _ = ops.linear_bwd_fused(gy, x1, ctx.act, ctx.packed_t, dW=gW, col0=in0)
                else:
                    ops.linear_wgrad_cols(x1, gy, ctx.act, gW, in0)
                    if want1:
                        _, gx1 = ops.linear_dgrad(gy, W, x0, ctx.act, x1, False, True, packed_t=ctx.packed_t)
            return gx0, gx1, gW, gb, None, None
        if (want0 or want1) and ctx.fast:
            gx0, gx1 = ops.linear_dgrad(gy, W, x0, ctx.act, x1, want0, want1, packed_t=ctx.packed_t)
        elif want0 or want1:
            g_act = ops.linear_f32(gy, W.t().contiguous(), None)  # [N, in0+in1] = dL/d act(x)
            if want0:
                g0 = g_act[:, :in0].contiguous()
                gx0 = ops.act_backward(x0, g0, ctx.act) if ctx.act != "none" else g0
            if want1:
                g1 = g_act[:, in0:].contiguous()
                gx1 = ops.act_backward(x1, g1, ctx.act) if ctx.act != "none" else g1
        if ctx.needs_input_grad[2] or (ctx.has_b and ctx.needs_input_grad[3]):
            gW, gb = ops.linear_wgrad(x0, gy, ctx.act, x1, want_bias=ctx.has_b, split_bf16=ctx.fast)
        return gx0, gx1, gW, gb, None, None


class HashEncodeFn(Function):
    """HashEncoder.forward (src/neural_blocks.py:139-193): gradients w.r.t. the tables (scatter) and, for
    deformation models whose canonical positions are predicted, w.r.t. the positions."""

    @staticmethod
    def forward(ctx, x, tables, include_input):
        ctx.save_for_backward(x, tables)
        ctx.include_input = include_input
        return ops.hash_encode(x, tables, include_input)

    @staticmethod
    def backward(ctx, g):
        x, tables = ctx.saved_tensors
        g = g.contiguous()
        gx = gt = None
        if ctx.needs_input_grad[0]:
            gx = ops.hash_encode_backward_input(x, tables, g, ctx.include_input)
        if ctx.needs_input_grad[1]:
            gt = ops.hash_encode_backward(x, g, ctx.include_input)
        return gx, gt, None


class HashInitFn(Function):
    """The init rows cat([p, enc(p)]) = [x | x | features] of a hash-encoded SkipConnMLP without a latent (src/neural_blocks.py:139-193,
    283-287) as ONE node and ONE kernel (round 6: hash_encode + cat forward; slice copy + scatter backward before).  The backward reads
    the rows' gradient in place: tables through the scatter kernel, positions (D-NeRF's warped points) through the input-gradient
    kernel, which adds both copies of x like autograd's accumulation did."""

    @staticmethod
    def forward(ctx, x, tables, include_input):
        ctx.save_for_backward(x, tables)
        ctx.include_input = include_input
        return ops.hash_encode_rows(x, tables, include_input, 1)

    @staticmethod
    def backward(ctx, g):
        x, tables = ctx.saved_tensors
        g = g.contiguous()
        gx = gt = None
        if ctx.needs_input_grad[0]:
            gx = ops.hash_encode_backward_input_rows(x, tables, g, ctx.include_input, 1)
        if ctx.needs_input_grad[1]:
            gt = ops.hash_encode_backward_rows(x, g, 3 * (1 + int(ctx.include_input)))
        return gx, gt, None


class PlainHeadFn(Function):
    """PlainNeRF between its two networks in training (src/nerf.py:338-357, src/refl.py:190-207): first_out [N, 1 + C] ->
    (density [N], the View MLP's init rows [N, 5 + C]) by one kernel; the backward writes [g_density | g_rows[:, 5:]] side by side
    (and the points' three columns when they carry a gradient: D-NeRF)."""

    @staticmethod
    def forward(ctx, first_out, pts, dirs, pre=None):
        if pre is not None:
            # (round 6: the one-launch training forward wrote density and rows itself; `first_out` is a placeholder nobody reads)
            return pre
        return ops.plain_head_rows(first_out, pts, dirs)

    @staticmethod
    def backward(ctx, g_density, g_rows):
        g_first, g_pts = ops.plain_head_rows_backward(g_density, g_rows.contiguous(), ctx.needs_input_grad[1])
        return g_first, g_pts, None, None


class HashJvpFn(Function):
    """d hash_encode(x)/dx . tangent (rows [tangent | per-level features]: na_hash_encode_jvp) as a graph node: the gradient
    w.r.t. the tables is the adjoint scatter na_hash_encode_jvp_backward (the features are linear in the tables); positions
    and tangent carry no gradient (floor() has none, and the callers' directions are constants)."""

    @staticmethod
    def forward(ctx, x, tables, tangent, include_input):
        ctx.save_for_backward(x, tangent)
        ctx.include_input = include_input
        return ops.hash_encode_jvp(x, tables, tangent, include_input)

    @staticmethod
    def backward(ctx, g):
        x, tangent = ctx.saved_tensors
        gt = ops.hash_encode_jvp_backward(x, tangent, g.contiguous(), ctx.include_input) if ctx.needs_input_grad[1] else None
        return None, gt, None, None


class LaplaceDensityFn(Function):
    """VolSDF density 1/beta * laplace_cdf(-sdf, beta) (src/nerf.py:985-990, src/utils.py:50-58)."""

    @staticmethod
    def forward(ctx, sdf, beta):
        ctx.save_for_backward(sdf, beta)
        return ops.laplace_density(sdf, beta)

    @staticmethod
    def backward(ctx, g):
        sdf, beta = ctx.saved_tensors
        g_sdf, g_beta = ops.laplace_density_backward(sdf, beta, g.contiguous(), want_beta=ctx.needs_input_grad[1])
        return g_sdf, (g_beta.reshape(beta.shape) if g_beta is not None else None)


class BezierWarpFn(Function):
    """DynamicNeRF spline warp (src/nerf.py:1267-1278) -> (warped pts, dp, rigidity [, refl_latent when n_rl > 0]); gradient
    w.r.t. the estimator output and the (pass-through) points."""

    @staticmethod
    def forward(ctx, est, pts, t, n_ctrl, n_rl=0):
        ctx.save_for_backward(est, t)
        ctx.n_ctrl, ctx.n_rl = n_ctrl, n_rl
        return ops.bezier_warp(est, pts, t, n_ctrl, n_rl)

    @staticmethod
    def backward(ctx, g_pts, g_dp, g_rig, g_enc=None):
        est, t = ctx.saved_tensors
        g_est = ops.bezier_warp_backward(est, t, ctx.n_ctrl, g_pts, g_dp, g_rig, ctx.n_rl, g_enc)
        return g_est, (g_pts if ctx.needs_input_grad[1] else None), None, None, None


class SigmoidFn(Function):
    @staticmethod
    def forward(ctx, x, kind):
        ctx.kind = kind
        ctx.save_for_backward(x)
        return ops.sigmoid(x, kind)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.sigmoid_backward(x, g.contiguous(), ctx.kind), None


class PosLinearCombineFn(Function):
    """(sigmoid(lin)/2 + 0.5) * pos[..., :C] (src/refl.py:288-290)."""

    @staticmethod
    def forward(ctx, lin, pos, c_out):
        ctx.c_out = c_out
        ctx.save_for_backward(lin, pos)
        with torch.no_grad():
            return ops.pos_linear_combine(lin, pos, c_out)

    @staticmethod
    def backward(ctx, g):
        lin, pos = ctx.saved_tensors
        g_lin, g_pos = ops.pos_linear_combine_backward(lin, pos, g.contiguous(), ctx.c_out, ctx.needs_input_grad[0],
                                                       ctx.needs_input_grad[1])
        return g_lin, g_pos, None


class ActDerivFn(Function):
    """m = act'(x) as a differentiable node (its own derivative is act''(x): -sin for sin, 0 for LeakyReLU)."""

    @staticmethod
    def forward(ctx, x, act):
        ctx.act = act
        ctx.save_for_backward(x)
        return ops.act_deriv(x, act, 1)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if ctx.act != "sin":
            return torch.zeros_like(x), None
        return ops.mul_bcast(ops.act_deriv(x, ctx.act, 2), g.contiguous().unsqueeze(0)).squeeze(0), None


class MulBcastFn(Function):
    """out[j] = a * b[j]: the multiplier row a [N,K] is shared by the J tangent rows b [J,N,K]."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.mul_bcast(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = ops.mul_reduce(g, b) if ctx.needs_input_grad[0] else None
        gb = ops.mul_bcast(a, g) if ctx.needs_input_grad[1] else None
        return ga, gb


class EikonalFn(Function):
    """utils.eikonal_loss (src/utils.py:31) on tangent-major normals [3, N]."""

    @staticmethod
    def forward(ctx, normals):
        ctx.save_for_backward(normals)
        return ops.eikonal_loss(normals)

    @staticmethod
    def backward(ctx, g):
        (normals,) = ctx.saved_tensors
        return ops.eikonal_loss_backward(normals, g.contiguous())


class CompositeFn(Function):
    """alpha_from_density + volumetric_integrate + sky (src/nerf.py:60-80,96-98).  Returns (out, alpha, weights);
    alpha/weights are auxiliary (non-differentiable) outputs."""

    @staticmethod
    def forward(ctx, density, feat, ts, rays, softplus, bg, rand=None):
        out, alpha, weights = ops.composite(density, feat, ts, rays, softplus=softplus, bg=bg, rand=rand)
        ctx.save_for_backward(density, feat, ts, rays)
        ctx.softplus, ctx.bg, ctx.rand = softplus, bg, rand  # (rand: the per-ray draw of bg "random", a constant)
        ctx.mark_non_differentiable(alpha, weights)
        ctx.set_materialize_grads(False)  # (no zero-filled [T, R] gradients for the two auxiliary outputs: two fill launches per step)
        return out, alpha, weights

    @staticmethod
    def backward(ctx, g_out, _ga, _gw):
        density, feat, ts, rays = ctx.saved_tensors
        gd, gf = ops.composite_backward(density, feat, ts, rays, g_out.contiguous(), ctx.softplus, ctx.bg, rand=ctx.rand)
        return gd, gf, None, None, None, None, None


class SplitHeadFn(Function):
    """(y[..., 0] contiguous, y[..., 1:] as a view) of a network output [.., 1 + C] -- PlainNeRF's density | intermediate
    (src/nerf.py:338-342).  The same values as the two slices; the point is the BACKWARD: autograd's own slice gradients are a
    zero-filled [.., 1 + C] tensor per slice plus their sum (fill + copy + fill + copy + add over a [N, 65] tensor: ~110 us per
    training step at N = 262 144); here the two gradients are written side by side into one uninitialised buffer: two copies."""

    @staticmethod
    def forward(ctx, y):
        ctx.shape = y.shape
        tail = y[..., 1:]
        return y[..., 0].contiguous(), tail

    @staticmethod
    def backward(ctx, g_head, g_tail):
        g = torch.empty(ctx.shape, device=(g_head if g_head is not None else g_tail).device, dtype=torch.float32)
        if g_head is not None:
            g[..., 0].copy_(g_head)
        else:
            g[..., 0].zero_()
        if g_tail is not None:
            g[..., 1:].copy_(g_tail)
        else:
            g[..., 1:].zero_()
        return g


class MlpTrainFn(Function):
    """A whole SkipConnMLP (src/neural_blocks.py:279-296) as ONE autograd node (round 5): the forward is the same chain of training
    Linears as LinearFn; the backward walks the layers itself -- the one-pass input + weight gradient kernel per source
    (csrc/train_bwd.hip) WITHOUT its reduction -- and sums the partial gradients of ALL Linears with one launch at the end
    (14 reduce launches and their gaps per PlainNeRF step before), with one host round trip through autograd instead of one per
    Linear.  Used when every Linear of the network takes the fused kernels (SkipConnMLP._mlp_fn_ok); otherwise the per-layer nodes."""

    @staticmethod
    def forward(ctx, init, spec, *params):
        """spec: {"act": str, "skips": [bool per hidden layer], "packs": [(packed W, packed W^T)] per Linear};
        params: W, b of init, the hidden layers, out (b may be None)."""
        act, skips, packs = spec["act"], spec["skips"], spec["packs"]
        Ws, bs = params[0::2], params[1::2]
        L = len(Ws)
        acts = ["none"] + [act] * (L - 1)
        x1s = [None] + [init if s else None for s in skips] + [None]
        pre = spec.get("pre")
        if pre is not None:
            # round 6: the forward has run already -- both networks of PlainNeRF in ONE launch of the layer-synchronous engine
            # (ops.train_plain_view_ls), which left every Linear's output rows in HBM: pre = ([L - 1 row tensors], the network's output)
            rows, x = pre
            assert len(rows) == L - 1 and all(r.shape == (init.shape[0], W.shape[0]) for r, W in zip(rows, Ws)), (L, [r.shape for r in rows])
            xs = [init] + list(rows)
            if x is None:
                # the network's output went straight into its consumer's format (PlainNeRF's `first`: density + the View MLP's init
                # rows, PlainHeadFn with `pre`): a placeholder of the right shape keeps the graph's edges, its values are never read
                # -- and if a future caller does read them it gets NaN, not stale memory (one cached scalar per device, broadcast: no launch)
                x = _nan_scalar(init.device).expand(init.shape[0], Ws[-1].shape[0])
        else:
            xs, x = [], init
            for li in range(L):
                xs.append(x)
                x = ops.linear_f32(x, Ws[li], bs[li], pre_act=acts[li], x1=x1s[li], split_bf16=True, packed=packs[li][0])
        ctx.save_for_backward(*xs, *Ws)
        ctx.L, ctx.acts, ctx.has_x1 = L, acts, [t is not None for t in x1s]
        ctx.has_b = [b is not None for b in bs]
        ctx.packed_t = [p[1] for p in packs]
        return x

    @staticmethod
    def backward(ctx, gy):
        L = ctx.L
        saved = ctx.saved_tensors
        xs, Ws = saved[:L], saved[L:]
        init = xs[0]
        g = gy.contiguous()
        pending, gWs, gbs, g_init = [], [None] * L, [None] * L, None
        for li in range(L - 1, -1, -1):
            x0, W = xs[li], Ws[li]
            out, in0 = W.shape[0], x0.shape[1]
            in1 = init.shape[1] if ctx.has_x1[li] else 0
            nW = out * (in0 + in1)
            pad = (-nW) % 4
            buf = torch.empty(nW + pad + (out if ctx.has_b[li] else 0), device=g.device, dtype=torch.float32)
            dW = buf[:nW].view(out, in0 + in1)
            db = buf[nW + pad:] if ctx.has_b[li] else None
            # (the init Linear's input gradient takes the skip layers' gradient of the same tensor as an addend: no add launch)
            fuse_add = li == 0 and g_init is not None and in0 <= 128
            gx0, ws, npart = ops.linear_bwd_partials(g, x0, ctx.acts[li], ctx.packed_t[li], 0, ctx.has_b[li], add=g_init if fuse_add else None)
            if fuse_add:
                g_init = None
            pending.append((ws, npart, out, in0, dW, 0, db))
            if in1:
                gx1, ws1, np1 = ops.linear_bwd_partials(g, init, ctx.acts[li], ctx.packed_t[li], in0, False)
                pending.append((ws1, np1, out, in1, dW, in0, None))
                g_init = gx1 if g_init is None else g_init + gx1
            gWs[li], gbs[li] = dW, db
            if li == 0:
                g_init = gx0 if g_init is None else g_init + gx0
            else:
                g = gx0
        ops.train_reduce_many(pending)
        grads = []
        for li in range(L):
            grads += [gWs[li], gbs[li]]
        return (g_init if ctx.needs_input_grad[0] else None, None, *grads)


class ViewInputFn(Function):
    """cat([x, dir_to_elev_azim(view)]) with the view direction of a ray broadcast along its samples (src/refl.py:190-207) as one
    kernel; the directions carry no gradient (they come from the camera), the points get their three columns back."""

    @staticmethod
    def forward(ctx, x, dirs):
        return ops.view_rows(x, dirs)

    @staticmethod
    def backward(ctx, g):
        return (g[..., :3] if ctx.needs_input_grad[0] else None), None
