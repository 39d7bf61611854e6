"""Training and evaluation loops with the reference's recipe (runner.py:609-850 train, :855-995 test,
:1221-1322 main), restricted to what the five hot-path configs use: l2/l1/rmse loss in RGB, Adam(eps 1e-7) with the
cosine schedule, random crops/views from Python's `random`, pixel jitter 0.1, stratified sampling and density noise
in training mode, `--volsdf-scale-decay`, `--delta-x-decay`, `--offset-decay`, `--sdf-eikonal` (SDF normals by forward-mode
tangents through the MLP), `--smooth-normals` in the reference's default epsilon-perturbation form (`--smooth-eps`,
`--smooth-eps-rng`, `--smooth-n-ord`), `--dyn-diverge-decay` (one differentiable direction tangent), `--ffjord-div-decay`
(forward-mode divergence estimate); `--smooth-normals --smooth-eps 0` (double backward) raises.

Every forward and backward is a HIP kernel (nerf_atlas_amd/autograd.py); torch.optim owns the parameter update, like
in the reference.  With `replay_reference_rng=True` the stochastic tensors come from torch's CPU generator in the
reference's order, so a run is comparable with the reference's iteration by iteration (tests/test_gpu_train.py
against tests/golden/train_parity_*.json).
"""
import argparse
import os
import random
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import autograd as ag
from . import dist as na_dist
from . import config, loaders, nerf, refl, utils
from .render import render, render_frame

# the reference's CLI defaults for the fields used here (runner.py:38-424)
DEFAULTS = dict(
    data=None, data_kind="original", derive_kind=True, size=32, render_size=16, epochs=30000, batch_size=8,
    crop_size=16, test_crop_size=0, steps=64, mip=None, sigmoid_kind="upshifted", feature_space=3, model="plain",
    dyn_model=None, bg="black", learning_rate=5e-4, seed=1337, decay=0.0, loss_fns=["l2"], sched_min=5e-5,
    no_sched=False, shape_to_refl_size=64, refl_kind="view", space_kind="identity", normal_kind=None,
    refl_bidirectional=True, sdf_kind="mlp", near=2.0, far=6.0, spline=0, dyn_refl_latent=0, time_gamma=False,
    volsdf_scale_decay=0.0, delta_x_decay=0.0, opt_step=1, clip_gradients=0.0, train_imgs=-1, serial_idxs=False,
    higher_end_chance=0, opt_kind="adam", light_kind=None, occ_kind=None, volsdf_alternate=False, test_white_bg=False,
    sdf_eikonal=0.0, ffjord_div_decay=0.0, offset_decay=0.0, dyn_diverge_decay=0.0, smooth_normals=0.0, smooth_eps=1e-3,
    smooth_eps_rng=False, smooth_n_ord=[2],
    neural_upsample=False, quiet=False,
)

loss_map = {
    "l2": F.mse_loss,
    "l1": F.l1_loss,
    "rmse": lambda x, ref: F.mse_loss(x, ref).clamp(min=1e-10).sqrt(),
}


def make_args(**overrides) -> SimpleNamespace:
    """Namespace with the reference's defaults; mirrors the post-processing of runner.arguments() (:430-437)."""
    unknown = set(overrides) - set(DEFAULTS)
    assert not unknown, f"unknown arguments {sorted(unknown)}"
    a = SimpleNamespace(**{**DEFAULTS, **overrides})
    if not a.neural_upsample:
        a.render_size = a.size
        a.feature_space = 3
    if a.test_crop_size <= 0:
        a.test_crop_size = a.crop_size
    return a


def args_from_argv(argv) -> SimpleNamespace:
    """Parse the subset of the reference's command line that maps onto DEFAULTS (flags keep their spelling)."""
    p = argparse.ArgumentParser(allow_abbrev=False)
    p.add_argument("-d", "--data")
    for k, v in DEFAULTS.items():
        if k == "data":
            continue
        flag = "--" + k.replace("_", "-")
        if k == "learning_rate":
            p.add_argument("-lr", flag, type=float, default=v)
        elif isinstance(v, bool) and v:
            # defaults that are True (derive_kind, refl_bidirectional) get --flag / --no-flag so they can be switched off
            p.add_argument(flag, action=argparse.BooleanOptionalAction, default=True)
        elif isinstance(v, bool):
            p.add_argument(flag, action="store_true", default=False)
        elif isinstance(v, list):
            p.add_argument(flag, nargs="+", default=v, type=type(v[0]))
        elif v is None:
            p.add_argument(flag, default=None, type=(int if k == "spline" else str))
        else:
            p.add_argument(flag, type=type(v), default=v)
    p.add_argument("--nosave", action="store_true")
    p.add_argument("--notraintest", action="store_true")
    p.add_argument("--valid-freq", type=int, default=500)
    p.add_argument("--outdir", default="outputs/")
    ns = vars(p.parse_args(argv))
    return make_args(**{k: ns[k] for k in DEFAULTS if k in ns})


def seed(s):
    """runner.py:1214-1218."""
    if s == -1:
        return
    torch.manual_seed(s)
    random.seed(s)
    np.random.seed(s)


def load_model(args, is_dyn=False, device="cuda"):
    """runner.py:1169-1211 for the supported kinds."""
    if args.model == "sdf":
        raise NotImplementedError("--model sdf (surface rendering) is outside the volume-rendering hot path")
    model = nerf.load_nerf(args)
    if is_dyn:
        model = nerf.load_dyn(args, model, device)
    model.set_refl(refl.load(args, args.refl_kind, args.space_kind, model.intermediate_size))
    return model.to(device)


def load_loss_fn(args):
    fns = [loss_map[k] for k in args.loss_fns]
    assert len(fns) > 0, "must provide at least 1 loss function"
    if len(fns) == 1:
        return fns[0]
    return lambda x, ref: sum(fn(x, ref) for fn in fns) / len(fns)


class NaAdam(torch.optim.Adam):
    """torch.optim.Adam whose step is ONE kernel (na_adam_step) for the parameter groups it fits: fp32 contiguous device tensors with
    dense gradients, no amsgrad / maximize / weight decay / capturable / differentiable -- per element exactly the operations of
    torch's foreach implementation (torch/optim/adam.py::_multi_tensor_adam) in its order, each rounded like the ATen kernel rounds it
    (FMA_MASK: which multiply-adds ATen contracts).  Same state (`step`, `exp_avg`, `exp_avg_sq`): state_dicts interchange with
    torch.optim.Adam.  Any other group takes torch's own step."""
    FMA_MASK = 7

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            ok = (not group.get("amsgrad") and not group.get("maximize") and group.get("weight_decay", 0) == 0
                  and not group.get("capturable") and not group.get("differentiable") and not group.get("fused")
                  and not isinstance(group["lr"], torch.Tensor)
                  and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                          and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in ps))
            if not ok or not ps:
                self._torch_group_step(group)
                continue
            beta1, beta2 = group["betas"]
            exp_avgs, exp_avg_sqs, steps = [], [], []
            for p in ps:
                st = self.state[p]
                if len(st) == 0:  # (torch.optim.Adam._init_group's state)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.append(float(st["step"]))
                exp_avgs.append(st["exp_avg"])
                exp_avg_sqs.append(st["exp_avg_sq"])
            # one launch per distinct step count (parameters that joined later): the scalars as torch's Python computes them
            for t in sorted(set(steps)):
                idx = [i for i, s in enumerate(steps) if s == t]
                bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
                ops.adam_step([ps[i] for i in idx], [ps[i].grad for i in idx], [exp_avgs[i] for i in idx], [exp_avg_sqs[i] for i in idx],
                              1 - beta1, beta2, 1 - beta2, bc2 ** 0.5, group["eps"], (group["lr"] / bc1) * -1, self.FMA_MASK)
            # the kernel wrote the parameters through raw pointers: tell torch (an in-place update, like torch.optim.Adam's own) -- the
            # packed weight streams of the fused renderers are keyed on the version counters (utils.pack_stamp) and a validation render
            # between two training steps would otherwise keep the weights of its first call
            torch.autograd.graph.increment_version(ps)
        return loss

    def _torch_group_step(self, group):
        """torch.optim.Adam.step for ONE group (a group the kernel does not fit)"""
        keep = self.param_groups
        try:
            self.param_groups = [group]
            torch.optim.Adam.step(self)
        finally:
            self.param_groups = keep


def load_optim(args, params):
    """runner.py:448-458."""
    if args.opt_kind != "adam":
        raise NotImplementedError(f"opt kind {args.opt_kind}")
    params = list(params)
    # torch.optim.Adam's update (runner.py:448-458) as ONE launch where every parameter lives on the GPU (NaAdam below: the foreach
    # form's operations, order and rounding -- bit for bit the same trajectory, tests/test_gpu_train.py); NA_ADAM=torch keeps
    # torch's own seven launches.  (torch's `fused=True` is NOT used: its update rounds differently and the `--dyn-diverge-decay`
    # recipe, whose end point measures accumulation precision, leaves the reference's basin with it --
    # profiles/r06/adam_fused_dnerf_div.log.)
    cls = NaAdam if os.environ.get("NA_ADAM") != "torch" else torch.optim.Adam
    return cls(params, lr=args.learning_rate, eps=1e-7, weight_decay=args.decay)


def offset_decay_term(model, curr_percent: float):
    """NR-NeRF offset regulariser of `make dnerf` (runner.py:633-636,777-781): exp_ratio * mean(weights.detach() *
    (|dp|^(2 - rigidity) + 3e-3 * rigidity)), exp_ratio = (1/100)^(1 - progress).  Elementwise loss arithmetic on the
    model's stored [T,B,H,W,*] outputs; its gradient enters the HIP backward through BezierWarpFn's dp / rigidity
    inputs."""
    exp_ratio = (1 / 100) ** (1 - curr_percent)
    norm_dp = torch.linalg.vector_norm(model.dp, dim=-1, keepdim=True).pow(2 - model.rigidity)
    reg = model.canonical.weights.detach()[None, ..., None] * (norm_dp + 3e-3 * model.rigidity)
    return exp_ratio * reg.mean()


def train(model, cam, labels, opt, args, sched=None, on_iter=None, rank: int = 0, world: int = 1):
    """runner.py:609-850.  Returns the list of per-iteration l2 losses (what save_losses() plots).
    world > 1: data-parallel replicas (one process per GPU, identical seeds): every rank draws the same views and
    crop, renders views idxs[rank::world], and the gradients are averaged with one flat RCCL all-reduce before the
    optimiser step (dist.allreduce_gradients); the reference's own --data-parallel is broken (SURVEY header table)."""
    if args.epochs == 0:
        return []
    if getattr(args, "dyn_diverge_decay", 0) > 0 and not hasattr(model, "sum_jacobian_div"):
        raise ValueError("--dyn-diverge-decay needs a dynamic model (--data-kind dnerf --dyn-model plain --spline N): the "
                         "reference reads model.pts / model.dp (runner.py:694-696)")
    if args.smooth_normals > 0 and args.smooth_eps <= 0:
        raise NotImplementedError("--smooth-normals with --smooth-eps 0 differentiates the normals again (double backward); "
                                  "the epsilon-perturbation form (--smooth-eps > 0, the reference's default) is implemented")
    if args.smooth_normals > 0 and not hasattr(model, "sdf"):
        raise ValueError("--smooth-normals needs an SDF model (--model volsdf)")
    if args.ffjord_div_decay and not hasattr(model, "ffjord_div"):
        raise ValueError("--ffjord-div-decay needs a dynamic model (--data-kind dnerf --dyn-model plain --spline N): the "
                         "reference reads model.pts / model.rigid_dp (runner.py:698-699)")
    if args.sdf_eikonal > 0 and not hasattr(model, "sdf"):
        raise ValueError("--sdf-eikonal needs an SDF model (--model volsdf)")
    device = next(model.parameters()).device
    loss_fn = load_loss_fn(args)
    times = None
    if type(labels) is tuple:
        times = labels[-1].to(device)
        labels = labels[0]
    batch_size = min(args.batch_size, labels.shape[0])
    cs = args.crop_size
    if cs != 0:
        get_crop = lambda: (random.randint(0, args.render_size - cs), random.randint(0, args.render_size - cs), cs, cs)
    else:
        get_crop = lambda: (0, 0, args.size, args.size)
    train_choices = range(labels.shape[0])
    if args.higher_end_chance > 0:
        train_choices = list(train_choices) + [0] * args.higher_end_chance + [labels.shape[0] - 1] * args.higher_end_chance
    next_idxs = (lambda i: [i % len(cam)] * batch_size) if args.serial_idxs else (lambda _: random.sample(train_choices, batch_size))

    if world > 1 and batch_size % world != 0:
        # equal shards: the mean of the per-replica mean losses is then the single-process batch loss
        raise ValueError(f"--batch-size {batch_size} must be a multiple of the {world} training replicas "
                         f"(every replica renders batch_size / world views per step)")
    losses = []
    reg_terms = train.last_reg_terms = []  # (the --dyn-diverge-decay term per iteration: a diagnostic next to the losses)
    model.train()
    opt.zero_grad()
    for i in range(args.epochs):
        idxs = next_idxs(i)
        if world > 1:
            idxs = na_dist.shard_batch(idxs, rank, world)
        ts = None if times is None else times[idxs]
        c0, c1, c2, c3 = crop = get_crop()
        ref = labels[idxs][:, c0:c0 + c2, c1:c1 + c3, :3].to(device)
        out, _rays = render(model, cam[idxs], crop, size=args.render_size, times=ts)
        loss = loss_fn(out, ref)
        assert loss.isfinite(), f"Got {loss.item()} loss"
        losses.append(loss.item())
        if args.volsdf_scale_decay > 0 and isinstance(model, nerf.VolSDF):
            loss = loss + args.volsdf_scale_decay * model.scale_post_act
        if args.delta_x_decay > 0:
            loss = loss + args.delta_x_decay * model.dp.norm(dim=-1).mean()
        if args.offset_decay > 0:
            loss = loss + offset_decay_term(model, i / args.epochs) * args.offset_decay
        reg_pts = reg_n = None
        if args.sdf_eikonal > 0 or args.smooth_normals > 0:
            # runner.py:683-689: one set of 10240 points 5 * randn for both SDF regularisers; normals by forward-mode
            # tangents ([3, N], differentiable w.r.t. the weights with first-order autograd)
            reg_pts = 5 * utils.randn(((1 << 13) * 5 // 4, 3), device)
            # The smoothing term is a DIFFERENCE of two normals <= eps = 1e-3 apart: |delta n| ~ 1e-2 against |n| ~ 4, so the
            # 2^-16 relative error of the split-bf16 GEMMs is 400x larger relative to it (measured against the reference's
            # run: per-view PSNR 0.22 dB off instead of 0.1).  Its two tangent sweeps run in exact fp32 (10 240 points:
            # 0.1 ms), like the FFJORD tangent; the eikonal term alone keeps the configured arithmetic.
            with config.train_precision_as("fp32" if args.smooth_normals > 0 else config.train_precision):
                reg_n = model.sdf.underlying.normals_tangent_major(reg_pts)
        if args.sdf_eikonal > 0:
            # runner.py:691-692: E[|d sdf/dx|] = 1
            loss = loss + args.sdf_eikonal * ag.EikonalFn.apply(reg_n)
        if args.dyn_diverge_decay > 0:
            # runner.py:694-696: utils.divergence(model.pts, model.dp).mean(), with its graph (forward-mode tangent nodes)
            div_term = model.sum_jacobian_div().mean()
            reg_terms.append(float(div_term.detach()))
            loss = loss + args.dyn_diverge_decay * div_term
        if args.ffjord_div_decay:
            # runner.py:697-700: FFJORD divergence estimate of the rigid deformation, e = randn_like(rigid_dp).  The
            # reference's div_approx builds no graph (src/utils.py:471-477: autograd.grad without create_graph), so the
            # term shifts the loss value and advances the RNG stream but contributes no gradient -- same here.
            e = utils.randn(tuple(model.pts.shape), device)
            exp_ratio = (1 / 100) ** (1 - i / args.epochs)
            div = model.ffjord_div(e).abs().square()
            loss = loss + exp_ratio * args.ffjord_div_decay * (model.canonical.alpha.detach() * div).mean()
        if args.smooth_normals > 0:
            # runner.py:711-727, the epsilon-perturbation form "from unisurf": normals at the points and at points a random
            # direction of length eps away should agree.  Draw order of the reference: random.random() (with
            # --smooth-eps-rng), then randn_like(pts).
            s_eps = args.smooth_eps
            if args.smooth_eps_rng:
                s_eps = random.random() * s_eps
            perturb = F.normalize(utils.randn(tuple(reg_pts.shape), device), dim=-1) * s_eps
            with config.train_precision_as("fp32"):
                delta_n = reg_n - model.sdf.underlying.normals_tangent_major(reg_pts + perturb)  # [3, N]
            smoothness = 0
            for o in args.smooth_n_ord:
                smoothness = smoothness + torch.linalg.norm(delta_n, ord=int(o), dim=0).sum()
            loss = loss + args.smooth_normals * smoothness
        if args.opt_step != 1:
            loss = loss / args.opt_step
        loss.backward()
        if world > 1:
            na_dist.allreduce_gradients(model.parameters(), world)
        if args.clip_gradients > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_gradients)
        if i % args.opt_step == 0:
            opt.step()
            opt.zero_grad()
        if sched is not None:
            sched.step()
        if on_iter is not None:
            on_iter(i, losses[-1])
    return losses


def test(model, cam, labels, args):
    """runner.py:855-995: every view rendered in test_crop_size tiles without jitter; per-image PSNR."""
    device = next(model.parameters()).device
    times = None
    if type(labels) is tuple:
        times = labels[-1].to(device)
        labels = labels[0]
    if args.test_crop_size <= 0:
        args.test_crop_size = args.render_size
    psnrs, gots = [], []
    model.eval()
    for i in range(labels.shape[0]):
        ts = None if times is None else times[i:i + 1]
        exp = labels[i, ..., :3].to(device)
        got = render_frame(model, cam[i:i + 1], args.render_size, args.test_crop_size, times=ts)
        gots.append(got)
        psnrs.append(float(utils.mse2psnr(F.mse_loss(got, exp))))
    return psnrs, gots


def fit(args, device="cuda", replay_reference_rng=False, init=None, on_iter=None):
    """runner.main() (:1221-1322): seed, load the training set, build model + optimiser + schedule, train, load the
    test set, evaluate.  `init(model)` may overwrite the initial parameters (parity runs use procedural weights);
    with replay_reference_rng the torch stream is re-seeded with seed+1 after it, as tools/ref_train_fixture.py does."""
    rank, world = 0, 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # torchrun: one replica per GPU
        rank, world, local = na_dist.init_from_env()
        device = f"cuda:{local if torch.cuda.device_count() > local else 0}"  # debug: several replicas on one GPU
        torch.cuda.set_device(torch.device(device))
    seed(args.seed)
    labels, cam, _ = loaders.load(args, training=True)
    cam = cam.to(device)
    is_dyn = type(labels) is tuple
    model = load_model(args, is_dyn, device)
    if init is not None:
        init(model)
    prev = utils.random_source
    if replay_reference_rng:
        utils.set_random_source(utils.ReferenceStreamRandom())
        torch.manual_seed(args.seed + 1)
    try:
        if args.train_imgs > 0:
            labels = tuple(l[:args.train_imgs] for l in labels) if is_dyn else labels[:args.train_imgs]
            cam = cam[:args.train_imgs]
        model.nerf.steps, model.nerf.t_near, model.nerf.t_far = args.steps, args.near, args.far  # set_per_run :1048-1050
        opt = load_optim(args, model.parameters())
        sched = None if args.no_sched else torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.epochs,
                                                                                       eta_min=args.sched_min)
        losses = train(model, cam, labels, opt, args, sched=sched, on_iter=on_iter, rank=rank, world=world)
    finally:
        utils.set_random_source(prev)
    test_labels, test_cam, _ = loaders.load(args, training=False)
    test_cam = test_cam.to(device)
    if args.test_white_bg:
        model.set_bg("white")
    psnrs, gots = test(model, test_cam, test_labels, args)
    return dict(model=model, losses=losses, reg_terms=list(getattr(train, "last_reg_terms", [])), test_psnr=psnrs, test_psnr_mean=float(np.mean(psnrs)), frames=gots,
                test_labels=test_labels, rank=rank, world=world)
