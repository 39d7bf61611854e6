"""Process-wide knobs of the HIP path."""

# MLP arithmetic: "bf16x3" = 2-way split bf16 with 3 MFMA products (fp32-class accuracy, meets the
# 1e-4 RGB parity bound); "bf16" = plain bf16 operands with fp32 accumulation (fast mode); "f16" = IEEE half operands
# with fp32 accumulation: the fast mode's speed with 11-bit instead of 8-bit operands (L-inf 3e-4 instead of 3e-3 on the
# bench frame) -- implemented by the layer-synchronous renderers and the generic fused MLP kernels (not by the register-
# engine renderer, `set_engine("reg")`, which rejects it).  (Range: half saturates at 65504; the f16 kernels clamp instead of
# producing inf, see csrc/mlp_engine.h to_elem.)
# "f16x" = f16 main product + two MX-fp6 correction products (1.5 MFMA products per k, ~15-bit operands: a parity-class mode,
# L-inf ~2e-5): implemented by the layer-synchronous one-kernel renderers (PlainNeRF(view), TinyNeRF, VolSDF's View half and
# SIREN VolSDF) -- every other kernel (the generic fused MLPs, the register engine) runs its parity mode "bf16x3" when this is
# selected (`kernel_precision`).
precision = "bf16x3"


def set_precision(p: str):
    global precision
    if p not in ("bf16", "bf16x3", "f16", "f16x"):
        raise ValueError(p)
    precision = p


def kernel_precision(has_f16x: bool = False) -> str:
    """The precision a kernel family runs at: `precision`, except that "f16x" exists only where `has_f16x` says so and is
    replaced by the other parity-class mode, "bf16x3", everywhere else."""
    if precision == "f16x" and not (has_f16x and engine == "ls"):
        return "bf16x3"
    return precision


# f16x range guard at the model layer (the kernels' part: csrc/render_ls.hip g_lsx_saturated -- an activation at the half clamp
# turns the launch's WHOLE output into NaN, stream-ordered, no host synchronisation).  What the host does about it:
#   "nan"              (default) nothing: the caller sees the NaN frame; fully asynchronous
#   "raise"            every f16x launch is followed by a one-element read-back (a host synchronisation) and a flagged launch
#                      raises ops.F16xSaturated
#   "rerender_bf16x3"  as "raise", but the model's forward catches it and renders the same call again in bf16x3 (fp32 range)
f16x_on_saturation = "nan"


def set_f16x_on_saturation(policy: str):
    global f16x_on_saturation
    if policy not in ("nan", "raise", "rerender_bf16x3"):
        raise ValueError(policy)
    f16x_on_saturation = policy


class precision_as:
    """`with config.precision_as("bf16x3"): ...`"""

    def __init__(self, p: str):
        self.p = p

    def __enter__(self):
        self.prev = precision
        set_precision(self.p)

    def __exit__(self, *exc):
        set_precision(self.prev)


# Packed weight streams are cached per module and re-packed when a Parameter's version counter or address changes
# (utils.invalidate_packed documents what that misses: writes through `.data`, out-of-band copies).  True = pack on every
# forward instead (four small launches per model, ~30 us): for callers that update weights behind torch's back.
repack_always = False


def set_repack_always(on: bool):
    global repack_always
    repack_always = bool(on)


# Training-step GEMMs (forward / input gradient / weight gradient of every Linear while gradients are recorded):
# "bf16x3" = the same 2-way split on the bf16 matrix core (relative error ~2^-16 per GEMM, HBM-bound kernels);
# "fp32" = exact fp32 on the f32 matrix core (gradient-parity mode, ~4x slower per step).
train_precision = "bf16x3"


def set_train_precision(p: str):
    global train_precision
    if p not in ("fp32", "bf16x3"):
        raise ValueError(p)
    train_precision = p


class train_precision_as:
    """`with config.train_precision_as("fp32"): ...` -- the Linears recorded inside use that arithmetic (their backward too:
    the choice is stored with the graph node)."""

    def __init__(self, p: str):
        self.p = p

    def __enter__(self):
        self.prev = train_precision
        set_train_precision(self.p)

    def __exit__(self, *exc):
        set_train_precision(self.prev)


# Forward of a PlainNeRF(view) training step (round 6): "ls" = both networks in ONE launch of the layer-synchronous engine in the
# three-product split, every Linear's output rows written once for the backward pass (csrc/ls_kernel.h MODEL 9;
# PlainNeRF._train_forward_ls); "layers" = one training Linear per layer (csrc/train_fwd.hip).  Same arithmetic class, another
# summation order.  NA_TRAIN_LS=0 in the environment selects "layers" at import.
import os as _os
train_forward = "layers" if _os.environ.get("NA_TRAIN_LS") == "0" else "ls"


def set_train_forward(kind: str):
    global train_forward
    if kind not in ("ls", "layers"):
        raise ValueError(kind)
    train_forward = kind


# Fused PlainNeRF(view) renderer: "ls" = layer-synchronous engine (csrc/render_ls.hip: activations in LDS, weights
# streamed into registers, two sample groups in antiphase), "reg" = register-resident engine (csrc/render_fused.hip).
engine = "ls"


def set_engine(e: str):
    global engine
    if e not in ("ls", "reg"):
        raise ValueError(e)
    engine = e


# D-NeRF's deformation network in inference (precisions "f16x" / "bf16x3"; the fast modes run it in their own arithmetic):
#   "ls-bf16x3" (default since round 6): ONE launch of the layer-synchronous engine in the three-product bf16 split (csrc/render_ls.hip
#       MODEL 4, bf16x3) -- the accuracy class of the register engine's rows (1.3e-5 relative on the reference's golden weights), the
#       weights streamed once per 2 x NBLK blocks instead of per tile; up to 64 output rows (`--dyn-refl-latent`);
#   "generic": the register-engine fused MLP in the same three-product split (the default of rounds 2-5);
#   "ls": the layer-synchronous engine in f16x under precision "f16x" (MODEL 4, f16x: the fastest, but its rows carry 3x the error and
#       the canonical model's hash grid amplifies position errors -- on the reference's adversarial golden g9 the end-to-end RGB is
#       1.0-1.6e-4 with it, i.e. over north_star's 1e-4; 2.4e-5 on the trained model of tests/test_gpu_train.py).  Opt-in.
deformation_engine = "ls-bf16x3"


def set_deformation_engine(e: str):
    global deformation_engine
    if e not in ("generic", "ls", "ls-bf16x3"):
        raise ValueError(e)
    deformation_engine = e


# Reproducible training: gradients summed across workgroups (weight/bias gradients, hash-table scatter, d/dbeta)
# accumulate in 64-bit fixed point instead of fp32 atomics (na_set_deterministic): bitwise run-to-run reproducibility
# at ~the same speed.  Off by default (the fp32 atomics are the reference-like fast path).
_det_ws = None


def set_deterministic(on: bool, device="cuda"):
    """Process-wide.  Keeps a 16-MiB device workspace alive while on.  Limits (csrc/common.h): addends are quantised to
    2^-40 (anything below 4.5e-13 vanishes), the int64 sum holds |sum| < 8.4e6; addends of 2^20 and more and non-finite ones
    bypass the accumulator (plain fp32 atomics), so a diverged step still reads Inf / NaN.  The workspace is bound to
    `device`: deterministic entry points called with another current device raise, and they must not run concurrently on
    several streams."""
    global _det_ws
    import torch
    from . import _lib
    lib = _lib.load()
    if on:
        if _det_ws is None or str(_det_ws.device) != str(torch.device(device)):
            _det_ws = torch.empty(8 * 65536 * 4 * 8 + 4096, device=device, dtype=torch.uint8)
        _lib.check(lib.na_set_deterministic(_det_ws.data_ptr(), _det_ws.numel()))
    else:
        _lib.check(lib.na_set_deterministic(None, 0))
        _det_ws = None


def deterministic() -> bool:
    return _det_ws is not None
