"""Process-wide knobs of the HIP path."""

# MLP arithmetic: "bf16x3" = 2-way split bf16 with 3 MFMA products (fp32-class accuracy, meets the
# 1e-4 RGB parity bound); "bf16" = plain bf16 operands with fp32 accumulation (fast mode).
precision = "bf16x3"


def set_precision(p: str):
    global precision
    if p not in ("bf16", "bf16x3"):
        raise ValueError(p)
    precision = p


# Training-step GEMMs (forward / input gradient / weight gradient of every Linear while gradients are recorded):
# "bf16x3" = the same 2-way split on the bf16 matrix core (relative error ~2^-16 per GEMM, HBM-bound kernels);
# "fp32" = exact fp32 on the f32 matrix core (gradient-parity mode, ~4x slower per step).
train_precision = "bf16x3"


def set_train_precision(p: str):
    global train_precision
    if p not in ("fp32", "bf16x3"):
        raise ValueError(p)
    train_precision = p


# Fused PlainNeRF(view) renderer: "ls" = layer-synchronous engine (csrc/render_ls.hip: activations in LDS, weights
# streamed into registers, two sample groups in antiphase), "reg" = register-resident engine (csrc/render_fused.hip).
engine = "ls"


def set_engine(e: str):
    global engine
    if e not in ("ls", "reg"):
        raise ValueError(e)
    engine = e
