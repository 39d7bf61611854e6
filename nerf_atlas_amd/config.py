"""Process-wide knobs of the HIP path."""

# MLP arithmetic: "bf16x3" = 2-way split bf16 with 3 MFMA products (fp32-class accuracy, meets the
# 1e-4 RGB parity bound); "bf16" = plain bf16 operands with fp32 accumulation (fast mode).
precision = "bf16x3"


def set_precision(p: str):
    global precision
    if p not in ("bf16", "bf16x3"):
        raise ValueError(p)
    precision = p
