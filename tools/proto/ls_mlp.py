#!/usr/bin/env python3
"""Prototype driver for tools/proto/ls_mlp.hip (layer-synchronous MLP forward): packs random weights, checks the result
against torch (bf16 operands, fp32 accumulate) and reports Msamples/s and the fraction of the bf16 MFMA peak.

    python tools/proto/ls_mlp.py build [w8|w4|w8s] [ABLATE]   # hipcc -> tools/proto/libls_proto.so (no GPU needed)
    python tools/proto/ls_mlp.py run [L]                      # on the GPU box

Variants (same interface, same packed weights):
  w8   8 waves x 32 output rows: ONE MFMA per B fragment read from LDS
  w4   4 waves x 64 rows (one wave per SIMD): TWO MFMAs per fragment read
  w8s  8 waves = 4 row groups x 2 sample halves: two MFMAs per read at two waves per SIMD
LS_ABLATE bits: 1 no per-layer weight refetch, 2 no epilogue/stores, 4 no per-layer barrier, 8 no LDS stores,
16 no epilogue VALU.  Measured (MI355X, L = 12, % of the 2.5 PFLOP/s bf16 peak):
  w8  full 43.5, pure MFMA+read loop 56.5 (independent of the prefetch depth 4/8/12)
  w4  full 40.0, no epilogue 70.5, no epilogue + no refetch 98.7, pure loop 101
  w8s full 36.5, no refetch 47.6, no epilogue 58.0, neither 91.4, pure loop 96.9
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libls_proto.so")


def build(variant="w8s", ablate=0, *extra):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                    f"-DLS_ABLATE={ablate}", *extra, os.path.join(HERE, f"ls_mlp_{variant}.hip"), "-o", LIB], check=True)
    with open(LIB + ".variant", "w") as f:
        f.write(variant)
    print(LIB)


def pack_rows(W, row0):
    """W [rows, K] fp32 -> fragments for rows row0..row0+31: [K/16 chunks][64 lanes][8] bf16."""
    import torch
    K = W.shape[1]
    lanes = torch.arange(64)
    rows = row0 + (lanes & 31)
    out = torch.empty(K // 16, 64, 8, dtype=torch.bfloat16)
    for c in range(K // 16):
        cols = 16 * c + 8 * (lanes >> 5)
        for e in range(8):
            out[c, :, e] = W[rows, cols + e].to(torch.bfloat16)
    return out


def pack_rows_perm(W, row0):
    """Like pack_rows, for the w8g variant: the k order inside chunk 2*tile+q is the producer's accumulator register
    order (lane half h, element e -> feature 32*tile + (r&3) + 8*(r>>2) + 4*h with r = 8*q + e)."""
    import torch
    K = W.shape[1]
    lanes = torch.arange(64)
    rows = row0 + (lanes & 31)
    h = lanes >> 5
    out = torch.empty(K // 16, 64, 8, dtype=torch.bfloat16)
    for c in range(K // 16):
        tile, q = c >> 1, c & 1
        for e in range(8):
            r = 8 * q + e
            cols = 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * h
            out[c, :, e] = W[rows, cols].to(torch.bfloat16)
    return out


def run(L=6, perm=False):
    import torch
    torch.manual_seed(0)
    dev = "cuda"
    N = 8 * 1024 * 1024
    Wi = torch.randn(256, 16) * (6 / 16) ** 0.5 * 0.5
    Wh = [torch.randn(256, 256) * (2 / 256) ** 0.5 for _ in range(L)]
    Wo = torch.randn(32, 256) * (1 / 256) ** 0.5
    bi, bh, bo = torch.randn(256) * 0.1, torch.randn(L, 256) * 0.1, torch.randn(32) * 0.1
    w_init = torch.stack([pack_rows(Wi, 32 * w) for w in range(8)]).contiguous().to(dev)
    pk = pack_rows_perm if perm else pack_rows
    w_hid = torch.stack([torch.stack([pk(Wh[l], 32 * w) for w in range(8)]) for l in range(L)]).contiguous().to(dev)
    w_out = pk(Wo, 0).contiguous().to(dev)
    if perm:  # w8g: one weight buffer, the out layer stored as layer L / tile 0
        pad = torch.zeros(1, 8, 16, 64, 8, dtype=torch.bfloat16, device=dev)
        pad[0, 0] = w_out
        w_hid = torch.cat([w_hid, pad]).contiguous()
    x = torch.randn(N, 16, device=dev)
    y = torch.empty(N, 32, device=dev)
    lib = C.CDLL(LIB)
    fn = lib.ls_mlp_forward
    fn.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_void_p]
    args = [w_init.data_ptr(), w_hid.data_ptr(), w_out.data_ptr(), bi.to(dev).data_ptr(), bh.to(dev).contiguous().data_ptr(),
            bo.to(dev).data_ptr(), x.data_ptr(), y.data_ptr(), N, L, torch.cuda.current_stream().cuda_stream]
    keep = (bi.to(dev), bh.to(dev).contiguous(), bo.to(dev))
    args[3], args[4], args[5] = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    assert fn(*args) == 0
    torch.cuda.synchronize()
    # reference: bf16 operands, fp32 accumulation, same rounding points
    M = 4096
    bf = lambda t: t.to(torch.bfloat16).float()
    h = bf(x[:M].cpu())
    h = torch.nn.functional.leaky_relu(h @ bf(Wi).T + bi, 0.01)
    for l in range(L):
        h = torch.nn.functional.leaky_relu(bf(h) @ bf(Wh[l]).T + bh[l], 0.01)
    ref = bf(h) @ bf(Wo).T + bo
    err = float((y[:M].cpu() - ref).abs().max())
    print(f"max |y - ref| over {M} samples: {err:.3e} (|ref| max {float(ref.abs().max()):.2f})")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn(*args)
    e0.record()
    for _ in range(10): fn(*args)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 10 * 1e-3
    flop = 2 * (16 * 256 + L * 256 * 256 + 256 * 32)
    if perm and hasattr(lib, "ls_mlp_forward_trace"):
        tr = torch.zeros(2 * 5 * 64, dtype=torch.int64, device=dev)
        ft = lib.ls_mlp_forward_trace
        ft.argtypes = fn.argtypes + [C.c_void_p]
        ft(*args, tr.data_ptr())
        torch.cuda.synchronize()
        t = tr.cpu().reshape(2, 5, 64)[:, :, 2:L - 1].double()
        for g in range(2):
            E, bw1, M, bw2 = (t[g, 1] - t[g, 0]).mean(), (t[g, 2] - t[g, 1]).mean(), (t[g, 3] - t[g, 2]).mean(), (t[g, 4] - t[g, 3]).mean()
            per = (t[g, 0, 1:] - t[g, 0, :-1]).mean()
            print(f"  trace group {g}: E {E:.0f}  wait {bw1:.0f}  M {M:.0f}  wait {bw2:.0f}  period {per:.0f} (s_memtime ticks, 100 MHz?)")
    print(f"L={L}: {dt * 1e3:.2f} ms for {N} samples = {N / dt / 1e6:.0f} Msamples/s, {N * flop / dt / 1e12:.0f} TFLOP/s "
          f"= {N * flop / dt / 2.5e15:.1%} of the bf16 MFMA peak")


if __name__ == "__main__":
    if sys.argv[1] == "build": build(*(sys.argv[2:3] or ["w8s"]), *(int(x) for x in sys.argv[3:4]), *sys.argv[4:])
    else:
        variant = open(LIB + ".variant").read().strip() if os.path.exists(LIB + ".variant") else ""
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 6, perm=variant.startswith("w8g"))
