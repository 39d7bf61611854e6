#!/usr/bin/env python3
"""Ablation timing of the fused renderer (cdna_hip_programming.md: "ablate before optimising").

  python tools/ablate.py build   # here (no GPU): one .so per ablation mask under gpurun_ablate/
  python tools/ablate.py run     # on the GPU box: time each variant on the 800x800x128 frame

Masks (csrc/mlp_engine.h NA_ABLATE): 1 no barrier  2 identity act  4 no MFMA  8 one LDS read/tile  16 no DMA  32 no hash.
Results with a non-zero mask are numerically wrong; only the time is meaningful.
"""
import concurrent.futures as cf
import ctypes as C
import json
import math
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_ablate")
CSRC = os.path.join(REPO, "nerf_atlas_amd", "csrc")
MASKS = [int(m) for m in os.environ.get('NA_MASKS', '0,28,29,31,63,61,60').split(',')]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]


def build_one(mask, prec=0):
    d = os.path.join(OUT, f"obj_{mask}_{prec}")
    os.makedirs(d, exist_ok=True)
    objs = []
    for src, extra in [("basic_ops.hip", []), ("mlp_fused.hip", []), ("linear_f32.hip", []),
                       ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=1"]), ("render_fused.hip", ["-DNA_PREC_INST=1"]),
                       ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=0"]), ("render_fused.hip", ["-DNA_PREC_INST=0"])]:
        # only the render kernels carry the ablation; the generic MLP units are needed for symbols
        o = os.path.join(d, src.replace(".hip", "") + "".join(extra).replace("-D", "_").replace("=", "") + ".o")
        abl = ([f"-DNA_ABLATE={mask & 63}"] + ([f"-DNA_KSTAGE={mask >> 8}"] if mask >> 8 else [])) if src == "render_fused.hip" else []
        if src == "mlp_fwd_inst.hip" and mask != 0:
            o0 = o.replace(f"obj_{mask}_", "obj_0_")
            if os.path.exists(o0):
                objs.append(o0)
                continue
        subprocess.run(["hipcc"] + FLAGS + extra + abl + ["-c", os.path.join(CSRC, src), "-o", o], check=True)
        objs.append(o)
    lib = os.path.join(OUT, f"lib_abl{mask}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def build():
    os.makedirs(OUT, exist_ok=True)
    build_one(0)
    with cf.ThreadPoolExecutor(8) as ex:
        list(ex.map(build_one, [m for m in MASKS if m != 0]))
    print("built", sorted(os.listdir(OUT)))


def run(precisions=("bf16",)):
    import torch
    sys.path.insert(0, REPO)
    from nerf_atlas_amd import _lib, ops
    import nerf_atlas_amd.nerf as nerf
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    size, T = 800, 128
    model = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev).eval()
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, size, (0, 0, size, size))
    ts, _ = ops.compute_ts(2.0, 6.0, T, dev)
    tables = model.first.enc.tables()
    R = size * size
    res = {}
    for prec in precisions:
        _, pf = model.first.packed(prec, "plain_first")
        _, pv = model.refl.mlp.packed(prec, "plain_view")
        for mask in MASKS:
            path = os.path.join(OUT, f"lib_abl{mask}.so")
            if not os.path.exists(path):
                continue
            lib = C.CDLL(path)
            lib.na_render_workspace_bytes.restype = C.c_size_t
            lib.na_render_workspace_bytes.argtypes = [C.c_int, C.c_int64]
            fn = lib.na_render_plain_view
            fn.argtypes = _lib.SIGNATURES["na_render_plain_view"][1]
            fn.restype = C.c_int
            ws = torch.empty(int(lib.na_render_workspace_bytes(T, R)), device=dev, dtype=torch.uint8)
            out = torch.empty(R, 3, device=dev)
            def call():
                rc = fn(rays.data_ptr(), R, ts.data_ptr(), T, tables.data_ptr(), pf.data_ptr(), pv.data_ptr(),
                        ops.PREC[prec], 4, 0, None, None, out.data_ptr(), ws.data_ptr(), ws.numel(),
                        torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            call(); torch.cuda.synchronize()
            times = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); torch.cuda.synchronize()
                times.append(a.elapsed_time(b))
            res[f"{prec}:{mask}"] = min(times)
            print(f"{prec} mask {mask:3d}: {min(times):8.2f} ms  ({R * T / min(times) / 1e3:8.1f} Msamples/s)", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
