#!/usr/bin/env python3
"""A/B builds of ONE render_ls unit with extra -D flags, timed on the full bench frame.

  python tools/ls_variant.py build NAME [--prec f16x] -DNA_FOO=1 ...   # here: gpurun_ablate/lib_var_NAME.so
  python tools/ls_variant.py run NAME [NAME ...] [--prec f16x]         # on the GPU box: ms / frame, Msamples/s, L-inf vs the
                                                                       # shipped library's frame (experiments may be wrong on purpose)
The other objects come from nerf_atlas_amd/build/ (run `python -m nerf_atlas_amd.build` first).
"""
import ctypes as C
import math
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_ablate")
sys.path.insert(0, REPO)
SUFFIX = {"bf16": "_bf16", "bf16x3": "_bf16x3", "f16": "_f16", "f16x": "_f16x"}


def pop_prec(argv):
    prec = "f16x"
    if "--prec" in argv:
        i = argv.index("--prec")
        prec = argv[i + 1]
        del argv[i:i + 2]
    return prec


def build(name, prec, extra, src="render_ls.hip"):
    from nerf_atlas_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    unit = [u for u in B.UNITS if u[0] == src and (src != "render_ls.hip" or u[2] == SUFFIX[prec])][0]
    o = os.path.join(OUT, f"var_{name}.o")
    subprocess.run([B.hipcc()] + B.FLAGS + unit[1] + list(extra) + ["-c", os.path.join(B.CSRC, unit[0]), "-o", o], check=True)
    objs = [B._obj_path(u) for u in B.UNITS if u is not unit]
    subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", os.path.join(OUT, f"lib_var_{name}.so"), o] + objs, check=True)
    os.remove(o)
    print("built", name, " ".join(extra))


def run(names, prec):
    import torch
    import bench
    from nerf_atlas_amd import _lib, ops
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    size, T = bench.SIZE, bench.STEPS_PER_RAY
    focal = 0.5 * size / math.tan(0.5 * bench.FOV)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, size, (0, 0, size, size))
    ts, _ = ops.compute_ts(bench.NEAR, bench.FAR, T, dev)
    tables = model.first.enc.tables()
    R = size * size
    ref = None
    for name in ["shipped"] + list(names):
        path = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if name == "shipped" else os.path.join(OUT, f"lib_var_{name}.so")
        lib = C.CDLL(path)
        lib.na_render_ls_workspace_bytes.restype = C.c_size_t
        lib.na_render_ls_workspace_bytes.argtypes = [C.c_int, C.c_int64]
        lib.na_render_ls_packed_bytes.restype = C.c_size_t
        fn = lib.na_render_plain_view_ls
        fn.argtypes = _lib.SIGNATURES["na_render_plain_view_ls"][1]
        fn.restype = C.c_int
        # every variant packs its own stream (experiments may change the layout)
        packed = pack_with(lib, model, prec, dev)
        ws = torch.zeros(int(lib.na_render_ls_workspace_bytes(T, R)), device=dev, dtype=torch.uint8)
        out = torch.empty(R, 3, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 5
        for i in range(2 + n):
            if i == 2: ev[0].record()
            rc = fn(rays.data_ptr(), None, R, ts.data_ptr(), T, tables.data_ptr(), packed.data_ptr(), ops.PREC[prec], 4, 0, None,
                    None, out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (name, rc)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        if ref is None: ref = out.clone()
        print(f"{name:>14s} [{prec}]: {ms:8.2f} ms/frame = {R * T / ms / 1e3:6.0f} Msamples/s   L-inf vs shipped {float((out - ref).abs().max()):.3e}", flush=True)


def pack_with(lib, model, prec, dev):
    """the stream of model (PlainNeRF(view)) packed by THIS library"""
    import torch
    from nerf_atlas_amd import ops
    first = [model.first.init, *model.first.layers, model.first.out]
    view = [model.refl.mlp.init, *model.refl.mlp.layers, model.refl.mlp.out]
    lib.na_render_ls_packed_bytes.argtypes = [C.c_int]
    packed = torch.zeros(int(lib.na_render_ls_packed_bytes(ops.PREC[prec])), dtype=torch.uint8, device=dev)
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    keep = [l.weight.detach().float().contiguous() for l in first + view] + [l.bias.detach().float().contiguous() for l in first + view]
    nf, nv = len(first), len(view)
    w0, w1 = keep[:nf], keep[nf:nf + nv]
    b0, b1 = keep[nf + nv:2 * nf + nv], keep[2 * nf + nv:]
    lib.na_render_ls_pack.argtypes = None
    lib.na_render_ls_pack.restype = C.c_int
    rc = lib.na_render_ls_pack(C.c_int(ops.PREC[prec]), arr(w0), arr(b0), arr(w1), arr(b1), C.c_void_p(packed.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return packed


if __name__ == "__main__":
    argv = sys.argv[1:]
    prec = pop_prec(argv)
    if argv[0] == "build":
        build(argv[1], prec, argv[2:])
    elif argv[0] == "build-unit":  # build-unit SRC NAME flags...: any other unit (time it with the tool of that unit)
        build(argv[2], prec, argv[3:], src=argv[1])
    else:
        run(argv[1:], prec)
