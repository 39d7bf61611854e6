#!/usr/bin/env python3
"""In-kernel timeline of the layer-synchronous renderer: s_memtime stamps around every barrier of the second pass of
workgroup 0, for wave 0 (sample group 0) and wave 4 (group 1).

  python tools/ls_trace.py build          # here: gpurun_ablate/lib_lstrace.so  (-DNA_LS_TRACE=1 [extra flags])
  python tools/ls_trace.py run [bf16|bf16x3]   # on the GPU box: per-phase durations and barrier waits (shader cycles)
"""
import concurrent.futures as cf
import ctypes as C
import math
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_ablate")
sys.path.insert(0, REPO)
PHASES = ["EP (composite+hash)", "M first.init", "E1", "M first.L0", "E2", "M first.L1", "E3", "M first.L2", "E4",
          "M first.L3", "E5", "M first.out", "E6 (latent,geo)", "M view.init", "E7", "M view.L0", "E8", "M view.L1", "E9",
          "M view.L2", "E10", "M view.L3", "E11", "M view.out"]


def build(*extra):
    from nerf_atlas_amd import build as B
    d = os.path.join(OUT, "obj_lstrace")
    os.makedirs(d, exist_ok=True)

    def one(u):
        src, fl, suf = u[:3]
        o = os.path.join(d, os.path.splitext(src)[0] + suf + ".o")
        subprocess.run([B.hipcc()] + B.FLAGS + fl + ["-DNA_LS_TRACE=1", *extra, "-c", os.path.join(B.CSRC, src), "-o", o], check=True)
        return o
    with cf.ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, B.UNITS))
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "lib_lstrace.so")] + objs, check=True)
    print("built")


def run(prec="bf16"):
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
    packed = model.packed_ls(prec)
    lib = C.CDLL(os.path.join(OUT, "lib_lstrace.so"))
    lib.na_render_ls_workspace_bytes.restype = C.c_size_t
    lib.na_render_ls_workspace_bytes.argtypes = [C.c_int, C.c_int64]
    fn = lib.na_render_plain_view_ls
    fn.argtypes = _lib.SIGNATURES["na_render_plain_view_ls"][1]
    fn.restype = C.c_int
    nbytes = int(lib.na_render_ls_workspace_bytes(T, R))
    ws = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    out = torch.empty(R, 3, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(3):
        if i == 2: ev[0].record()
        rc = fn(rays.data_ptr(), None, R, ts.data_ptr(), T, tables.data_ptr(), packed.data_ptr(), ops.PREC[prec], 4, 0, None,
                None, out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    nb = (T + 31) // 32
    base = ((ws.data_ptr() + 255) & ~255) - ws.data_ptr() + R * 8
    base = ((ws.data_ptr() + base + 255) & ~255) - ws.data_ptr()
    raw = ws[base: base + 2 * 128 * 8].cpu().view(torch.int64).reshape(2, 128)
    print(f"{prec}: {ms:.2f} ms/frame = {R * T / ms / 1e3:.0f} Msamples/s (traced build)")
    ep = ws[base + 256 * 8: base + 256 * 8 + 2 * 16 * 8].cpu().view(torch.int64).reshape(2, 16)
    for g in range(2):
        st = [int(v) for v in ep[g][:8]]
        if st[0]:
            names = ["geom + composite(prev)", "", "level 0", "level 1", "level 2", "level 3", "pos chunk"]
            # (bf16x3: levels 2,3 are gathered by the helper row groups; their stamps are not written)
            last, parts = st[0], []
            for i, n in enumerate(names):
                if st[i + 1]:
                    parts.append(f"{n} {st[i + 1] - last}")
                    last = st[i + 1]
            print(f"   EP of group {g}: " + ", ".join(parts) + f" (total {st[7] - st[0]})")
            fine = [int(v) for v in ep[g][8:11]]
            if all(fine):
                print(f"      within the first piece: locate + ts loads issued {fine[0] - st[0]}, scalar ray loads {fine[1] - fine[0]}, "
                      f"positions (waits for ts) {fine[2] - fine[1]}, compositing of the previous pass {st[1] - fine[2]}")
    for g in range(2):
        t = [int(v) for v in raw[g][:48]]
        if t[0] == 0:
            print(f"group {g}: no trace"); continue
        # stamps: [pre-barrier k, post-barrier k] for k = 0..23; phase k runs from post-barrier k-1 to pre-barrier k
        print(f"== group {g} (wave {4 * g}): phase duration / barrier wait, shader cycles; pass total {t[47] - t[1] + (t[1] - t[0])}")
        tot_p = tot_w = 0
        for k in range(24):
            dur = t[2 * k] - t[2 * k - 1] if k > 0 else None
            wait = t[2 * k + 1] - t[2 * k]
            if dur is not None:
                tot_p += dur
            tot_w += wait
            print(f"  {PHASES[k]:22s} {'' if dur is None else dur:>7}  wait {wait:6d}")
        print(f"  sum of phases (w/o EP) {tot_p}, sum of waits {tot_w}")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](*sys.argv[2:])
