#!/usr/bin/env python3
"""In-kernel timeline of the one-launch PlainNeRF + Positional / PosLinearView renderers (MODEL 7 / 8): s_memtime stamps around
every barrier of the second pass of workgroup 0, for wave 0 (sample group 0) and wave 4 (group 1).

  python tools/ls_trace.py build               # here: gpurun_ablate/lib_lstrace.so  (-DNA_LS_TRACE=1)
  python tools/head_trace.py run pos|plv [n_rl] # on the GPU box: per-phase durations and barrier waits (shader cycles)
"""
import ctypes as C
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_ablate")
sys.path.insert(0, REPO)
FIRST = ["EP", "M first.init", "E", "M first.L0", "E", "M first.L1", "E", "M first.L2", "E", "M first.L3", "E", "M first.out",
         "E first.out (latent, hash')", "M pos.init", "E", "M pos.L0a", "regen latent", "M pos.L0b", "E", "M pos.L1"]
PHASES = {"pos": FIRST + ["E", "M pos.L2", "E", "M pos.L3a", "regen latent", "M pos.L3b", "E", "M pos.L4", "E", "M pos.out"],
          "plv": FIRST + ["E", "M pos.out", "E pos.out (act, im, latent)", "M view.init", "E (sin groups)", "M view.L0", "E", "M view.L1",
                          "E", "M view.out"]}


def run(kind="pos", n_rl="0"):
    import torch
    import nerf_atlas_amd.nerf as nerf
    import nerf_atlas_amd.refl as refl
    from nerf_atlas_amd import _lib, ops
    n_rl = int(n_rl)
    dev = torch.device("cuda", 0)
    size, T = 800, 128
    torch.manual_seed(2)
    m = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted", bg="black")
    m.set_refl(refl.refl_kinds["pos" if kind == "pos" else "pos-linear-view"](latent_size=64 + n_rl, act="upshifted", out_features=3))
    m = m.to(dev).eval()
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, size, (0, 0, size, size))
    ts, _ = ops.compute_ts(2.0, 6.0, T, dev)
    R = size * size
    packed = m.packed_head_ls(kind, "f16x", n_rl)
    t1 = m.first.enc.tables()
    t2 = (m.refl.mlp if kind == "pos" else m.refl.pos).enc.tables()
    rl = torch.randn(T * R, max(n_rl, 1), device=dev) * 0.3
    lib = C.CDLL(os.path.join(OUT, "lib_lstrace.so"))
    lib.na_render_head_ls_workspace_bytes.restype = C.c_size_t
    lib.na_render_head_ls_workspace_bytes.argtypes = [C.c_int, C.c_int64]
    name = f"na_render_plain_{kind}_ls"
    fn = getattr(lib, name)
    fn.argtypes = _lib.SIGNATURES[name][1]
    fn.restype = C.c_int
    nbytes = int(lib.na_render_head_ls_workspace_bytes(T, R))
    ws = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
    out = torch.empty(R, 3, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(3):
        if i == 2: ev[0].record()
        if kind == "pos":
            rc = fn(rays.data_ptr(), None, R, ts.data_ptr(), T, t1.data_ptr(), t2.data_ptr(), packed.data_ptr(), ops.PREC["f16x"], 4, 0,
                    None, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        else:
            rc = fn(rays.data_ptr(), None, R, ts.data_ptr(), T, t1.data_ptr(), t2.data_ptr(), rl.data_ptr() if n_rl else None, max(n_rl, 1),
                    n_rl, packed.data_ptr(), ops.PREC["f16x"], 4, 0, None, None, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    base = ((ws.data_ptr() + 255) & ~255) - ws.data_ptr() + 256 * 2 * 2 * 1 * 8192
    raw = ws[base: base + 2 * 128 * 8].cpu().view(torch.int64).reshape(2, 128)
    print(f"{kind} n_rl={n_rl}: {ms:.2f} ms/frame = {R * T / ms / 1e3:.0f} Msamples/s (traced build)")
    names = PHASES[kind]
    ep = ws[base + 256 * 8: base + 256 * 8 + 2 * 16 * 8].cpu().view(torch.int64).reshape(2, 16)
    for g in range(2):
        st = [int(v) for v in ep[g][:12]]
        print(f"   stamps of group {g} (deltas):", [st[i + 1] - st[i] if st[i] and st[i + 1] else None for i in range(11)])
    for g in range(2):
        t = [int(v) for v in raw[g][:2 * len(names)]]
        if t[0] == 0:
            print(f"group {g}: no trace"); continue
        print(f"== group {g} (wave {4 * g}): phase duration / barrier wait, shader cycles; pass total {t[-1] - t[0]}")
        tot_p = tot_w = 0
        for k in range(len(names)):
            dur = t[2 * k] - t[2 * k - 1] if k > 0 else None
            wait = t[2 * k + 1] - t[2 * k]
            if dur is not None: tot_p += dur
            tot_w += wait
            print(f"  {names[k]:30s} {'' if dur is None else dur:>7}  wait {wait:6d}")
        print(f"  sum of phases (w/o EP) {tot_p}, sum of waits {tot_w}")


if __name__ == "__main__":
    run(*sys.argv[2:])
