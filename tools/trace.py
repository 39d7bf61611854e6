#!/usr/bin/env python3
"""In-kernel timeline of the fused renderer (s_memtime stamps logged by waves 0 and NWAVES/2 of workgroup 0).

  python tools/trace.py build     # here: gpurun_ablate/lib_trace.so  (-DNA_TRACE=1)
  python tools/trace.py run       # on the GPU box: prints per-tile phase durations (cycles)
"""
import ctypes as C, math, os, subprocess, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_ablate")
CSRC = os.path.join(REPO, "nerf_atlas_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]
EV = {0: "tile_start", 1: "first_mfma", 2: "mid_arrive", 3: "vmcnt_done", 4: "barrier_done", 5: "dma_issued", 6: "tile_end",
      10: "pass_start", 11: "prologue_done", 12: "mlps_done", 13: "pass_end"}


def build():
    d = os.path.join(OUT, "obj_trace"); os.makedirs(d, exist_ok=True)
    objs = []
    for src, extra in [("basic_ops.hip", []), ("mlp_fused.hip", []), ("linear_f32.hip", []), ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=1"]),
                       ("render_fused.hip", ["-DNA_PREC_INST=1", "-DNA_TRACE=1"]), ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=0"]),
                       ("render_fused.hip", ["-DNA_PREC_INST=0", "-DNA_TRACE=1"])]:
        o = os.path.join(d, src.replace(".hip", "") + "".join(extra).replace("-D", "_").replace("=", "") + ".o")
        subprocess.run(["hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", o], check=True)
        objs.append(o)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "lib_trace.so")] + objs, check=True)
    print("built")


def run(prec="bf16"):
    import torch
    sys.path.insert(0, REPO)
    from nerf_atlas_amd import _lib, ops
    import nerf_atlas_amd.nerf as nerf
    dev = torch.device("cuda", 0); torch.manual_seed(2)
    size, T = 800, 128
    model = nerf.PlainNeRF(steps=T, t_near=2.0, t_far=6.0, intermediate_size=64, sigmoid_kind="upshifted").to(dev).eval()
    focal = 0.5 * size / math.tan(0.5 * 0.6911)
    c2w = torch.tensor([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]]], device=dev)
    rays = ops.raygen(c2w, focal, size, (0, 0, size, size)); ts, _ = ops.compute_ts(2.0, 6.0, T, dev)
    tables = model.first.enc.tables(); R = size * size
    _, pf = model.first.packed(prec, "plain_first"); _, pv = model.refl.mlp.packed(prec, "plain_view")
    lib = C.CDLL(os.path.join(OUT, "lib_trace.so"))
    lib.na_render_workspace_bytes.restype = C.c_size_t; lib.na_render_workspace_bytes.argtypes = [C.c_int, C.c_int64]
    fn = lib.na_render_plain_view; fn.argtypes = _lib.SIGNATURES["na_render_plain_view"][1]; fn.restype = C.c_int
    base = int(lib.na_render_workspace_bytes(T, R))
    ws = torch.zeros(base + 64 * 1024, device=dev, dtype=torch.uint8); out = torch.empty(R, 3, device=dev)
    for _ in range(2):
        rc = fn(rays.data_ptr(), R, ts.data_ptr(), T, tables.data_ptr(), pf.data_ptr(), pv.data_ptr(), ops.PREC[prec], 4, 0, None, None,
                out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    off = ((ws.data_ptr() + 255) & ~255) - ws.data_ptr() + R * 4 * 8 * 4
    raw = ws[off: off + 2 * 8 * 1401].cpu().view(torch.int64)
    for w in range(2):
        n = int(raw[w * 1401]); ev = [(int(v) >> 8, int(v) & 255) for v in raw[w * 1401 + 1: w * 1401 + 1 + n]]
        if not ev: continue
        t0 = ev[0][0]
        print(f"== wave {'0' if w == 0 else 'NW/2'}: {n} events")
        # per-tile phases
        agg = collections.defaultdict(list); last = {}
        tiles = []
        cur = {}
        for tm, e in ev:
            if e == 0: cur = {0: tm}
            elif e in (1, 2, 3, 4, 5): cur[e] = tm
            elif e == 6 and 0 in cur:
                cur[6] = tm; tiles.append(cur); cur = {}
        def seg(a, b): return [t[b] - t[a] for t in tiles if a in t and b in t]
        import statistics as st
        for name, a, b in [("start->first_mfma", 0, 1), ("first_mfma->mid", 1, 2), ("vmcnt wait", 2, 3), ("barrier wait", 3, 4),
                           ("dma issue", 4, 5), ("mid->end", 5, 6), ("tile total", 0, 6)]:
            x = seg(a, b)
            if x: print(f"  {name:20s} mean {st.mean(x):8.1f}  median {st.median(x):8.1f}  max {max(x):8d}  n={len(x)}")
        gaps = [tiles[i + 1][0] - tiles[i][6] for i in range(len(tiles) - 1)]
        print(f"  inter-tile gap        mean {st.mean(gaps):8.1f} median {st.median(gaps):8.1f} max {max(gaps)}")
        ps = [tm for tm, e in ev if e == 10]; pe = [tm for tm, e in ev if e == 13]; pr = [tm for tm, e in ev if e == 11]; pm = [tm for tm, e in ev if e == 12]
        for i in range(min(len(ps), len(pe))):
            print(f"  pass {i}: prologue {pr[i]-ps[i]}  mlps {pm[i]-pr[i]}  composite {pe[i]-pm[i]}  total {pe[i]-ps[i]}")
        # dump first 2 passes' tile table
        print("  tiles (start->first, first->mid, vm, bar, dma, mid->end): ")
        for t in tiles[:12] + tiles[40:52]:
            print("   ", [t.get(b, 0) - t.get(a, 0) for a, b in [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)]])


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]](*sys.argv[2:])
