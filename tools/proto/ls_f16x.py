#!/usr/bin/env python3
"""Prototype driver for tools/proto/ls_mlp_f16x.hip: the 1.5-product parity mode (f16 main product + two MX-fp6 correction
products on v_mfma_scale_f32_32x32x64_f8f6f4) on the layer-synchronous data flow.  VERDICT r02, "next round" item 1.

    python tools/proto/ls_f16x.py build            # here: hipcc -> tools/proto/lib_f16x_<variant>.so (they travel with gpurun)
    python tools/proto/ls_f16x.py run [L]          # on the GPU box: calibration, accuracy, timing of every built variant

What it measures: (i) the operand layout / scale semantics of the MX instructions (probed, not assumed); (ii) L-inf of the
HW result against an fp64 chain, next to the pure-f16 kernel on the same network; (iii) Msamples/s and the share of the bf16
MFMA peak for the full kernel and its ablations (no epilogue, no weight stream, pure MFMA loop).
"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {  # name -> (LS_ABLATE, extra flags)
    "full": (0, []), "f16only": (1, []), "noepi": (2, []), "now": (8, []), "noepi_now": (10, []), "loop": (2 | 8 | 128, []),
    "loop_f16": (1 | 2 | 8 | 128, []), "loop_mx": (2 | 8 | 128 | 256, []), "epi_light": (32, []), "f16only_noepi": (3, []),
    "lean": (0, ["-DLEAN=1"]), "lean_noprio": (0, ["-DLEAN=1", "-DPRIO=1"]), "lean_eprio": (0, ["-DLEAN=1", "-DPRIO=2"]),
    "noprio": (0, ["-DPRIO=1"]), "f16only_lean": (1, ["-DLEAN=1"]), "f16only_lean_noprio": (1, ["-DLEAN=1", "-DPRIO=1"]),
    "lean_now": (8, ["-DLEAN=1"]), "lean_now_noprio": (8, ["-DLEAN=1", "-DPRIO=1"]),
}


def lib_path(name):
    return os.path.join(HERE, f"lib_f16x_{name}.so")


def build():
    for name, (ab, extra) in VARIANTS.items():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                        f"-DLS_ABLATE={ab}", *extra, os.path.join(HERE, "ls_mlp_f16x.hip"), "-o", lib_path(name)], check=True)
        print(lib_path(name))


# ---- fp6 e2m3 on the host
def fp6_values():
    import torch
    v = [m * 0.125 for m in range(8)]
    for e in range(1, 4):
        v += [(1 + m / 8) * 2.0 ** (e - 1) for m in range(8)]
    return torch.tensor(v, dtype=torch.float64)  # code -> value, codes 0..31 (sign = bit 5)


def fp6_encode(q):
    """q float64 tensor (already divided by the block scale) -> uint8 codes, round to nearest, saturating at 7.5"""
    import torch
    vals = fp6_values()
    a = q.abs().clamp(max=7.5)
    idx = torch.bucketize(a, vals)  # first value >= a
    idx = idx.clamp(1, 31)
    lo, hi = vals[idx - 1], vals[idx]
    code = torch.where((a - lo) <= (hi - a), idx - 1, idx)
    tie = (a - lo) == (hi - a)
    code = torch.where(tie & (((idx - 1) & 1) == 1), idx, code)  # ties to even code
    return (code | ((q < 0).long() << 5)).to(torch.uint8)


def fp6_decode(code):
    vals = fp6_values()
    v = vals[(code & 31).long()]
    return v * (1 - 2 * ((code >> 5) & 1).double())


def pack_bits6(codes):
    """codes [..., 32] uint8 -> [..., 6] int32 (slot j at bits 6j..6j+5, little endian)"""
    import torch
    c = codes.to(torch.int64)
    out = torch.zeros(*codes.shape[:-1], 6, dtype=torch.int64)
    for j in range(32):
        bit = 6 * j
        w, s = bit // 32, bit % 32
        out[..., w] |= (c[..., j] << s) & 0xFFFFFFFF
        if s + 6 > 32:
            out[..., w + 1] |= c[..., j] >> (32 - s)
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out)
    return out.to(torch.int32)


def unpack_bits6(words):
    import torch
    w = words.to(torch.int64) & 0xFFFFFFFF
    codes = torch.zeros(*words.shape[:-1], 32, dtype=torch.int64)
    for j in range(32):
        bit = 6 * j
        k, s = bit // 32, bit % 32
        v = w[..., k] >> s
        if s + 6 > 32:
            v = v | (w[..., k + 1] << (32 - s))
        codes[..., j] = v & 63
    return codes.to(torch.uint8)


def calibrate(lib):
    """Returns (sigma, scale_div): the device conversion puts input position p into operand slot sigma[p]."""
    import torch
    dev = "cuda"
    fn = lib.ls_calib
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    lanes = torch.arange(64)
    row, half = lanes & 31, lanes >> 5
    ONE = 8  # code of 1.0
    res = {}

    def call(acodes, bcodes=None, bf=None, cscale=1.0, mode=0, sa=0x7F7F7F7F, sb=0x7F7F7F7F, opsel=0):
        araw = pack_bits6(acodes).contiguous().to(dev)
        braw = pack_bits6(bcodes if bcodes is not None else torch.zeros(64, 32, dtype=torch.uint8)).contiguous().to(dev)
        bfd = (bf if bf is not None else torch.zeros(64, 32)).float().contiguous().to(dev)
        d = torch.zeros(32, 32, device=dev)
        bout = torch.zeros(64, 6, dtype=torch.int32, device=dev)
        sa = sa - 2 ** 32 if sa >= 2 ** 31 else sa
        sb = sb - 2 ** 32 if sb >= 2 ** 31 else sb
        assert fn(araw.data_ptr(), braw.data_ptr(), bfd.data_ptr(), cscale, mode, sa, sb, opsel, d.data_ptr(), bout.data_ptr(), st) == 0
        torch.cuda.synchronize()
        return d.cpu(), unpack_bits6(bout.cpu())

    # E1: raw x raw pairing.  A row i (< 6): slot j of half h is 1.0 iff bit i of (j + 32 h)
    acodes = torch.zeros(64, 32, dtype=torch.uint8)
    for i in range(6):
        for h in range(2):
            for j in range(32):
                if ((j + 32 * h) >> i) & 1:
                    acodes[i + 32 * h, j] = ONE
    ok = True
    for hp in range(2):
        bcodes = torch.zeros(64, 32, dtype=torch.uint8)
        for n in range(32):
            bcodes[n + 32 * hp, n] = ONE
        d, _ = call(acodes, bcodes)
        idx = sum((d[i] > 0.5).long() << i for i in range(6))
        exp = torch.arange(32) + 32 * hp
        ok &= bool((idx == exp).all())
        if not (idx == exp).all():
            print("  E1 pairing half", hp, ":", idx.tolist())
    res["raw_slot_pairing_identity"] = ok
    # E2: order of the device conversions
    sig = {}
    for mode in (1, 2):
        bf = torch.zeros(64, 32)
        for n in range(32):
            bf[n, n] = 1.0
            bf[n + 32, n] = 1.0
        _, bout = call(acodes, bf=bf, mode=mode)
        s = []
        for n in range(32):
            nz = (bout[n] != 0).nonzero().flatten().tolist()
            s.append(nz[0] if len(nz) == 1 and int(bout[n, nz[0]]) == ONE else -1)
        sig[mode] = s
    res["sigma_2xpk16_f32"] = sig[1]
    res["sigma_pk32_f16"] = sig[2]
    # E3: scale semantics of the conversion: x = 1.0 at position 0, scale 2.0 / 0.5 / 3.0
    sem = {}
    for sc in (2.0, 0.5, 3.0, 1.0):
        bf = torch.zeros(64, 32)
        bf[:, 0] = 1.0
        _, bout = call(acodes, bf=bf, cscale=sc, mode=1)
        slot = sig[1][0]
        sem[sc] = float(fp6_decode(bout[0, slot]))
    res["cvt_of_1.0_by_scale"] = sem
    scale_div = 1 if abs(sem[2.0] - 0.5) < 1e-6 else 0
    res["scale_div"] = scale_div
    # rounding / saturation of the conversion
    bf = torch.zeros(64, 32)
    probe = [0.0625, 0.1875, 0.3125, 1.0625, 1.1875, 7.74, 7.76, 9.0, -0.0625, -7.76, 0.06, 0.07, 3.125, 3.375, 100.0, 1e-8]
    for i, v in enumerate(probe):
        bf[:, i] = v
    _, bout = call(acodes, bf=bf, mode=1)
    res["cvt_probe"] = {str(v): float(fp6_decode(bout[0, sig[1][i]])) for i, v in enumerate(probe)}
    # E4: MFMA scale bytes and op_sel: A = all ones in row 0, B one-hot; scale dword bytes 127,128,129,130
    a1 = torch.zeros(64, 32, dtype=torch.uint8)
    a1[0, :] = ONE
    a1[32, :] = ONE
    bcodes = torch.zeros(64, 32, dtype=torch.uint8)
    bcodes[:32, 0] = ONE
    sel = {}
    for k in range(4):
        d, _ = call(a1, bcodes, sa=0x8281807F, sb=0x7F7F7F7F, opsel=k)
        sel[k] = float(d[0, 0])
    res["mfma_scaleA_bytes_7f_80_81_82_by_opsel"] = sel
    selb = {}
    for k in range(4):
        d, _ = call(a1, bcodes, sa=0x7F7F7F7F, sb=0x8281807F, opsel=k)
        selb[k] = float(d[0, 0])
    res["mfma_scaleB_bytes_7f_80_81_82_by_opsel"] = selb
    return res


def pi16(r, h):  # accumulator register r of lane half h -> feature inside the 32-row tile
    return (r & 3) + 8 * (r >> 2) + 4 * h


def pack_weights(Wh, sigma, layout="v1"):
    """Wh: list of L [256,256] float64 -> the stream [L][4 rg][4 Q][REC bytes] (uint8 tensor) + emulation planes"""
    import torch
    L = len(Wh)
    REC = 8192 + 4 * 1536 + 256
    out = torch.zeros(L, 4, 4, REC, dtype=torch.uint8)
    lanes = torch.arange(64)
    rowl, h = lanes & 31, lanes >> 5
    inv = [0] * 32
    for p, s in enumerate(sigma):
        inv[s] = p
    emu = []
    for l in range(L):
        W = Wh[l]
        Wh16 = W.to(torch.float16).double()
        Wlo = W - Wh16
        Wl6v, Wt6v = torch.zeros_like(W), torch.zeros_like(W)
        for rg in range(4):
            for Q in range(4):
                rec = out[l, rg, Q]
                for t in range(2):
                    rows = 32 * (2 * rg + t) + rowl  # [64]
                    # f16 fragments: chunk c = 2 tt + qq, element e <-> producer register r = 8 qq + e of tile tt
                    for c in range(4):
                        tt, qq = c >> 1, c & 1
                        frag = torch.empty(64, 8, dtype=torch.float16)
                        for e in range(8):
                            cols = 64 * Q + 32 * tt + pi16(8 * qq + e, h)
                            frag[:, e] = W[rows, cols].to(torch.float16)
                        off = (t * 4 + c) * 1024
                        rec[off:off + 1024] = frag.view(torch.uint8).reshape(-1)
                    # fp6 operands: slot j <-> conversion input position p = inv[j]: tile p >> 4, register p & 15
                    cols = torch.stack([64 * Q + 32 * (inv[j] >> 4) + pi16(inv[j] & 15, h) for j in range(32)], dim=1)  # [64,32]
                    for which, src in ((0, Wlo), (1, W)):
                        blk = src[rows[:, None], cols]  # [64,32]
                        m = blk.abs().amax(dim=1)
                        e = torch.floor(torch.log2(m.clamp(min=2.0 ** -120))).long() - 2
                        e = e.clamp(min=-126)
                        codes = fp6_encode(blk / (2.0 ** e.double())[:, None])
                        dec = fp6_decode(codes) * (2.0 ** e.double())[:, None]
                        (Wl6v if which == 0 else Wt6v)[rows[:, None], cols] = dec
                        words = pack_bits6(codes)  # [64,6]
                        i = 2 * t + which
                        if layout == "v1":  # per operand: a 16-byte part and an 8-byte part per lane
                            base = 8192 + i * 1536
                            rec[base:base + 1024] = words[:, :4].contiguous().view(torch.uint8).reshape(-1)
                            rec[base + 1024:base + 1536] = words[:, 4:].contiguous().view(torch.uint8).reshape(-1)
                        else:  # "v2": per tile {WL6 | WT6} = 12 dwords per lane as three lane-linear 16-byte parts
                            base = 8192 + t * 3072
                            view = rec[base:base + 3072].view(torch.int32).reshape(3, 64, 4)
                            for dw in range(6):
                                g = 6 * which + dw
                                view[g // 4, :, g % 4] = words[:, dw]
                        sc = (e + 127).to(torch.uint8)
                        rec[8192 + 6144 + i:8192 + 6144 + 256:4] = sc
        emu.append((Wh16, Wl6v, Wt6v))
    return out, emu


def emulate(x, Wi, bi, Wh, bh, emu, mode):
    """fp64 chain with operand rounding.  mode: 'exact' | 'f16' | 'f16x' (per-(sample, 32-feature block of a row group) MX scales)"""
    import torch
    leaky = lambda t: torch.where(t > 0, t, 0.01 * t)
    f16 = lambda t: t.to(torch.float16).double()

    h = leaky(x @ Wi.T + bi)
    L = len(Wh)
    feats = torch.arange(256)
    Qf, hf = feats >> 6, (feats >> 2) & 1
    for l in range(L):
        if mode == "exact":
            z = h @ Wh[l].T + bh[l]
        else:
            hh16 = f16(h)
            z = hh16 @ emu[l][0].T + bh[l]
            if mode == "f16x":
                T6, R6 = torch.zeros_like(h), torch.zeros_like(h)
                res = h.float().double() - hh16  # the kernel works on fp32 activations
                for q in range(4):
                    for hv in range(2):
                        sel = (Qf == q) & (hf == hv)
                        blk = h[:, sel]
                        m = blk.abs().amax(dim=1)
                        ev = torch.floor(torch.log2(m.clamp(min=2.0 ** -120)))
                        sT, sR = 2.0 ** (ev - 2), 2.0 ** (ev - 13)
                        T6[:, sel] = fp6_decode(fp6_encode(blk / sT[:, None])) * sT[:, None]
                        R6[:, sel] = fp6_decode(fp6_encode(res[:, sel] / sR[:, None])) * sR[:, None]
                z = z + T6 @ emu[l][1].T + R6 @ emu[l][2].T
        if l == L - 1:
            return z[:, :32]
        h = leaky(z)


def run(L=12):
    import torch
    torch.manual_seed(0)
    dev = "cuda"
    lib0 = C.CDLL(lib_path("full"))
    cal = calibrate(lib0)
    print("calibration:", json.dumps(cal))
    sigma = cal["sigma_2xpk16_f32"]
    assert sorted(sigma) == list(range(32)), "conversion order not a permutation"
    N = 4 * 1024 * 1024
    Wi = (torch.randn(256, 16, dtype=torch.float64) * (6 / 16) ** 0.5 * 0.5).to(torch.float16).double()
    Wh = [torch.randn(256, 256, dtype=torch.float64) * (2 / 256) ** 0.5 for _ in range(L)]
    Wh = [w.float().double() for w in Wh]
    bi, bh = (torch.randn(256) * 0.1).double(), (torch.randn(L, 256) * 0.1).double()
    # init fragments, standard k order: lane l element e = Wi[row, 8 (l >> 5) + e]
    lanes = torch.arange(64)
    w_init = torch.empty(8, 64, 8, dtype=torch.float16)
    for t in range(8):
        for e in range(8):
            w_init[t, :, e] = Wi[32 * t + (lanes & 31), 8 * (lanes >> 5) + e].to(torch.float16)
    stream_w, emu = pack_weights(Wh, sigma)
    x = torch.randn(N, 16).to(torch.float16).float()
    M = 2048
    xr = x[:M].double()
    ref = emulate(xr, Wi, bi, Wh, bh, emu, "exact")
    e16 = emulate(xr, Wi, bi, Wh, bh, emu, "f16")
    e16x = emulate(xr, Wi, bi, Wh, bh, emu, "f16x")
    sc = float(ref.abs().max())
    print(f"host emulation over {M} samples (|ref| max {sc:.2f}): f16 L-inf {float((e16 - ref).abs().max()):.3e}, "
          f"f16 + fp6 corrections {float((e16x - ref).abs().max()):.3e}")
    lanes_h = torch.arange(2)
    b_pack = torch.empty(L, 4, 2, 2, 16)
    for rg in range(4):
        for t in range(2):
            for hh in range(2):
                for r in range(16):
                    b_pack[:, rg, t, hh, r] = bh[:, 64 * rg + 32 * t + pi16(r, hh)].float()
    b_pack = b_pack.contiguous().to(dev)
    d = dict(w_init=w_init.contiguous().to(dev), w=stream_w.contiguous().to(dev), bi=bi.float().to(dev),
             bh=bh.float().contiguous().to(dev), x=x.to(dev), y=torch.zeros(N, 32, device=dev))
    flop = 2 * (16 * 256 + L * 256 * 256)
    results = {}
    for name in VARIANTS:
        if not os.path.exists(lib_path(name)):
            continue
        lib = C.CDLL(lib_path(name))
        fn = lib.ls_mlp_forward_trace
        fn.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        d["y"].zero_()
        args = [d["w_init"].data_ptr(), d["w"].data_ptr(), d["bi"].data_ptr(), d["bh"].data_ptr(), d["x"].data_ptr(), d["y"].data_ptr(),
                N, L, cal["scale_div"], st, None, b_pack.data_ptr()]
        assert fn(*args) == 0
        torch.cuda.synchronize()
        err = float((d["y"][:M].cpu().double() - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn(*args)
        e0.record()
        for _ in range(10):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 10 * 1e-3
        tr = torch.zeros(2 * 5 * 64, dtype=torch.int64, device=dev)
        args[-2] = tr.data_ptr()
        fn(*args)
        torch.cuda.synchronize()
        t = tr.cpu().reshape(2, 5, 64)[:, :, 2:L - 1].double()
        line = f"{name:14s} L-inf {err:.3e}  {dt * 1e3:7.2f} ms  {N / dt / 1e6:7.0f} Msamples/s  {N * flop / dt / 2.5e15:6.1%} of bf16 peak"
        for g in range(2):
            E, w1, Mm, w2 = [(t[g, i + 1] - t[g, i]).mean() for i in range(4)]
            per = (t[g, 0, 1:] - t[g, 0, :-1]).mean()
            line += f" | g{g}: E {E:.0f} wait {w1:.0f} M {Mm:.0f} wait {w2:.0f} period {per:.0f}"
        print(line, flush=True)
        results[name] = dict(linf=err, ms=dt * 1e3, msamples=N / dt / 1e6, frac=N * flop / dt / 2.5e15)
    print(json.dumps(dict(L=L, N=N, ref_max=sc, calibration=cal, results=results)))


W4X_VARIANTS = {"w4x": (0, 0), "w4x_pipe": (0, 1), "w4x_now": (2, 0), "w4x_pipe_now": (2, 1)}


def build_w4x():
    for name, (ab, pipe) in W4X_VARIANTS.items():
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                        "-mllvm", "-pragma-unroll-threshold=1000000",  # the pipelined slots exceed the default 16 k: acc[] would stay in scratch
                        f"-DW4X_ABLATE={ab}", f"-DW4X_PIPE={pipe}", os.path.join(HERE, "ls_mlp_w4x.hip"), "-o", lib_path(name)], check=True)
        print(lib_path(name))


def run_w4x(L=12):
    """tools/proto/ls_mlp_w4x.hip: one wave per SIMD works for both sample groups, f16 weights resident per layer"""
    import torch
    torch.manual_seed(0)
    dev = "cuda"
    sigma = [2 * i for i in range(16)] + [2 * i + 1 for i in range(16)]  # probed by `run` (calibrate)
    N = 4 * 1024 * 1024
    Wi = (torch.randn(256, 16, dtype=torch.float64) * (6 / 16) ** 0.5 * 0.5).to(torch.float16).double()
    Wh = [(torch.randn(256, 256, dtype=torch.float64) * (2 / 256) ** 0.5).float().double() for _ in range(L)]
    bi, bh = (torch.randn(256) * 0.1).double(), (torch.randn(L, 256) * 0.1).double()
    lanes = torch.arange(64)
    w_init = torch.empty(8, 64, 8, dtype=torch.float16)
    for t in range(8):
        for e in range(8):
            w_init[t, :, e] = Wi[32 * t + (lanes & 31), 8 * (lanes >> 5) + e].to(torch.float16)
    stream_w, emu = pack_weights(Wh, sigma, layout="v2")
    b_pack = torch.empty(L, 4, 2, 2, 16)
    for rg in range(4):
        for t in range(2):
            for hh in range(2):
                for r in range(16):
                    b_pack[:, rg, t, hh, r] = bh[:, 64 * rg + 32 * t + pi16(r, hh)].float()
    x = torch.randn(N, 16).to(torch.float16).float()
    M = 4096
    sel = (torch.arange(M) % 128) >= 64  # the prototype reports the second sample group of every pass
    xr = x[:M].double()
    ref = emulate(xr, Wi, bi, Wh, bh, emu, "exact")[sel]
    e16 = emulate(xr, Wi, bi, Wh, bh, emu, "f16")[sel]
    sc = float(ref.abs().max())
    print(f"host emulation (|ref| max {sc:.2f}): f16 L-inf {float((e16 - ref).abs().max()):.3e}")
    d = dict(w_init=w_init.contiguous().to(dev), w=stream_w.contiguous().to(dev), bi=bi.float().to(dev), bp=b_pack.contiguous().to(dev),
             x=x.to(dev), y=torch.zeros(N, 32, device=dev))
    flop = 2 * (16 * 256 + L * 256 * 256)
    for name in W4X_VARIANTS:
        if not os.path.exists(lib_path(name)):
            continue
        lib = C.CDLL(lib_path(name))
        fn = lib.ls_w4x_forward
        fn.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        d["y"].zero_()
        args = [d["w_init"].data_ptr(), d["w"].data_ptr(), d["bi"].data_ptr(), d["bp"].data_ptr(), d["x"].data_ptr(), d["y"].data_ptr(),
                N, L, st, None]
        assert fn(*args) == 0
        torch.cuda.synchronize()
        err = float((d["y"][:M].cpu().double()[sel] - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn(*args)
        e0.record()
        for _ in range(10):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 10 * 1e-3
        tr = torch.zeros(64, dtype=torch.int64, device=dev)
        args[-1] = tr.data_ptr()
        fn(*args)
        torch.cuda.synchronize()
        t = tr.cpu()[: 2 * min(L, 32)].double()
        ph = (t[1:] - t[:-1])[2:-2]
        print(f"{name:14s} L-inf {err:.3e}  {dt * 1e3:7.2f} ms  {N / dt / 1e6:7.0f} Msamples/s  {N * flop / dt / 2.5e15:6.1%} of bf16 peak"
              f" | phase {ph.mean():.0f} cycles (96 MFMAs: {ph.mean() / 96:.1f} per MFMA)", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "build_w4x":
        build_w4x()
    elif sys.argv[1] == "run_w4x":
        run_w4x(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
