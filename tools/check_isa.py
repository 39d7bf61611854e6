#!/usr/bin/env python3
"""ISA guard of the layer-synchronous renderer.

`render_ls_kernel` was once not bit-reproducible under timing changes; the events disappeared with `-fno-slp-vectorize`, i.e.
without compiler-formed packed fp32 arithmetic (`v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 ... op_sel`) in the kernel
(DESIGN.md 3b "Reproducibility"; the mechanism was never root-caused).  The build therefore checks the thing the flag is
for: `nerf_atlas_amd/build.py` compiles the three render_ls units with -save-temps and fails if any function whose symbol
contains `render_ls_kernel` holds one of those instructions.  This script runs the same check on the listings of the last
build (or on listings given on the command line) and prints what it scanned.

Second check (round 3): `v_cvt_scalef32_2xpk16_{fp6,bf6}_f32` writes its six destination registers while it still reads its
scale and the tails of its sources; the compiler does not mark the destination early-clobber, so a register allocation can
overlap them and the instruction then packs wrong values (tools/hw/cvt_fp6_overlap.hip; found through an
`-mllvm -amdgpu-sched-strategy=...` build whose fp6 weight pack differed).  Every instance of every function of the listing is
checked (`build.check_cvt_overlap`); the build fails on an offender.

    python tools/check_isa.py [listing.s ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_atlas_amd import build as B  # noqa: E402


def main(argv):
    if argv and argv[0] == "--bless-headline":
        path = [p for n, p in B.isa_listings() if n == "render_ls_f16x.o"][0]
        d = B.function_isa_digest(path, B.HEADLINE_SYMBOL)
        with open(B.HEADLINE_PIN, "w") as fh:
            fh.write(f"{d[0]}  {d[1]} instructions  render_ls_kernel<NA_PREC_F16X, 0>\n")
        print("pinned", d)
        return 0
    items = [(os.path.basename(p), p) for p in argv] or B.isa_listings()
    if not items:
        print("no listings: run `python -m nerf_atlas_amd.build` first")
        return 2
    rc = 0
    for name, path in items:
        bad, seen = B.check_isa(path)
        nbad = sum(len(v) for v in bad.values())
        print(f"{name}: {len(seen)} {B.ISA_KERNEL} instantiation(s) scanned, {nbad} packed-fp32 instruction(s)")
        for k, v in bad.items():
            print(f"  {k}: " + ", ".join(f"{m}@{n}" for n, m in v[:8]))
        cbad, ncvt = B.check_cvt_overlap(path)
        print(f"  {ncvt} multi-pass fp6 conversion(s) scanned, {len(cbad)} with an operand overlapping the destination")
        for n, text in cbad[:8]:
            print(f"    line {n}: {text}")
        if not seen or nbad or cbad:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
