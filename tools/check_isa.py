#!/usr/bin/env python3
"""ISA guard of the layer-synchronous renderer.

`render_ls_kernel` was once not bit-reproducible under timing changes; the events disappeared with `-fno-slp-vectorize`, i.e.
without compiler-formed packed fp32 arithmetic (`v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 ... op_sel`) in the kernel
(DESIGN.md 3b "Reproducibility"; the mechanism was never root-caused).  The build therefore checks the thing the flag is
for: `nerf_atlas_amd/build.py` compiles the three render_ls units with -save-temps and fails if any function whose symbol
contains `render_ls_kernel` holds one of those instructions.  This script runs the same check on the listings of the last
build (or on listings given on the command line) and prints what it scanned.

    python tools/check_isa.py [listing.s ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_atlas_amd import build as B  # noqa: E402


def main(argv):
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
        if not seen or nbad:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
