#!/usr/bin/env python3
"""A/B of the training step between the shipped library and experiment builds (tools/ls_variant.py build-unit ...):
python tools/train_bench_ab.py shipped NAME [NAME ...] [--rounds 2]   -- one subprocess per library, interleaved"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--rounds"]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
code = ("import sys, runpy; sys.path.insert(0, {repo!r}); import nerf_atlas_amd._lib as L; L.LIB_PATH = {lib!r}; "
        "sys.argv = ['train_bench.py']; runpy.run_path({script!r}, run_name='__main__')")
for _ in range(rounds):
    for n in names:
        lib = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if n == "shipped" else os.path.join(REPO, "gpurun_ablate", f"lib_var_{n}.so")
        r = subprocess.run([sys.executable, "-c", code.format(repo=REPO, lib=lib, script=os.path.join(REPO, "tools", "train_bench.py"))],
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "ms_per_step" in l]
        print(f"{n:10s} {line[-1][line[-1].index('ms_per_step'):][:24] if line else r.stderr[-300:]}", flush=True)
