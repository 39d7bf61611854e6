#!/usr/bin/env python3
"""Scheduler fuzzing: the same sources compiled under LLVM's other AMDGPU instruction schedulers must give the same BITS.
A kernel whose results change with the schedule leans on something the compiler does not know -- in round 3 this found the fp6
conversion that reads its scale after it starts writing its destination and a hand-issued load path of the training
GEMMs whose in-flight registers the compiler copied.

    python tools/sched_fuzz.py build          # here (hipcc cross-compiles): gpurun_ablate/lib_var_<unit>_<strategy>.so
    python tools/sched_fuzz.py run            # on the GPU box: every variant against the shipped library, bit for bit

The whole library under another scheduler (round 3: max-ilp, 248 GPU tests green; iterative-ilp fails to compile basic_ops.hip):
    build it into a side directory (nerf_atlas_amd.build with OBJ / LIB / FLAGS pointed elsewhere) and run
    NA_LIB_PATH=<that>/libnerf_atlas_amd.so python -m pytest tests -m gpu -q
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRATEGIES = ("max-ilp", "iterative-ilp")
PRECS = ("f16x", "bf16x3", "bf16", "f16")
LSV = os.path.join(REPO, "tools", "ls_variant.py")


def tag(s):
    return s.split("-")[0]


def build():
    for st in STRATEGIES:
        flags = ["-mllvm", f"-amdgpu-sched-strategy={st}"]
        for prec in PRECS:
            subprocess.run([sys.executable, LSV, "build", f"{prec}_{tag(st)}", "--prec", prec] + flags, check=True)
        subprocess.run([sys.executable, LSV, "build-unit", "train_gemm.hip", f"tg_{tag(st)}"] + flags, check=True)
        for unit, t in (("train_bwd.hip", "tb"), ("train_fwd.hip", "tf")):  # round 5 (iterative-ilp crashes the compiler on both: skipped)
            r = subprocess.run([sys.executable, LSV, "build-unit", unit, f"{t}_{tag(st)}"] + flags)
            if r.returncode != 0:
                print(f"{unit} does not compile under {st}: no variant")


def run():
    import torch
    rc = 0
    for prec in PRECS:
        r = subprocess.run([sys.executable, LSV, "run"] + [f"{prec}_{tag(st)}" for st in STRATEGIES] + ["--prec", prec], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if "L-inf vs shipped" in l]
        print("\n".join(lines), flush=True)
        if r.returncode != 0 or any(not l.rstrip().endswith("0.000e+00") for l in lines) or len(lines) != 1 + len(STRATEGIES):
            rc = 1
    code = ("import sys, runpy; sys.path.insert(0, {repo!r}); import nerf_atlas_amd._lib as L; L.LIB_PATH = {lib!r}; "
            "sys.argv = ['x', {out!r}]; runpy.run_path({script!r}, run_name='__main__')")
    outs = {}
    names = ["shipped"] + [f"tg_{tag(st)}" for st in STRATEGIES]
    for n in names:
        lib = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if n == "shipped" else os.path.join(REPO, "gpurun_ablate", f"lib_var_{n}.so")
        out = f"/tmp/sched_fuzz_{n}.pt"
        r = subprocess.run([sys.executable, "-c", code.format(repo=REPO, lib=lib, out=out, script=os.path.join(REPO, "tools", "gemm_ls_vs_tiled.py"))],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(n, "failed:", r.stderr[-800:])
            return 1
        outs[n] = torch.load(out)
    for n in names[1:]:
        same, worst = True, 0.0
        for k in outs["shipped"]:
            for a, b in zip(outs["shipped"][k], outs[n][k]):
                if a is None:
                    continue
                same = same and bool(torch.equal(a, b))
                worst = max(worst, float((a - b).abs().max()))
        print(f"{n:>14s} [training GEMMs, forward + input gradient of 5 shapes]: bit-identical to shipped {same} (max abs diff {worst:.2e})")
        rc = rc or (0 if same else 1)
    # round 5: the one-pass backward and the register-resident forward (each unit under each strategy, against the shipped library)
    fouts = {}
    fnames = ["shipped"] + [f"{u}_{tag(st)}" for u in ("tb", "tf") for st in STRATEGIES
                            if os.path.exists(os.path.join(REPO, "gpurun_ablate", f"lib_var_{u}_{tag(st)}.so"))]
    for n in fnames:
        lib = os.path.join(REPO, "nerf_atlas_amd", "libnerf_atlas_amd.so") if n == "shipped" else os.path.join(REPO, "gpurun_ablate", f"lib_var_{n}.so")
        out = f"/tmp/sched_fuzz_fused_{n}.pt"
        r = subprocess.run([sys.executable, "-c", code.format(repo=REPO, lib=lib, out=out, script=os.path.join(REPO, "tools", "train_fused_dump.py"))],
                           capture_output=True, text=True, env=dict(os.environ, NA_TRAIN_FUSED_FWD="all"))
        if r.returncode != 0:
            print(n, "failed:", r.stderr[-800:])
            return 1
        fouts[n] = torch.load(out)
    for n in fnames[1:]:
        same = all(torch.equal(a, b) for k in fouts["shipped"] for a, b in zip(fouts["shipped"][k], fouts[n][k]))
        print(f"{n:>14s} [one-pass backward + register-resident forward, 4 cases]: bit-identical to shipped {same}")
        rc = rc or (0 if same else 1)
    print("scheduler fuzz:", "all variants bit-identical" if rc == 0 else "DIFFERENCES")
    return rc


if __name__ == "__main__":
    sys.exit(build() if sys.argv[1:] == ["build"] else run())
