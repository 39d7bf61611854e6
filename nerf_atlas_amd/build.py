"""Build libnerf_atlas_amd.so (HIP, gfx950) in-tree with hipcc.  No CPU fallback exists: if this library is
missing the package raises on first use.

    python -m nerf_atlas_amd.build [--force] [--jobs N]
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libnerf_atlas_amd.so")
ARCH = "gfx950"

# (source, extra flags, object suffix)
UNITS = [
    ("basic_ops.hip", [], ""),
    ("linear_f32.hip", [], ""),
    ("backward.hip", [], ""),
    ("train_gemm.hip", [], ""),
    ("march.hip", [], ""),
    ("mlp_fused.hip", [], ""),
    ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=0"], "_bf16"),
    ("mlp_fwd_inst.hip", ["-DNA_PREC_INST=1"], "_bf16x3"),
    ("render_fused.hip", ["-DNA_PREC_INST=0"], "_bf16"),
    ("render_fused.hip", ["-DNA_PREC_INST=1"], "_bf16x3"),
]
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"] + os.environ.get("NA_EXTRA_HIPCC_FLAGS", "").split()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the HIP extension")
    return exe


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(unit):
    src, extra, suffix = unit
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + suffix + ".o")
    cmd = [hipcc()] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}{suffix}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, jobs: int = 0, verbose: bool = True) -> str:
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u[0]))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or min(len(units), os.cpu_count() or 4)
    if verbose:
        print(f"[nerf_atlas_amd] compiling {len(units)} units for {ARCH} with {jobs} jobs", file=sys.stderr)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(_compile, units))
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=0)
    a = ap.parse_args()
    print(build(a.force, a.jobs))
