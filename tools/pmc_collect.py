#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for one kernel of a command, in SEPARATE passes (the TCC counters do not fit one pass
and counter collection must not be mixed with trace domains: MI355X_MICROARCH.md "HBM", "rocprofv3"), and aggregate
them per launch.  Runs on the GPU box:

    python tools/pmc_collect.py --kernel render_plain_view_kernel --out gpurun_out/pmc.json -- \\
        python bench.py --steps 3 --warmup 1 --no-cpu-baseline

Derived values: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x CUs x 4 SIMDs); HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB with
the gfx950 correction of the guide (FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE uncalibrated).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

PASSES = {
    "sq": ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
           "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"],
    "lds": ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS",
            "GRBM_GUI_ACTIVE"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}


def run_pass(name, counters, cmd, kernel, workdir):
    out = os.path.join(workdir, name)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", out, "--"] + cmd
    r = subprocess.run(full, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError(f"pass {name}: no counter_collection.csv (rc={r.returncode})\n{r.stderr[-2000:]}")
    per_counter, launches, dispatch = {}, set(), {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            launches.add(row.get("Dispatch_Id"))
            per_counter[row["Counter_Name"]] = per_counter.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if not dispatch:
                dispatch = {k: row.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                                    "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in row}
    n = max(len(launches), 1)
    return {k: v / n for k, v in per_counter.items()}, n, dispatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--extra-pass", action="append", default=[], help="NAME:CTR1,CTR2,... one more counter pass (raw values only)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    cmd = [os.path.abspath(c) if (c.endswith(".py") and os.path.exists(c)) else c for c in cmd]
    work = tempfile.mkdtemp(prefix="pmc_")
    res = {"kernel": a.kernel, "command": " ".join(cmd), "per_launch": {}}
    try:
        passes = dict(PASSES)
        for spec in a.extra_pass:
            nm, ctrs = spec.split(":", 1)
            passes[nm] = ctrs.split(",")
        for name, counters in passes.items():
            vals, n, dispatch = run_pass(name, counters, cmd, a.kernel, work)
            res["per_launch"][name] = vals
            res["launches_profiled"] = n
            if dispatch:
                res["dispatch"] = dispatch
    finally:
        shutil.rmtree(work, ignore_errors=True)
    sq, lds = res["per_launch"]["sq"], res["per_launch"]["lds"]
    fetch = res["per_launch"]["fetch"].get("FETCH_SIZE", 0.0)
    write = res["per_launch"]["write"].get("WRITE_SIZE", 0.0)
    gui = lds.get("GRBM_GUI_ACTIVE", 0.0)
    d = {}
    if gui:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs of all CUs
        cycles = gui / 8.0
        d["gpu_cycles_per_launch"] = cycles
        d["mfma_busy_frac"] = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cycles * a.cus * 4)  # 4 SIMDs per CU
        d["mfma_insts_per_cu_cycle"] = sq.get("SQ_INSTS_MFMA", 0.0) / (cycles * a.cus)
    if lds.get("SQ_ACTIVE_INST_LDS"):
        d["lds_bank_conflict_frac"] = lds.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds["SQ_ACTIVE_INST_LDS"]
    d["hbm_bytes_per_launch_corrected"] = (2.0 * fetch + write) * 1024.0
    d["fetch_KiB_raw"], d["write_KiB_raw"] = fetch, write
    res["derived"] = d
    os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["derived"], indent=1))


if __name__ == "__main__":
    sys.exit(main())
