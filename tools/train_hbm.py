#!/usr/bin/env python3
"""HBM bytes of ONE training step (tools/train_bench.py), all kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes, no trace domains -- MI355X_MICROARCH.md "HBM"), summed per kernel name and divided by the steps of the run.
Bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (the guide's gfx950 correction: FETCH_SIZE reports half of a wide coalesced read).
    python tools/train_hbm.py --out gpurun_out/train_hbm.json [--iters 10]"""
import argparse, csv, glob, json, os, shutil, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, cmd, work):
    out = os.path.join(work, counter)
    r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + cmd, cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError(f"{counter}: no counter_collection.csv (rc={r.returncode})\n{r.stderr[-2000:]}")
    per, calls = {}, {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0][:80]
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
            calls[k] = calls.get(k, 0) + 1
    return per, calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    steps = a.iters + 3  # train_bench.py runs 3 untimed steps first
    cmd = [sys.executable, os.path.join(REPO, "tools", "train_bench.py"), "--iters", str(a.iters)]
    work = tempfile.mkdtemp(prefix="trainhbm_")
    try:
        fetch, calls = one_pass("FETCH_SIZE", cmd, work)
        write, _ = one_pass("WRITE_SIZE", cmd, work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        b = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 / steps
        rows.append({"kernel": k, "launches_per_step": round(calls.get(k, 0) / steps, 2), "MB_per_step": round(b / 1e6, 1)})
    total = sum(r["MB_per_step"] for r in rows)
    res = {"what": "HBM bytes per PlainNeRF(view) training step of 262 144 samples (tools/train_bench.py), (2 x FETCH_SIZE + WRITE_SIZE) KiB",
           "steps_profiled": steps, "GB_per_step": round(total / 1e3, 2),
           # which forward the profiled step ran (bench.py quotes this file only for the same one)
           "train_forward": "layers" if os.environ.get("NA_TRAIN_LS") == "0" else "ls", "kernels": rows}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({"GB_per_step": res["GB_per_step"], "top": rows[:8]}, indent=1))


if __name__ == "__main__":
    main()
