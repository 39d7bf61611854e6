#!/usr/bin/env python3
"""The ordered kernel launches of ONE training step out of rocprofv3's kernel trace of tools/train_bench.py (profiles/r06/train_trace*.txt):

    python tools/train_trace.py gpurun_out/r06_final/kernel_trace_train.csv > profiles/r06/train_trace.txt

One step = the launches after the last-but-one optimiser launch up to the last one; columns: start (us from the step's first launch),
duration (us), kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_many" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f}  {r['Kernel_Name'][:110]}")
print(f"kernel sum {tot:.1f} us, span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches "
      "(one step of tools/train_bench.py under rocprofv3 --kernel-trace; start us, duration us, kernel)")
