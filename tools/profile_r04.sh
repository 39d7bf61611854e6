#!/bin/bash
# GPU-box profiling batch of round 4: bench line, rocprofv3 kernel stats and PMC passes of the same command, per-phase trace.
cd "$(dirname "$0")/.."
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 5 --warmup 2"
$B > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OLDPWD/$O/bench_under_rocprof.json 2> $OLDPWD/$O/rocprof.err)
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
rm -rf $O/prof_bench
for spec in "3:f16x" "1:bf16x3" "0:bf16" "2:f16"; do
  id=${spec%%:*}; name=${spec##*:}
  python tools/pmc_collect.py --kernel "render_ls_kernel<$id, 0>" --out $O/pmc_render_ls_$name.json -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc_$name.log 2>&1
done
python - <<'PY'
import json, os
O = "gpurun_out/r04"
out = {}
for name in ("f16x", "bf16x3", "bf16", "f16"):
    try:
        d = json.load(open(f"{O}/pmc_render_ls_{name}.json"))
        out[f"ls/{name}"] = d["derived"]["hbm_bytes_per_launch_corrected"]
    except Exception as e:
        print(name, e)
json.dump(out, open(f"{O}/hbm_traffic.json", "w"), indent=1)
print(out)
PY
python tools/ls_trace.py run f16x > $O/ls_trace_f16x.log 2>&1
python tools/ls_trace.py run bf16x3 > $O/ls_trace_bf16x3.log 2>&1
ls -la $O
head -c 600 $O/bench_default.json
bash tools/power_probe.sh f16x > $O/power_probe_f16x.log 2>&1
# hardware probes behind the round's hazard finding + the reproducibility probe of the mip renderer
for p in mfma_use_hazard trans_use_hazard pk_after_trans_mfma; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/hw/$p.hip -o /tmp/$p 2>/dev/null && /tmp/$p > $O/$p.log 2>&1
done
python tools/mip_det_probe.py 40 > $O/mip_det_probe.log 2>&1
python tools/cfg_bench.py 1 3 4 5m 5 --iters 5 > $O/cfg_bench.log 2>&1
