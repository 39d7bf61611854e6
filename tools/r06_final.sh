#!/bin/bash
# Round-6 record batch on one GPU box: the driver's bench line, the rocprofv3 kernel table of the same command, the training step's
# kernel table, trace and HBM bytes, the three colour heads, PMC of the headline kernel.   gpurun -- 'bash tools/r06_final.sh'
# (the whole GPU suite is a separate call: profiles/r06/pytest_gpu.log)
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --no-traffic > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2>/dev/null)
find /tmp/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
for i in 1 2 3; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1; done > $O/train_step.json
python tools/train_bench.py --crop 128 --iters 10 2>/dev/null | tail -1 >> $O/train_step.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
find /tmp/prof_train -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_train.csv
timeout 600 python tools/train_hbm.py --out $O/train_hbm.json > /dev/null 2>&1
python tools/head_bench.py 5 > $O/head_bench.log 2>&1; cp gpurun_out/head_bench.json $O/head_bench.json
python tools/pmc_collect.py --kernel "render_ls_kernel" --out $O/pmc_render_ls_f16x.json -- python $PWD/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --no-traffic > /dev/null 2>&1
python -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('mfma_busy_frac'), d['roofline']['traffic'], d['train_step']['ms_per_step'], d['train_step_1m']['ms_per_step'])
for r in d['other_configs']:
    if r.get('dtype') == 'f16x': print(r['config'][:40], r['Msamples_s'], r['frac'])
"
cat $O/train_step.json | cut -c90-170
