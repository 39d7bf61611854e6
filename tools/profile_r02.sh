#!/bin/bash
# GPU-box profiling batch of round 2: bench line, rocprofv3 kernel stats and PMC passes of the same command.
cd "$(dirname "$0")/.."
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 5 --warmup 2 --engine reg --no-cpu-baseline > $O/bench_engine_reg.json 2> $O/bench_reg.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$O/bench_under_rocprof.json 2> $OLDPWD/$O/rocprof.err)
find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
python tools/pmc_collect.py --kernel "render_ls_kernel<1, 0>" --out $O/pmc_render_ls_bf16x3.json -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_x3.log 2>&1
python tools/pmc_collect.py --kernel "render_ls_kernel<0, 0>" --out $O/pmc_render_ls_bf16.json -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_bf16.log 2>&1
rm -rf $O/prof_bench
ls -la $O
cat $O/bench_default.json
python tools/ls_bench.py 3 > $O/ls_bench.log 2>&1
python tools/ls_trace.py run bf16 > $O/ls_trace_bf16.log 2>&1
python tools/ls_trace.py run bf16x3 > $O/ls_trace_bf16x3.log 2>&1
python tools/kernel_bench.py --json $O/kernel_bench.json > $O/kernel_bench.log 2>&1
python tools/train_bench.py > $O/train_bench.log 2>&1
