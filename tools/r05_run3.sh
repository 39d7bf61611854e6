O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c1-200; done | tee $O/train_step.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
find /tmp/prof_train -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_train.csv
timeout 600 python tools/train_hbm.py --out $O/train_hbm.json 2>&1 | tail -30
