mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest.log
timeout 600 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
for i in 1 2 3; do timeout 300 python tools/train_bench.py --iters 30 2>/dev/null | tail -1; done > gpurun_out/r05a/train_step.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05a/kernel_stats_train.csv
tail -3 gpurun_out/r05a/pytest.log; cat gpurun_out/r05a/train_step.json
