O=gpurun_out/r05g; mkdir -p $O; rm -f $O/fwd_bench.log
timeout 900 python -m pytest tests/test_gpu_train_gemm.py -x -q -k "register_resident or packed_operands or whole_network" 2>&1 | tail -2 | tee $O/pytest.log
for i in 1 2; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c90-160; NA_TRAIN_FUSED_FWD=0 python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c90-160 | sed 's/^/fwd-off /'; done | tee $O/train_step.log
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
grep "lsfw\|lsnt" $O/kernel_stats_train.csv | cut -d, -f1-4 | cut -c1-120
