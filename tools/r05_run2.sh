O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_gemm.py tests/test_gpu_backward.py tests/test_gpu_train.py -x -q 2>&1 | tail -5 | tee $O/pytest.log
for o in 256 65; do python tools/bwd_bench.py $o 2>/dev/null | tee -a $O/bwd_bench.log; done
for i in 1 2 3; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c1-200; done | tee $O/train_step.json
for i in 1 2; do NA_TRAIN_FUSED_BWD=0 python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/unfused /'; done | tee -a $O/train_step.json
