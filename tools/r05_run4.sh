O=gpurun_out/r05g; mkdir -p $O; rm -f $O/fwd_bench.log
timeout 900 python -m pytest tests/test_gpu_train_gemm.py -x -q -k "register_resident or packed_operands" 2>&1 | tail -2 | tee $O/pytest.log
for i1 in 0 38; do python tools/fwd_bench.py $i1 2>/dev/null | tee -a $O/fwd_bench.log; done
python tools/fwd_bench.py 0 sin 2>/dev/null | tee -a $O/fwd_bench.log
for v in 1 2 4 8; do NA_LIB_PATH=$PWD/gpurun_ablate/lib_var_tfw$v.so python tools/fwd_bench.py 0 2>/dev/null | sed "s/^/ablate$v /" | tee -a $O/fwd_bench.log; done
for i in 1 2; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c90-160; NA_TRAIN_FUSED_FWD=0 python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c90-160 | sed 's/^/fwd-off /'; done | tee $O/train_step.log
