O=gpurun_out/r05c; mkdir -p $O; rm -f $O/ablate.log
timeout 600 python -m pytest tests/test_gpu_train_gemm.py -x -q -k fused_backward 2>&1 | tail -3
for o in 256 65 3; do python tools/bwd_bench.py $o 2>/dev/null | sed 's/^/shipped /' | tee -a $O/ablate.log; done
python tools/bwd_bench.py 256 sin 2>/dev/null | sed 's/^/shipped /' | tee -a $O/ablate.log
for v in ${VARIANTS:-1 4 8 5}; do NA_LIB_PATH=$PWD/gpurun_ablate/lib_var_tbw$v.so python tools/bwd_bench.py 256 2>/dev/null | sed "s/^/ablate$v /" | tee -a $O/ablate.log; done
python tools/pmc_collect.py --kernel "lsbw::kernel" --out $O/pmc_bwd.json -- python $PWD/tools/bwd_bench.py 256 > /dev/null 2>$O/pmc.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r05c/pmc_bwd.json'))
print(d['derived']); print(d['per_launch']['lds']); print(d['per_launch']['sq'])
P
