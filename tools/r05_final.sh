#!/bin/bash
# Round-5 record batch on one GPU box: the whole GPU suite, the driver's bench line, the training step's kernel table and HBM
# bytes, PMC of the one-pass backward and of the headline kernel.   gpurun -- 'bash tools/r05_final.sh'
O=gpurun_out/r05_final; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
for i in 1 2 3; do python tools/train_bench.py --iters 30 2>/dev/null | tail -1; done > $O/train_step.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --no-other-configs --no-cpu-baseline --no-traffic > /dev/null 2>&1)
find /tmp/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv
timeout 600 python tools/train_hbm.py --out $O/train_hbm.json > /dev/null 2>&1
python tools/pmc_collect.py --kernel "lsbw::kernel<true, 1, true, true>" --out $O/pmc_train_bwd.json -- python $PWD/tools/train_bench.py --iters 10 > /dev/null 2>&1
python tools/pmc_collect.py --kernel "lsfw::kernel<1, 0, false>" --out $O/pmc_train_fwd.json -- python $PWD/tools/train_bench.py --iters 10 > /dev/null 2>&1
python tools/pmc_collect.py --kernel "render_ls_kernel" --out $O/pmc_render_ls_f16x.json -- python $PWD/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --no-traffic > /dev/null 2>&1
cat $O/train_step.json | cut -c90-170
