#!/bin/bash
# GPU-box profiling batch of round 3's training path (profiles/r03/README.md): step time, rocprofv3 kernel table, the three GEMMs
# per shape, PMC passes of the forward / input-gradient GEMM, their in-kernel traces, the hash scatter, the alignment probe.
# Before it, here (hipcc cross-compiles):  python tools/ls_variant.py build-unit train_gemm.hip tgtrace -DTGL_TRACE=1
#                                          hipcc --offload-arch=gfx950 -O3 tools/hw/unaligned_probe.hip -o gpurun_ablate/unaligned_probe
cd "$(dirname "$0")/.."
O=gpurun_out/r03_train; mkdir -p $O
export TMPDIR=/tmp
python tools/train_bench.py 2>/dev/null | tail -1 > $O/train_step.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $OLDPWD/tools/train_bench.py > /dev/null 2>&1)
find /tmp/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
python tools/gemm_check.py --time > $O/train_gemm_ls.log 2>&1
NA_TRAIN_GEMM=tiled python tools/gemm_check.py --time > $O/train_gemm_tiled.log 2>&1
python tools/pmc_collect.py --kernel "lsnt::kernel<0, false, false>" --out $O/pmc_train_fwd.json --extra-pass "ic:SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES,SQ_IFETCH" -- python tools/gemm_variant_run.py fwd shipped > /dev/null 2>&1
python tools/pmc_collect.py --kernel "lsnt::kernel<1, true, false>" --out $O/pmc_train_dgrad.json -- python tools/gemm_variant_run.py dgrad shipped > /dev/null 2>&1
python tools/pmc_collect.py --kernel "lstn::kernel<true, true>" --out $O/pmc_train_wgrad.json -- python tools/gemm_variant_run.py wgrad shipped > /dev/null 2>&1
[ -f gpurun_ablate/lib_var_tgtrace.so ] && for m in fwd dgrad; do python tools/gemm_variant_run.py $m tgtrace; done > $O/train_gemm_trace.log 2>&1
python tools/hash_bwd_run.py shipped > $O/hash_backward.log 2>&1
[ -x gpurun_ablate/unaligned_probe ] && ./gpurun_ablate/unaligned_probe > $O/unaligned_probe.log 2>&1
ls -la $O; cat $O/train_step.json
