#!/bin/bash
# A/B of the training step: the round-4 snapshot under gpurun_ablate/old_r04 (package + library as committed at a01bcef) against the
# working tree, interleaved in ONE gpurun call (box-to-box spread is +-4 %: only same-box pairs mean anything).
#   bash tools/train_ab_old.sh [rounds=3]
R=${1:-3}
for i in $(seq $R); do
  (cd gpurun_ablate/old_r04 && python tools/train_bench.py --iters 30 | sed 's/^/old  /' | cut -c1-150)
  python tools/train_bench.py --iters 30 | sed 's/^/new  /' | cut -c1-150
done
