#!/bin/bash
# same-call A/B of the training step (tools/train_bench.py): layer-by-layer forward (NA_TRAIN_LS=0) vs the one-launch forward, two sizes
for i in 1 2; do
NA_TRAIN_LS=0 python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c80-200
NA_TRAIN_LS=1 python tools/train_bench.py --iters 30 2>/dev/null | tail -1 | cut -c80-200
done
NA_TRAIN_LS=0 python tools/train_bench.py --crop 128 --iters 10 2>/dev/null | tail -1 | cut -c80-200
NA_TRAIN_LS=1 python tools/train_bench.py --crop 128 --iters 10 2>/dev/null | tail -1 | cut -c80-200
