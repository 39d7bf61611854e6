#!/bin/bash
# GPU-box driver for the w8g prototype ablations (tools/proto/ls_mlp_w8g.hip)
cd "$(dirname "$0")"
mkdir -p ../../gpurun_out
out=../../gpurun_out/proto_w8g.log
: > $out
for cfg in "0" "2" "4" "8" "10" "0 -DNBLK=3" "0 -DPF=8" "0 -DNBLK=2"; do
  set -- $cfg
  echo "=== w8g ablate=$1 extra=${@:2}" >> $out
  python ls_mlp.py build w8g $1 ${@:2} >> $out 2>&1 && timeout 120 python ls_mlp.py run 12 >> $out 2>&1
done
for v in w8 w8s; do
  echo "=== $v (round-1 reference)" >> $out
  python ls_mlp.py build $v 0 >> $out 2>&1 && timeout 120 python ls_mlp.py run 12 >> $out 2>&1
done
cat $out
