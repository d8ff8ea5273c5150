#!/bin/bash
# Same-box ablation A/B (tools/ablate_bench.py): ms_per_step with whole kernel families turned into no-ops.  usage: tools/ablate_ab.sh OUT "<spec>" ...
OUT=$1; shift
mkdir -p $(dirname $OUT)
F="--steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-second-dtype"
for rep in 1 2; do
  for spec in "none" "$@"; do
    if [ "$spec" = none ]; then L=$(timeout 300 python bench.py $F 2>/dev/null | tail -1); else L=$(timeout 300 python tools/ablate_bench.py "$spec" $F 2>/dev/null | tail -1); fi
    echo "$rep  $spec  $(echo $L | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)" >> $OUT
  done
done
cat $OUT
