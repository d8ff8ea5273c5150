#!/bin/bash
# PMC passes over a narrow kbench filter.  usage: tools/pmc_kbench.sh <kbench-filter> <outname> "<CTRS pass 1>" "<CTRS pass 2>" ...
# (counters in separate rocprofv3 runs, no tracing besides --kernel-trace, as the pool rules require)
set -u
FILTER=$1; OUT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
rm -f $ROOT/gpurun_out/pmc_${OUT}.txt
for CTRS in "$@"; do
  i=$((i+1))
  D=$ROOT/gpurun_out/pmc_${OUT}_$i
  rm -rf $D
  KBENCH_EAGER=1 timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $CTRS -d $D -o p -- python $ROOT/tools/kbench.py $FILTER > $D.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  python $ROOT/tools/pmc_summary.py $DB >> $ROOT/gpurun_out/pmc_${OUT}.txt
done
