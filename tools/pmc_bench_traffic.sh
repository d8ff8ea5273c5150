#!/bin/bash
# HBM traffic of the roofline kernel family (Linear weight gradients: wgrad_wide_bf16_kernel + wgrad_wide_reduce_kernel / wgradw_kernel<.., XRows>) from rocprofv3 PMC passes over bench.py, both precision modes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE runs with --kernel-trace only (pool rule), summarised by tools/roofline_traffic.py into
# gpurun_out/traffic/{16f,bf16,f32}.{json,csv}.  usage (GPU box): bash tools/pmc_bench_traffic.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for DT in 16f bf16 f32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/${DT}_$C; rm -rf $D
    LEOD_FAMILY_MARKERS=1 timeout 600 rocprofv3 --kernel-trace --pmc $C -d $D -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --dtype $DT \
        --no-cpu-baseline --no-second-dtype --no-roofline --no-plan --single-stream > $D.log 2>&1
  done
  python $ROOT/tools/roofline_traffic.py $OUT/${DT}_FETCH_SIZE $OUT/${DT}_WRITE_SIZE $OUT/$DT.json $OUT/$DT.csv
  rm -rf $OUT/${DT}_FETCH_SIZE $OUT/${DT}_WRITE_SIZE
done
