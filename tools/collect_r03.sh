#!/bin/bash
# Round-3 evidence at HEAD (GPU box): bench line, kernel stats (two-stream and single-stream), main-stream gaps, PMC passes for the
# attention and GEMM families, HBM traffic of the roofline kernel.  usage: bash tools/collect_r03.sh <tag>
set -u
TAG=${1:-r03_x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $ROOT
python bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench_line_default.json
for MODE in two single; do
  D=$OUT/prof_$MODE; rm -rf $D
  WS=1; [ $MODE = single ] && WS=0
  LEOD_WGRAD_STREAM=$WS rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype --no-roofline > $D.log 2>&1
  SUF=""; [ $MODE = single ] && SUF="_single_stream"
  python tools/rocprof_summary.py $D $OUT/${TAG}_bench_bf16${SUF}_steps5_kernel_stats.csv > /dev/null 2>&1
  [ $MODE = two ] && python tools/stream_gaps.py $D > $OUT/${TAG}_main_stream_gaps_under_rocprof.txt 2>&1
  [ $MODE = two ] && python tools/stream_tail.py $D > $OUT/${TAG}_stream_tail_under_rocprof.txt 2>&1
  python tools/last_step_kernels.py $D $OUT/${TAG}_last_step${SUF}_kernels.csv > /dev/null 2>&1
  rm -rf $D
done
# PMC: attention (bf16 tiles) and the Linear GEMM family
LEOD_PRECISION=bf16 bash tools/pmc_kbench.sh 16 ${TAG}_attn "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" > /dev/null 2>&1
LEOD_PRECISION=bf16 KBENCH_FILTER=ln_qkv,fc2_lsres,dgrad_fc1 bash tools/pmc_kbench.sh __none__ ${TAG}_dummy "SQ_WAVES" > /dev/null 2>&1
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  D=$OUT/pmc_gemm; rm -rf $D
  LEOD_PRECISION=bf16 KBENCH_FILTER=ln_qkv,ln_fc1,fc2_lsres,dgrad_fc1,dgrad_qkv rocprofv3 --kernel-trace --pmc $C -d $D -o p -- python tools/kbench_gemm.py 3,4 3 > $D.log 2>&1
  python tools/pmc_summary.py $(find $D -name "*.db" | head -1) gemm_ >> $OUT/pmc_${TAG}_gemm.txt
  rm -rf $D
done
bash tools/pmc_bench_traffic.sh > /dev/null 2>&1
rm -rf $OUT/pmc_${TAG}_attn_1 $OUT/pmc_${TAG}_attn_2 $OUT/pmc_${TAG}_dummy*
ls $OUT | head -50
