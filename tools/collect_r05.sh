#!/bin/bash
# Round-5 evidence at HEAD (GPU box).  usage: bash tools/collect_r05.sh <tag>
set -u
TAG=${1:-r05_z}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $ROOT
# 1. the driver's command (plans on) and the eager line of the same box
python bench.py --steps 100 --warmup 10 > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench_line_default.json
python bench.py --steps 100 --warmup 10 --no-plan --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_eager.json
# 2. the labelled-frame count drawn per step (plan cache under a data-dependent B'): long warm-up so that every count has been seen
python bench.py --vary-labels 12:40 --steps 100 --warmup 150 --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_vary_labels.json
python bench.py --vary-labels 12:40 --steps 100 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_vary_labels_cold.json
python bench.py --dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_mode_bf16.json
# 3. kernel stats: eager two-stream / single-stream, plan-replayed
for MODE in two single plan; do
  D=$OUT/prof_$MODE; rm -rf $D
  case $MODE in two) E="LEOD_WGRAD_STREAM=1"; F="--no-plan";; single) E="LEOD_WGRAD_STREAM=0 LEOD_HEAD_STREAMS=0"; F="--no-plan";; plan) E="LEOD_PLAN=1"; F="";; esac
  env $E rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-second-dtype --no-roofline $F > $D.log 2>&1
  SUF="_$MODE"
  python tools/rocprof_summary.py $D $OUT/${TAG}_bench_16f${SUF}_steps5_kernel_stats.csv > /dev/null 2>&1
  python tools/last_step_kernels.py $D $OUT/${TAG}_last_step${SUF}_kernels.csv > /dev/null 2>&1
  [ $MODE != single ] && python tools/stream_gaps.py $D 8 > $OUT/${TAG}_stream_gaps${SUF}_under_rocprof.txt 2>&1
  [ $MODE != single ] && python tools/stream_tail.py $D > $OUT/${TAG}_stream_tail${SUF}_under_rocprof.txt 2>&1
  rm -rf $D
done
# 4. the other configs from bench.py
python bench.py --pseudo --batch 16 --seq-len 21 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_pseudo.json
python bench.py --dataset gen4 --full-res --size base --seq-len 11 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_1mpx.json
python bench.py --dataset gen4 --size base --seq-len 5 --batch 12 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_gen4ds2.json
# 5. HBM traffic of the roofline family (PMC, family markers)
bash tools/pmc_bench_traffic.sh > /dev/null 2>&1
for DT in 16f bf16 f32; do cp $OUT/traffic/$DT.csv $OUT/${TAG}_hbm_traffic_pmc_$DT.csv; cp $OUT/traffic/$DT.json $OUT/${TAG}_traffic_$DT.json; done
python tools/kbench_mlp.py > $OUT/${TAG}_kbench_mlp.txt 2>&1
ls $OUT | grep ${TAG} | head -40
