#!/bin/bash
# Round-6 evidence at HEAD (GPU box).  usage: bash tools/collect_r06.sh <tag>
set -u
TAG=${1:-r06_z}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
# 0. the GPU suite as the driver runs it
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15) > $OUT/pytest_gpu.txt
# 1. the driver's command (plans on) and the eager lines of the same box
python bench.py --steps 100 --warmup 10 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench_line_default.json
python bench.py --steps 100 --warmup 10 --no-plan --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/bench_line_eager.json
python bench.py --steps 20 --warmup 5 --no-plan --single-stream --no-second-dtype --no-cpu-baseline --dump-calls $OUT/calls_single_stream_step.txt 2>/dev/null | tail -1 > $OUT/bench_line_eager_single_stream.json
python bench.py --vary-labels 12:40 --steps 100 --warmup 150 --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/bench_line_vary_labels.json
python bench.py --dtype bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 > $OUT/bench_line_mode_bf16.json
# 2. kernel stats: eager single-stream, plan-replayed
for MODE in single plan; do
  D=$OUT/prof_$MODE; rm -rf $D
  case $MODE in single) F="--no-plan --single-stream";; plan) F="";; esac
  rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-second-dtype --no-roofline $F > $D.log 2>&1
  python tools/rocprof_summary.py $D $OUT/bench_16f_${MODE}_steps5_kernel_stats.csv > /dev/null 2>&1
  rm -rf $D
done
# 3. the other configs
python bench.py --pseudo --batch 16 --seq-len 21 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line_pseudo.json
python bench.py --dataset gen4 --full-res --size base --seq-len 11 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_1mpx.json
python bench.py --dataset gen4 --size base --seq-len 5 --batch 12 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_gen4ds2.json
# 3b. 1 Mpx: kernel stats (single stream) + per-launch table; one rank with every collective of the N > 1 path issued
D=$OUT/prof_1mpx; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --dataset gen4 --full-res --size base --seq-len 11 --batch 2 --steps 5 --warmup 4 --no-cpu-baseline --no-second-dtype --no-roofline --no-plan --single-stream > $D.log 2>&1
python tools/rocprof_summary.py $D $OUT/bench_1mpx_single_steps5_kernel_stats.csv > /dev/null 2>&1
rm -rf $D
python bench.py --dataset gen4 --full-res --size base --seq-len 11 --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-dtype --dump-calls $OUT/calls_1mpx_single_stream_step.txt 2>/dev/null | tail -1 > /dev/null
(export LEOD_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533; python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_line_forced_collectives.json)
# 4. HBM traffic of the roofline family (PMC, family markers; separate FETCH_SIZE / WRITE_SIZE passes)
bash tools/pmc_bench_traffic.sh > /dev/null 2>&1
for DT in 16f bf16 f32; do cp $ROOT/gpurun_out/traffic/$DT.csv $OUT/hbm_traffic_pmc_$DT.csv; cp $ROOT/gpurun_out/traffic/$DT.json $OUT/traffic_$DT.json; done
(cd tools; python kbench_wgrad_small.py) > $OUT/kbench_wgrad_small.txt 2>&1
ls $OUT
