#!/bin/bash
# Round-4 evidence at HEAD (GPU box).  usage: bash tools/collect_r04.sh <tag>
set -u
TAG=${1:-r04_x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $ROOT
# 1. the driver's command (plans on) and the eager line of the same box
python bench.py --steps 100 --warmup 10 > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench_line_default.json
python bench.py --steps 100 --warmup 10 --no-plan --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_eager.json
# 2. N > 1 path on one rank: every RCCL collective issued (LEOD_FORCE_COLLECTIVES=1), plans with host callbacks vs eager vs no collectives
{
  echo "# bench.py --steps 60 --warmup 10, one MI355X, RCCL communicator of one rank: ms_per_step / event-frames/s / host enqueue ms"
  for v in "no_collectives_plans=" "force_collectives_plans=LEOD_FORCE_COLLECTIVES=1" "force_collectives_eager=LEOD_FORCE_COLLECTIVES=1 --no-plan" "no_collectives_eager=--no-plan"; do
    name=${v%%=*}; rest=${v#*=}; envs=""; flags=""
    for tok in $rest; do case $tok in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; esac; done
    line=$(env $envs MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline $flags 2>/dev/null | grep "^{\"metric" | tail -1)
    echo "$name $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], d["value"], c.get("host_enqueue_ms_per_step"), c.get("collective_backend"), (c.get("launch_plans") or {}).get("backward"))' 2>/dev/null || echo FAILED)"
  done
} > $OUT/${TAG}_rccl_force_collectives.txt 2>&1
# 3. kernel stats: eager two-stream / single-stream, plan-replayed
for MODE in two single plan; do
  D=$OUT/prof_$MODE; rm -rf $D
  case $MODE in two) E="LEOD_WGRAD_STREAM=1"; F="--no-plan";; single) E="LEOD_WGRAD_STREAM=0 LEOD_HEAD_STREAMS=0"; F="--no-plan";; plan) E="LEOD_PLAN=1"; F="";; esac
  env $E rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-second-dtype --no-roofline $F > $D.log 2>&1
  SUF="_$MODE"
  python tools/rocprof_summary.py $D $OUT/${TAG}_bench_bf16${SUF}_steps5_kernel_stats.csv > /dev/null 2>&1
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
cp $OUT/traffic/bf16.csv $OUT/${TAG}_hbm_traffic_pmc_bf16.csv; cp $OUT/traffic/f32.csv $OUT/${TAG}_hbm_traffic_pmc_f32.csv
cp $OUT/traffic/bf16.json $OUT/${TAG}_traffic_bf16.json; cp $OUT/traffic/f32.json $OUT/${TAG}_traffic_f32.json
python tools/kbench_mlp.py > $OUT/${TAG}_kbench_mlp.txt 2>&1
ls $OUT | grep ${TAG} | head -40
