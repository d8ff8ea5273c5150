cd ${GRAFT_REPO_ROOT:-/root/repo}
{
  echo "# bench.py --steps 60 --warmup 10, one MI355X, RCCL communicator of one rank (LEOD_FORCE_COLLECTIVES=1: parameter broadcast, 47 SyncBatchNorm exchanges, gradient buckets are issued on RCCL): ms_per_step / event-frames/s / host enqueue ms of one isolated step / backend / backward plan"
  echo "# plans_head_eager = planned backbone + eager head (EagerHeadGate, LEOD_PLAN_HEAD_EAGER=1), plans_head_captured = LEOD_PLAN_HEAD_EAGER=0 (every collective a plan segment boundary)"
  for rep in 1 2; do
  for v in "no_collectives_plans=" "force_collectives_plans_head_eager=LEOD_FORCE_COLLECTIVES=1 LEOD_PLAN_HEAD_EAGER=1" "force_collectives_plans_head_captured=LEOD_FORCE_COLLECTIVES=1" "force_collectives_eager=LEOD_FORCE_COLLECTIVES=1 --no-plan" "no_collectives_eager=--no-plan"; do
    name=${v%%=*}; rest=${v#*=}; envs=""; flags=""
    for tok in $rest; do case $tok in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; esac; done
    line=$(env $envs MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline $flags 2>/dev/null | grep '^{"metric' | tail -1)
    echo "$name $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], d["value"], c.get("host_enqueue_ms_per_step"), c.get("collective_backend"), (c.get("launch_plans") or {}).get("backward"))' 2>/dev/null || echo FAILED)"
  done; done
} > gpurun_out/r05_p_rccl_force_collectives.txt 2>&1
cat gpurun_out/r05_p_rccl_force_collectives.txt
