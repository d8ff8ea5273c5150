#!/bin/bash
# One rank, every collective of the N > 1 path issued on RCCL (LEOD_FORCE_COLLECTIVES=1): SyncBatchNorm exchanges as host callbacks between plan
# segments (default) vs captured into the launch plan by ProcessGroupNCCL (LEOD_PLAN_CAPTURE_COLLECTIVES=1) vs no collectives.  GPU box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "# bench.py --steps 40 --warmup 10, one MI355X: ms_per_step / host enqueue ms / backward plan / forward plan"
for rep in 1 2; do for v in "callbacks=LEOD_FORCE_COLLECTIVES=1" "captured=LEOD_FORCE_COLLECTIVES=1 LEOD_PLAN_CAPTURE_COLLECTIVES=1" "none="; do
  name=${v%%=*}; rest=${v#*=}
  line=$(env $rest MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline 2>/dev/null | grep '^{"metric' | tail -1)
  echo "$name $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["ms_per_step"], c.get("host_enqueue_ms_per_step"), (c.get("launch_plans") or {}).get("backward"), (c.get("launch_plans") or {}).get("forward"))' 2>/dev/null || echo FAILED)"
done; done
} > gpurun_out/r05_q_captured_collectives.txt 2>&1
cat gpurun_out/r05_q_captured_collectives.txt
