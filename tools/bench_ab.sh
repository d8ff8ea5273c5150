#!/bin/bash
# Same-box A/B of bench.py variants: usage tools/bench_ab.sh "<name>=<env and flags>" ...   e.g.  "eager=--no-plan" "plan4=LEOD_PLAN_LANES=4"
# Each variant runs twice (interleaved) with --steps 60 --warmup 10, no CPU / second-dtype / roofline legs; prints ms_per_step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for v in "$@"; do
    name=${v%%=*}; rest=${v#*=}
    envs=""; flags=""
    for tok in $rest; do case $tok in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
    line=$(env $envs python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-second-dtype --no-roofline $flags 2>/dev/null | tail -1)
    echo "$name rep$rep $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"].get("host_enqueue_ms_per_step"))' 2>/dev/null || echo FAILED)"
  done
done
