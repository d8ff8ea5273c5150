#!/bin/bash
# rocprofv3 kernel trace of bench.py (5 steps) in the given mode -> stream gap / tail summaries.  usage: tools/trace_step.sh <tag> [bench flags / ENV=..]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; TAG=$1; shift
cd $ROOT
envs=""; flags=""
for tok in "$@"; do case $tok in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
D=$OUT/trace_$TAG; rm -rf $D
env $envs rocprofv3 --kernel-trace --stats -d $D -o b -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-second-dtype --no-roofline $flags > $D.log 2>&1
python tools/stream_gaps.py $D 8 > $OUT/${TAG}_gaps.txt 2>&1
python tools/stream_tail.py $D > $OUT/${TAG}_tail.txt 2>&1
python tools/last_step_kernels.py $D $OUT/${TAG}_last_step_kernels.csv > /dev/null 2>&1
python tools/rocprof_summary.py $D $OUT/${TAG}_kernel_stats.csv > /dev/null 2>&1
rm -rf $D
head -12 $OUT/${TAG}_gaps.txt
