#!/bin/bash
# Round-5 PMC evidence (GPU box): attention kernels of precision mode 16f (LDS bank conflicts, MFMA busy) and the stage 3-4 Linear launches with the 16-bit
# weight shadows (wave wait cycles, MFMA busy).  Counters in separate rocprofv3 passes with --kernel-trace only (pool rules).
# usage: bash tools/collect_r05_pmc.sh <tag>
TAG=${1:-r05_z}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
LEOD_PRECISION=16f bash tools/pmc_kbench.sh 16 ${TAG}_attn "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" > /dev/null 2>&1
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $ROOT
F=$OUT/pmc_${TAG}_gemm.txt
{ echo "# tools/kbench_gemm.py 3,4 (stage 3-4 Linear launches), graph-timed, 16-bit weight shadows, precision mode 16f"; KBENCH_GRAPH=1 KBENCH_SHADOW=1 LEOD_PRECISION=16f python tools/kbench_gemm.py 3,4 20 2>/dev/null | grep -v "^sum"; } > $F
for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  D=$OUT/pmc_gemm; rm -rf $D
  KBENCH_SHADOW=1 LEOD_PRECISION=16f KBENCH_FILTER=ln_qkv,ln_fc1,fc2_lsres,dgrad_fc1,dgrad_qkv,dgrad_fc2 timeout 200 rocprofv3 --kernel-trace --pmc $C -d $D -o p -- python tools/kbench_gemm.py 3,4 3 > $D.log 2>&1
  python tools/pmc_summary.py $(find $D -name "*.db" | head -1) gemm_ >> $F
  rm -rf $D
done
rm -rf $OUT/pmc_${TAG}_attn_[12]
ls -la $OUT/pmc_${TAG}_attn.txt $F
