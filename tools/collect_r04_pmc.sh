#!/bin/bash
# Round-4 PMC evidence (GPU box): attention kernels on one-head vs two-head workgroups (LDS bank conflicts, MFMA busy), stage 3-4 Linear launches
# with the bf16 weight shadow (wave wait cycles, MFMA busy).  Counters in separate rocprofv3 passes with --kernel-trace only (pool rules).
# usage: bash tools/collect_r04_pmc.sh <tag>
TAG=${1:-r04_zz}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
LEOD_PRECISION=bf16 LEOD_ATTN_HG1=3 bash tools/pmc_kbench.sh 16 ${TAG}_attn_hg1 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" > /dev/null 2>&1
LEOD_PRECISION=bf16 LEOD_ATTN_HG1=0 bash tools/pmc_kbench.sh 16 ${TAG}_attn_hg2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" > /dev/null 2>&1
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $ROOT
for SH in 1 0; do
  { echo "# tools/kbench_gemm.py 3,4 (stage 3-4 Linear launches), KBENCH_SHADOW=$SH"; KBENCH_SHADOW=$SH LEOD_PRECISION=bf16 KBENCH_FILTER=ln_qkv,ln_fc1,fc2_lsres,dgrad_fc1,dgrad_qkv,dgrad_fc2,proj python tools/kbench_gemm.py 3,4 20 2>/dev/null | grep -v "^sum"; } >> $OUT/pmc_${TAG}_gemm_shadow$SH.txt
  for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    D=$OUT/pmc_gemm; rm -rf $D
    KBENCH_SHADOW=$SH LEOD_PRECISION=bf16 KBENCH_FILTER=ln_qkv,ln_fc1,fc2_lsres,dgrad_fc1,dgrad_qkv,dgrad_fc2 timeout 200 rocprofv3 --kernel-trace --pmc $C -d $D -o p -- python tools/kbench_gemm.py 3,4 3 > $D.log 2>&1
    python tools/pmc_summary.py $(find $D -name "*.db" | head -1) gemm_ >> $OUT/pmc_${TAG}_gemm_shadow$SH.txt
    rm -rf $D
  done
done
rm -rf $OUT/pmc_${TAG}_attn_hg1_[12] $OUT/pmc_${TAG}_attn_hg2_[12]
ls -la gpurun_out/pmc_${TAG}_attn_hg1.txt gpurun_out/pmc_${TAG}_attn_hg2.txt
