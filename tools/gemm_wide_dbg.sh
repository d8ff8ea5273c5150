#!/bin/bash
# Phase decomposition of gemm_wide_bf16_kernel at the stage-3 / stage-4 shapes (graph-timed launches, 16-bit weight shadows as in the step):
# LEOD_WIDE_DBG bit 0 skips the epilogue's row bodies (global stores), bit 1 the MFMAs (and fragment reads), bit 2 the global loads after
# the prologue, bit 3 the staging step (conversions + LDS writes), bit 4 the epilogue's LDS exchange, bit 5 the per-tile row bookkeeping (LayerNorm statistics
# rows always those of the first tile).  usage (GPU box): bash tools/gemm_wide_dbg.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1 2 4 7 15 23 39 63; do echo "=== LEOD_WIDE_DBG=$d ==="; KBENCH_GRAPH=1 KBENCH_SHADOW=1 LEOD_WIDE_DBG=$d LEOD_PRECISION=16f python tools/kbench_gemm.py 3,4 20 2>&1 | grep -v "amdgpu.ids"; done > gpurun_out/gemm_wide_dbg.txt 2>&1
cat gpurun_out/gemm_wide_dbg.txt
