cd $GRAFT_REPO_ROOT
O=gpurun_out/sw; mkdir -p $O; rm -rf $O/*
for BL in 768 1024 1536 2048; do
echo "=== WGRADW_BLOCKS=$BL" >> $O/k.txt
LEOD_PRECISION=bf16 LEOD_WGRADW_BLOCKS=$BL timeout 300 python tools/kbench.py wgrad 2>&1 | grep -i "wgrad" | awk '{s+=$3; printf "%s ", $3} END {print " sum", s}' >> $O/k.txt
done
cat $O/k.txt
