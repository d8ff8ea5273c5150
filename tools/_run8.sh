cd $GRAFT_REPO_ROOT
O=gpurun_out/st; mkdir -p $O; rm -rf $O/*
for i in 1 2; do for S in 1 0; do
LEOD_STEM_WGRAD_SIDE=$S timeout 600 python bench.py --no-cpu-baseline --no-second-dtype --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STEM_SIDE=$S', d['value'], d['ms_per_step'])" >> $O/bench.txt
done; done
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_module_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O/bench.txt
cat $O/bench.txt
