cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fin3; mkdir -p $O; rm -rf $O/*
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench_default.txt 2>&1
tail -3 $O/pytest_gpu.txt; tail -1 $O/smoke.txt; tail -1 $O/bench_default.txt | cut -c1-300
