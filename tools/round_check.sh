mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2i/pytest.log
timeout 300 python bench.py > gpurun_out/r2i/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2i/bench.log | cut -c1-400
timeout 200 python bench.py --no-defer --no-cpu-baseline --no-fp32-parity > gpurun_out/r2i/bench_nodefer.log 2>&1; tail -1 gpurun_out/r2i/bench_nodefer.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r2i
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-parity > $GRAFT_REPO_ROOT/gpurun_out/r2i/bench_prof.log 2>&1
DB=$(find /tmp/prof_r2i -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/r2i/kernel_stats.txt
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py timeline $DB > $GRAFT_REPO_ROOT/gpurun_out/r2i/timeline.txt
head -20 $GRAFT_REPO_ROOT/gpurun_out/r2i/kernel_stats.txt
