#!/bin/bash
# One GPU-box pass of the round's checks: native tests (+ perf variants), pytest -m gpu, bench.py (default line), and a
# rocprofv3 kernel trace of bench.py turned into the per-kernel summary and a two-step timeline.  Outputs: gpurun_out/$1/.
OUT=gpurun_out/${1:-check}
mkdir -p $OUT
if [ "$2" != "quick" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log
fi
timeout 100 build/native/test_spmm > $OUT/spmm.log 2>&1; echo "spmm rc=$?"; tail -3 $OUT/spmm.log | cut -c1-200
timeout 500 build/native/test_engine perf > $OUT/engine.log 2>&1; echo "engine rc=$?"; grep -E "FAIL|TESTS" $OUT/engine.log; grep -A1 "^\[perf\]" $OUT/engine.log | grep -v "^--"
timeout 300 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | cut -c1-330
timeout 200 python bench.py --opt two_stream=0 --no-cpu-baseline --no-fp32-parity > $OUT/bench_ab.log 2>&1; tail -1 $OUT/bench_ab.log | cut -c1-330
timeout 200 python bench.py --opt sparse_in=0 --opt small_fwd=0 --opt small_bwd=0 --no-cpu-baseline --no-fp32-parity > $OUT/bench_generic.log 2>&1; tail -1 $OUT/bench_generic.log | cut -c1-330
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_rc
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rc -o p -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-parity > $R/$OUT/bench_prof.log 2>&1
DB=$(find /tmp/prof_rc -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/kernel_stats.txt
python $R/tools/rocprof_summary.py timeline $DB > $R/$OUT/timeline.txt
head -16 $R/$OUT/kernel_stats.txt
cd $R
timeout 300 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
