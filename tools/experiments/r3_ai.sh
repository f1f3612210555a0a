#!/bin/bash
OUT=gpurun_out/${1:-r3ai}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_sv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sv -o p -- python $R/tools/bench_svae.py --steps 600 --warmup 50 --cpu-seconds 0 > $R/$OUT/bench.log 2>&1
DB=$(find /tmp/prof_sv -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/kernel_stats.txt
cd $R
head -30 $OUT/kernel_stats.txt
