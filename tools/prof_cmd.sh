#!/bin/bash
# rocprofv3 kernel trace of one bench.py command line -> gpurun_out/$1/{kernel_stats,timeline}.txt   usage: prof_cmd.sh OUT <bench args>
OUT=gpurun_out/$1; shift
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_rc
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rc -o p -- python $R/bench.py --steps 60 --warmup 10 --windows 1 --no-cpu-baseline --no-fp32-parity --no-extras "$@" > $R/$OUT/bench_prof.log 2>&1
DB=$(find /tmp/prof_rc -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/kernel_stats.txt
python $R/tools/rocprof_summary.py timeline $DB > $R/$OUT/timeline.txt
cd $R
tail -1 $OUT/bench_prof.log | cut -c1-160
