#!/bin/bash
# rocprofv3 kernel trace of tools/bench_eval.py -> gpurun_out/$1/{eval_kernel_stats,eval_timeline}.txt   usage: prof_eval.sh OUT [users] [batch]
OUT=gpurun_out/$1; shift
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ev
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ev -o p -- python $R/tools/bench_eval.py "$@" > $R/$OUT/bench_eval_prof.log 2>&1
DB=$(find /tmp/prof_ev -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/eval_kernel_stats.txt
python $R/tools/rocprof_summary.py timeline $DB "k_gather_scatter<unsigned short>" 60 2 > $R/$OUT/eval_timeline.txt
cd $R
