#!/bin/bash
# rocprofv3 kernel trace of tools/bench_svae.py -> gpurun_out/$1/{svae_kernel_stats,svae_timeline}.txt   usage: prof_svae.sh OUT [bench args]
OUT=gpurun_out/$1; shift
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_sv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sv -o p -- python $R/tools/bench_svae.py "$@" > $R/$OUT/bench_svae_prof.log 2>&1
DB=$(find /tmp/prof_sv -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/svae_kernel_stats.txt
python $R/tools/rocprof_summary.py timeline $DB "k_sv_embed" 300 2 > $R/$OUT/svae_timeline.txt
cd $R
