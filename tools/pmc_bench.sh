#!/bin/bash
# HBM traffic per kernel of the flagship step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# bench.py run; summaries land in gpurun_out/pmc/ (copy the ones to keep into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /tmp/p_$c.log 2>&1
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py pmc $DB > $R/gpurun_out/pmc/bench_$c.txt
done
head -24 $R/gpurun_out/pmc/bench_FETCH_SIZE.txt
