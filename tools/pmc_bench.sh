#!/bin/bash
# Counters of the flagship step, per kernel: THREE separate rocprofv3 --pmc passes over a short bench.py run (the TCC block
# cannot hold FETCH_SIZE and WRITE_SIZE at once; counters and --kernel-trace/--stats only, as gpurun requires):
#   FETCH_SIZE, WRITE_SIZE                                   -> HBM bytes per launch (gfx950 correction: FETCH x 2)
#   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -> MFMA utilisation per kernel
# Summaries: gpurun_out/pmc/bench_<pass>.txt + gpurun_out/pmc/pmc_summary.json (bench.py --pmc-json reads the latter).
# Copy what is to be kept into profiles/.     usage: tools/pmc_bench.sh [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 30 --warmup 5 --windows 1 --preheat-seconds 0 --no-cpu-baseline --no-fp32-parity --no-extras "$@" > /tmp/p_$tag.log 2>&1
  DB=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py pmc $DB > $R/gpurun_out/pmc/bench_$tag.txt
done
RTX_PMC_BENCH_ARGS="$*" python $R/tools/rocprof_summary.py pmcjson $R/gpurun_out/pmc > $R/gpurun_out/pmc/pmc_summary.json
python $R/tools/rocprof_summary.py mfma $R/gpurun_out/pmc/bench_SQ_VALU_MFMA_BUSY_CYCLES.txt > $R/gpurun_out/pmc/mfma_util.txt
cat $R/gpurun_out/pmc/mfma_util.txt
