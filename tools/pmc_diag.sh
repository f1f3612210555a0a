#!/bin/bash
# Diagnostic counters for the step's kernels (round 6): L2 hit/miss, TCP->TCC requests, wave stall split, LDS conflicts.  One rocprofv3
# --pmc pass per counter group over a short bench.py run (counters + --kernel-trace only, as gpurun requires).  Kernels are serialised
# under --pmc: each is alone on the device.   usage: tools/pmc_diag.sh OUTDIR [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmcdiag}; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
while read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1)); tag=g$i
  rm -rf /tmp/pmcd_$tag
  timeout 240 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmcd_$tag -o p -- python $R/bench.py --steps 12 --warmup 3 --windows 1 --preheat-seconds 0 --no-cpu-baseline --no-fp32-parity --no-extras "$@" > /tmp/pd_$tag.log 2>&1
  echo "pass $tag [$pass] rc=$?"
  DB=$(find /tmp/pmcd_$tag -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocprof_summary.py pmc $DB > $OUT/diag_$tag.txt; else tail -5 /tmp/pd_$tag.log; fi
done <<PASSES
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
PASSES
python $R/tools/rocprof_summary.py diag $OUT > $OUT/diag_table.txt 2>&1
cat $OUT/diag_table.txt | head -60
