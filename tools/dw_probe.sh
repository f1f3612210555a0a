#!/bin/bash
# Where the fused weight-gradient + Adam kernel (dw_adam.hip) spends its time -> profiles/r4_dw_phase_probe.txt, r4_pmc_dw_adam_stalls.txt
#   (1) phase ablations + per-workgroup phase stamps (tests/native/test_gemm dwx; raw stamps: DWX_DUMP + tools/dw_stamps.py)
#   (2) stall / occupancy / L2 counters of the unmodified kernel (separate --pmc passes, counters + kernel trace only)
# usage: tools/gpu.sh --timeout 600 -- 'bash tools/dw_probe.sh OUT'
OUT=$PWD/gpurun_out/${1:-r4a}
mkdir -p $OUT
R=$PWD
T=$R/build/native/test_gemm
DWX_DUMP=$OUT/stamps timeout 100 $T dwx 0 > $OUT/dwx_cfg0.txt 2>&1; echo "dwx0 rc=$?"
timeout 60 $T dwx 1 > $OUT/dwx_cfg1.txt 2>&1; echo "dwx1 rc=$?"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 150 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmc_$i -o p -- $T dwx 0 base > $OUT/pmc_pass$i.log 2>&1
  DB=$(find /tmp/pmc_$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py pmc $DB > $OUT/pmc_pass$i.txt
  echo "pmc pass $i ($pass): rc=$? $(wc -l < $OUT/pmc_pass$i.txt 2>/dev/null) lines"
done
