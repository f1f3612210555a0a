#!/bin/bash
# One GPU-box pass of the round's checks: pytest -m gpu, the counters (three --pmc passes), bench.py (default line citing them),
# a rocprofv3 kernel trace of bench.py turned into the per-kernel summary and a two-step timeline, the data-parallel lines.
# Outputs: gpurun_out/$1/.   usage: tools/gpu.sh --timeout 2700 -- 'bash tools/round_check.sh r3final'
OUT=gpurun_out/${1:-check}
mkdir -p $OUT
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log
fi
timeout 400 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc/* $OUT/ 2>/dev/null
timeout 400 python bench.py --pmc-json gpurun_out/pmc/pmc_summary.json > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -1 $OUT/bench_default.json | cut -c1-330
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler')))
except Exception as e: print('no line', e)
")"; }
run fused
run one_stream --opt two_stream=0
run generic_kernels --opt sparse_in=0 --opt small_fwd=0 --opt small_bwd=0
run dp1_rccl_replicated --force-dp --replicated
run dp1_rccl_sharded --force-dp --sharded
run emu8_sharded --emulate-world 8
run emu4_sharded --emulate-world 4
run emu2_sharded --emulate-world 2
bash tools/prof_cmd.sh $1/prof_fused
bash tools/prof_cmd.sh $1/prof_emu8 --emulate-world 8
