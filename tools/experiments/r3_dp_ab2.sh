#!/bin/bash
OUT=gpurun_out/${1:-r3ab}
mkdir -p $OUT
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler')))
except Exception as e: print('no line', e)
")"; }
run fused
run emu1_repl --emulate-world 1 --replicated
RTX_PROBE_INIT_PG=1 run emu1_repl_pg --emulate-world 1 --replicated
run dp1_native_rccl --force-dp --replicated
run dp1_native_rccl_1stream --force-dp --replicated --opt two_stream=0
run dp1_native_rccl_prio --force-dp --replicated --opt side_low_prio=0
NCCL_DEBUG=INFO run dp1_native_rccl_dbg --force-dp --replicated --steps 20
grep -i "channel\|cu\b\|mask\|stream" $OUT/dp1_native_rccl_dbg.err | head -20
