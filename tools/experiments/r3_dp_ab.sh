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
run emu8_shard --emulate-world 8
run emu8_shard_dw3 --emulate-world 8 --opt dw_cfg=3
run emu8_repl --emulate-world 8 --replicated
run emu1_repl --emulate-world 1 --replicated
run dp1_native_torch --force-dp --replicated --dp-transport torch
run dp1_native_rccl --force-dp --replicated --dp-transport rccl
RTX_PROBE_IDLE_COMM=1 run dp1_python_idlecomm --force-dp --replicated --dp-engine python
run dp1_python --force-dp --replicated --dp-engine python
