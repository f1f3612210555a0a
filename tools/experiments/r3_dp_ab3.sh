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
export RTX_PROBE_INIT_PG=1
run emu1_pg --emulate-world 1 --replicated
run emu1_pg_prio0 --emulate-world 1 --replicated --opt side_low_prio=0
GPU_MAX_HW_QUEUES=2 run emu1_pg_q2 --emulate-world 1 --replicated
GPU_MAX_HW_QUEUES=8 run emu1_pg_q8 --emulate-world 1 --replicated
GPU_MAX_HW_QUEUES=16 run emu1_pg_q16 --emulate-world 1 --replicated
unset RTX_PROBE_INIT_PG
GPU_MAX_HW_QUEUES=8 run emu1_q8 --emulate-world 1 --replicated
run emu1_prio0 --emulate-world 1 --replicated --opt side_low_prio=0
