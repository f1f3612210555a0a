#!/bin/bash
OUT=gpurun_out/${1:-r3g}
mkdir -p $OUT
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler')))
except Exception as e: print('no line', e)
")"; }
run fused
run fused_prio0 --opt side_low_prio=0
run fused2
run fused_prio0_2 --opt side_low_prio=0
run emu8_prio0 --emulate-world 8 --opt side_low_prio=0
run emu8 --emulate-world 8
bash tools/prof_cmd.sh $1/emu8p --emulate-world 8 --opt side_low_prio=0
bash tools/prof_cmd.sh $1/emu1p --emulate-world 1 --replicated --opt side_low_prio=0
