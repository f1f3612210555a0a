#!/bin/bash
OUT=gpurun_out/${1:-r3ag}
mkdir -p $OUT
B="--steps 50 --no-cpu-baseline --no-fp32-parity --no-extras --workload netflix"
run() { name=$1; shift; timeout 400 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  loss %.3f' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['mean_loss']))
except Exception as e: print('no line', e)
")"; tail -2 $OUT/$name.err | cut -c1-200; }
run nflx_fused_b512 --batch 512
run nflx_dp1_b512 --batch 512 --force-dp --sharded
run nflx_emu8 --emulate-world 8
run nflx_emu8_weak --emulate-world 8 --scaling weak --batch 512
