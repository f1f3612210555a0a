#!/bin/bash
OUT=gpurun_out/${1:-r3o}
mkdir -p $OUT
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism']))
except Exception as e: print('no line', e)
")"; }
run fused_a
RTX_LIB_PATH=$PWD/build/librectorch_hip_s1.so run fused_s1_a
run fused_b
RTX_LIB_PATH=$PWD/build/librectorch_hip_s1.so run fused_s1_b
run netflix --workload netflix --steps 30
RTX_LIB_PATH=$PWD/build/librectorch_hip_s1.so run netflix_s1 --workload netflix --steps 30
bash tools/prof_cmd.sh $1/netflix --workload netflix --steps 20
head -20 $OUT/netflix/kernel_stats.txt
