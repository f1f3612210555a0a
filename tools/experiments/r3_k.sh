#!/bin/bash
OUT=gpurun_out/${1:-r3k}
mkdir -p $OUT
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s conc=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler'), d['config'].get('second_stream_concurrent')))
except Exception as e: print('no line', e)
")"; }
run fused
run emu8 --emulate-world 8
run emu1_repl --emulate-world 1 --replicated
run dp1_native_rccl --force-dp --replicated
run dp1_native_rccl_shard --force-dp --sharded
RTX_PROBE_INIT_PG=1 run emu8_pg --emulate-world 8
bash tools/prof_cmd.sh $1/dp1 --force-dp --replicated
