#!/bin/bash
OUT=gpurun_out/${1:-r3p}
mkdir -p $OUT
B="--steps 30 --no-cpu-baseline --no-fp32-parity --no-extras --workload netflix"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism']))
except Exception as e: print('no line', e)
")"; tail -2 $OUT/$name.err | cut -c1-200; }
run netflix_old --opt big_batch_tiles=0
run netflix_new
run netflix_new_dw0 --opt dw_cfg=0
run netflix_old_dw3 --opt big_batch_tiles=0 --opt dw_cfg=3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
