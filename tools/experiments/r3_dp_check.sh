#!/bin/bash
# data-parallel checks of round 3 on one GPU box: the two DP test scripts, then bench lines of every schedule
OUT=gpurun_out/${1:-r3dp}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "dp_path_world1 or dp_world2 or ml20m_shape_b500 or c_abi_rccl" > $OUT/pytest_dp.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_dp.log | cut -c1-300
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler')))
except Exception as e: print('no line', e)
")"; }
run fused
run dp1_native_repl --force-dp --replicated
run dp1_native_shard --force-dp --sharded
run dp1_python_repl --force-dp --replicated --dp-engine python
run emu8_shard --emulate-world 8
run emu8_repl --emulate-world 8 --replicated
run emu2_shard --emulate-world 2
run emu4_shard --emulate-world 4
