#!/bin/bash
OUT=gpurun_out/${1:-r3i}
mkdir -p $OUT
timeout 600 build/native/test_engine > $OUT/engine.log 2>&1; echo "engine rc=$?"; grep -E "FAIL|TESTS|PASSED" $OUT/engine.log | tail -3
python tools/adam_probe.py 2>&1 | grep "us$" > $OUT/adam_probe.txt; cat $OUT/adam_probe.txt
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler')))
except Exception as e: print('no line', e)
")"; }
run fused
run emu8 --emulate-world 8
run emu8_dw3 --emulate-world 8 --opt dw_cfg=3
run emu1_repl --emulate-world 1 --replicated
run dp1_native_rccl --force-dp --replicated
run dp1_native_rccl_shard --force-dp --sharded
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_path_world1 or dp_world2 or g2_train or g4_dae or config0 or random_arch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
