#!/bin/bash
OUT=gpurun_out/${1:-r3af}
mkdir -p $OUT
timeout 600 build/native/test_engine > $OUT/engine.log 2>&1; echo "engine rc=$?"; grep -E "FAIL|PASSED" $OUT/engine.log | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g1_ or g2_ or g3_ or g4_ or g7_ or edge or ml20m_shape or config0 or random_arch or cmvae or custom_op" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism']))
except Exception as e: print('no line', e)
")"; }
run new_a
RTX_LIB_PATH=$PWD/build/librectorch_hip_prev.so run prev_a
run new_b
RTX_LIB_PATH=$PWD/build/librectorch_hip_prev.so run prev_b
bash tools/prof_cmd.sh $1/prof
grep -E "dlogits|gemm_nt" $OUT/prof/kernel_stats.txt
