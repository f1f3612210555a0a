#!/bin/bash
# Round 4: dynamic persistent grid (per-XCD tile counters) -- kernel A/B, correctness, in-step A/B (alternating), timeline.
# Every command has a tight timeout of its own: a hung kernel must not eat the GPU budget.
OUT=$PWD/gpurun_out/${1:-r4c}
mkdir -p $OUT
T=$PWD/build/native/test_gemm
timeout 60 $T dwx 0 quick > $OUT/dwx_quick.txt 2>&1; rc=$?; echo "dwx quick rc=$rc"
[ $rc -ne 0 ] && { echo "persistent kernel broken: stop"; exit 1; }
timeout 120 $T dw > $OUT/test_gemm_dw.txt 2>&1; echo "test_gemm dw rc=$?"; tail -1 $OUT/test_gemm_dw.txt
echo skip test_engine
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
for rep in 1 2; do
for v in "default" "nopers --opt dw_persistent=0" "onestream --opt two_stream=0"; do
  set -- $v; name=$1; shift
  timeout 100 python bench.py $B "$@" > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
  echo "bench $name $rep rc=$? $(python -c "
import json
d=json.loads(open('$OUT/bench_${name}_$rep.json').read().strip().splitlines()[-1]); print('%.1f us/step; dW avg %.1f us; windows %s' % (d['ms_per_step']*1e3, d['roofline']['avg_us'], ['%.1f' % (w*1e6/d['steps']) for w in d['windows']['seconds']]))" 2>&1 | tail -1)"
done
done
