#!/bin/bash
# Round 4: raw phase stamps of the persistent grid and of one workgroup per tile; repeated-launch test; A/B against round 3's library
OUT=$PWD/gpurun_out/${1:-r4h}
mkdir -p $OUT
T=$PWD/build/native/test_gemm
DWX_DUMP=$OUT/stamps_pers timeout 40 $T dwx 0 stamps 1 > $OUT/dwx_stamps_pers.txt 2>&1; echo "stamps pers rc=$?"
DWX_DUMP=$OUT/stamps_tile timeout 40 $T dwx 0 stamps 0 > $OUT/dwx_stamps_tile.txt 2>&1; echo "stamps tile rc=$?"
timeout 150 $T dw > $OUT/test_gemm_dw.txt 2>&1; echo "test_gemm dw rc=$?"; grep -E "persistent|PASSED|FAILED" $OUT/test_gemm_dw.txt | tail -4
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
for rep in 1 2; do
for v in "r3" "new --opt dw_persistent=0" "new_logits32 --opt dw_persistent=0 --opt logits16=0"; do
  set -- $v; name=$1; shift
  if [ $name = r3 ]; then export RTX_LIB_PATH=$PWD/build/r3lib/librectorch_hip.so; else unset RTX_LIB_PATH; fi
  timeout 100 python bench.py $B "$@" > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
  echo "bench $name $rep rc=$? $(python -c "
import json
d=json.loads(open('$OUT/bench_${name}_$rep.json').read().strip().splitlines()[-1]); print('%.1f us/step; dW avg %.1f us; windows %s' % (d['ms_per_step']*1e3, d['roofline']['avg_us'], ['%.1f' % (w*1e6/d['steps']) for w in d['windows']['seconds']]))" 2>&1 | tail -1)"
done
done
unset RTX_LIB_PATH
bash tools/prof_cmd.sh ${1:-r4h}/prof_new --opt dw_persistent=0 > /dev/null 2>&1
