#!/bin/bash
# Round 4, call 2: persistent grid + hardware sqrt/rcp in the fused weight-gradient + Adam kernel (A/B inside one process),
# its phase stamps, the native correctness cases, the engine parity driver, and the bench line (half-precision logits in).
OUT=$PWD/gpurun_out/${1:-r4b}
mkdir -p $OUT
T=$PWD/build/native/test_gemm
timeout 200 $T dwx 0 quick > $OUT/dwx_quick.txt 2>&1; echo "dwx quick rc=$?"
timeout 300 $T dw > $OUT/test_gemm_dw.txt 2>&1; echo "test_gemm dw rc=$?"; tail -1 $OUT/test_gemm_dw.txt
timeout 600 $T > $OUT/test_gemm_all.txt 2>&1; echo "test_gemm rc=$?"; grep -E "FAIL|PASSED|FAILED|logits16" $OUT/test_gemm_all.txt | head -20
timeout 600 $PWD/build/native/test_engine > $OUT/test_engine.txt 2>&1; echo "test_engine rc=$?"; tail -3 $OUT/test_engine.txt
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
for v in "fused" "logits32 --opt logits16=0" "fused2"; do
  set -- $v; name=$1; shift
  timeout 300 python bench.py $B "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$? $(tail -1 $OUT/bench_$name.json | cut -c1-240)"
done
