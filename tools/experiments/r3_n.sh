#!/bin/bash
OUT=gpurun_out/${1:-r3n}
mkdir -p $OUT
for v in "" _s1 _s2; do
  timeout 300 build/native/test_gemm$v big > $OUT/gemm_big$v.log 2>&1; echo "gemm$v rc=$?"; grep -E "perf|FAIL|PASSED" $OUT/gemm_big$v.log
done
