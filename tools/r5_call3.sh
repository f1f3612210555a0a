#!/bin/bash
OUT=gpurun_out/r5c3; mkdir -p $OUT
for v in "" "hop_values=0" "prefetch=0" "two_stream=0"; do
  for i in 1 2 3; do
    RTX_ENGINE_OPTS="$v" timeout 300 python -m pytest tests -m gpu -x -q -k "prefetched" > $OUT/pf_"$v"_$i.log 2>&1; echo "prefetch test [$v] run $i rc=$?"
  done
done
GPU_MAX_HW_QUEUES=8 timeout 1200 python tools/dp_debug.py > $OUT/dp_debug.log 2>&1; echo "dp_debug rc=$?"
grep -v "n_diff \[0, 0, 0, 0, 0, 0, 0, 0\]" $OUT/dp_debug.log | cut -c1-300 | head -60
grep -c "n_diff \[0, 0, 0, 0, 0, 0, 0, 0\]" $OUT/dp_debug.log
