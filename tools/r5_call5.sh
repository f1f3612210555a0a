#!/bin/bash
OUT=gpurun_out/r5c5; mkdir -p $OUT
timeout 900 python tools/determinism_check.py > $OUT/det_small.log 2>&1; echo "rc=$?"; cat $OUT/det_small.log | grep -v amdgpu.ids | head -3
timeout 900 python tools/determinism_check.py 20108 600 200 500 12 6 > $OUT/det_ml20m.log 2>&1; echo "rc=$?"; cat $OUT/det_ml20m.log | grep -v amdgpu.ids | head -3
for i in 1 2 3; do timeout 300 python -m pytest tests -m gpu -x -q -k "prefetched" > $OUT/pf_$i.log 2>&1; echo "prefetch test run $i rc=$?"; done
GPU_MAX_HW_QUEUES=8 timeout 1200 python tools/dp_debug.py > $OUT/dp_debug.log 2>&1; echo "dp_debug rc=$?"
grep -v "n_diff \[0, 0, 0, 0, 0, 0, 0, 0\]" $OUT/dp_debug.log | cut -c1-300 | head -20
grep -c "n_diff \[0, 0, 0, 0, 0, 0, 0, 0\]" $OUT/dp_debug.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream_ordered" > $OUT/so.log 2>&1; echo "stream-ordered test rc=$?"; tail -3 $OUT/so.log
