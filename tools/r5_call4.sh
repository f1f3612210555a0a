#!/bin/bash
OUT=gpurun_out/r5c4; mkdir -p $OUT
timeout 900 python tools/determinism_check.py > $OUT/det_small.log 2>&1; echo "rc=$?"; cat $OUT/det_small.log | grep -v amdgpu.ids
timeout 900 python tools/determinism_check.py 20108 600 200 500 12 6 > $OUT/det_ml20m.log 2>&1; echo "rc=$?"; cat $OUT/det_ml20m.log | grep -v amdgpu.ids
