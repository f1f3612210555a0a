#!/bin/bash
OUT=gpurun_out/r5c16; mkdir -p $OUT
timeout 300 build/native/test_gemm dw > $OUT/dw_cases.log 2>&1; echo "dw cases rc=$?"; tail -2 $OUT/dw_cases.log; grep -c "FAIL" $OUT/dw_cases.log
timeout 300 build/native/test_gemm dwk > $OUT/dwk.log 2>&1; cat $OUT/dwk.log
B="--workload netflix --steps 40 --warmup 10 --no-cpu-baseline --no-fp32-parity --no-extras"
for cfg in 0 4 3; do timeout 200 python3 bench.py $B --opt dw_cfg=$cfg > $OUT/netflix_cfg$cfg.json 2> $OUT/netflix_cfg$cfg.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/netflix_cfg$cfg.json').read().strip().splitlines()[-1]); print('netflix B=4096 dw_cfg=$cfg: %.1f us/step  dw kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline']['avg_us']))
except Exception as e: print('cfg $cfg no line', e)
PY
done
