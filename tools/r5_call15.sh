#!/bin/bash
OUT=gpurun_out/r5c15; mkdir -p $OUT
timeout 420 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc/* $OUT/ 2>/dev/null
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --pmc-json gpurun_out/pmc/pmc_summary.json > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver cmd rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c15/bench_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver cmd %.1f us | bracket %.2f empty %.2f corrected %.2f frac %.3f traffic %s' % (d['ms_per_step']*1e3, r['avg_us_event_bracket'], r['event_bracket_overhead_us'], r['avg_us'], r['frac'], r['traffic']))
PY
