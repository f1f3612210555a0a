#!/bin/bash
OUT=gpurun_out/r5c14; mkdir -p $OUT
bash tools/prof_cmd.sh r5c14/prof
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c14/prof/bench_prof.log').read().strip().splitlines()[-1])
r=d['roofline']; print('under rocprof: events bracket %.2f  empty %.2f  corrected %.2f  frac %.3f' % (r['avg_us_event_bracket'], r['event_bracket_overhead_us'], r['avg_us'], r['frac']))
PY
head -6 gpurun_out/r5c14/prof/kernel_stats.txt
for i in 1 2; do timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras > $OUT/drv_$i.json 2>$OUT/drv_$i.err; python - <<PY
import json
d=json.loads(open('$OUT/drv_$i.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver cmd %.1f us | bracket %.2f empty %.2f corrected %.2f frac %.3f' % (d['ms_per_step']*1e3, r['avg_us_event_bracket'], r['event_bracket_overhead_us'], r['avg_us'], r['frac']))
PY
done
