#!/bin/bash
# the round's closing pass on one GPU box: the whole GPU suite, smoke(), the driver's own bench command, then the counter passes of HEAD
# (tools/round_check.sh holds the variants / profiles).  usage: tools/gpu.sh --timeout 3600 -- 'bash tools/closing_pass.sh'
OUT=gpurun_out/closing; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/parity_tests.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/parity_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver cmd rc=$?"
timeout 420 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc/* $OUT/ 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/closing/bench_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver cmd %.1f us %.0f users/s | bracket %.2f empty %.2f corrected %.2f frac %.3f traffic %s | step frac %.3f | api %.1f | fp32 %.1f' % (d['ms_per_step']*1e3, d['value'], r['avg_us_event_bracket'], r['event_bracket_overhead_us'], r['avg_us'], r['frac'], r['traffic'], d['step_roofline']['frac_of_hbm_peak'], d['train_batch_api']['ms_per_step']*1e3, d['fp32_parity']['ms_per_step']*1e3))
PY
