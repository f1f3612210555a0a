#!/bin/bash
OUT=gpurun_out/r5c2; mkdir -p $OUT
GPU_MAX_HW_QUEUES=8 timeout 900 python tools/dp_debug.py > $OUT/dp_debug.log 2>&1; echo "dp_debug rc=$?"; tail -5 $OUT/dp_debug.log
timeout 600 python -m pytest tests -m gpu -x -q -k "prefetched or mailbox" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver cmd rc=$?"
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-prefetch > $OUT/driver_cmd_noprefetch_$i.json 2> $OUT/driver_cmd_noprefetch_$i.err; echo "no prefetch rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'frac', round(d['roofline']['frac'] or 0,3), 'api', d.get('train_batch_api',{}).get('ms_per_step'), d['config'].get('batch_prefetch'))
    except Exception as e: print(f, 'no line', e)
PY
