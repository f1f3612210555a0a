#!/bin/bash
OUT=gpurun_out/r5c6; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_full.log
for i in 1 2 3; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras > $OUT/drv_$i.json 2> $OUT/drv_$i.err; echo "driver cmd rc=$?"
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras --no-prefetch > $OUT/drv_noprefetch_$i.json 2> $OUT/drv_noprefetch_$i.err; echo "no prefetch rc=$?"
done
timeout 200 python3 bench.py --no-cpu-baseline --no-fp32-parity > $OUT/bench_200.json 2> $OUT/bench_200.err
bash tools/prof_cmd.sh r5c6/prof
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c6/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'frac', round(d['roofline']['frac'] or 0,3), 'api', d.get('train_batch_api',{}).get('ms_per_step'), d['config'].get('batch_prefetch'))
    except Exception as e: print(f, 'no line', e)
PY
head -40 gpurun_out/r5c6/prof/timeline.txt
