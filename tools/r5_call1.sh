#!/bin/bash
# round 5, call 1: the new tests (self-launching bench, stream-ordered ranks, loss mailbox, ABI 7) + the driver's exact command
# with and without the pre-heat, alternating.
OUT=gpurun_out/r5c1; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "mailbox or stream_ordered or starts_its_own or dp_world or g2_train or rccl or native_driver or abi" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver cmd rc=$?"
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --preheat-seconds 0 > $OUT/driver_cmd_nopreheat_$i.json 2> $OUT/driver_cmd_nopreheat_$i.err; echo "no preheat rc=$?"
done
timeout 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'frac', round(d['roofline']['frac'] or 0,3), 'api', d.get('train_batch_api',{}).get('ms_per_step'), 'fp32', d.get('fp32_parity',{}).get('ms_per_step'), 'preheat', d['preheat']['steps'])
    except Exception as e: print(f, 'no line', e)
PY
