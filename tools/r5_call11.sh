#!/bin/bash
OUT=gpurun_out/r5c11; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "deferred_join or sequence_restart or prefetched or mailbox or stream_ordered or two_stream or g8 or ml20m_shape or reference_test or native_driver" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python tools/determinism_check.py 20108 600 200 500 12 4 > $OUT/det.log 2>&1; grep -v amdgpu $OUT/det.log | head -3
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras"
for i in 1 2 3; do
  timeout 200 python3 bench.py $B > $OUT/defer_$i.json 2> $OUT/defer_$i.err; echo "defer rc=$?"
  timeout 200 python3 bench.py $B --no-defer-join > $OUT/nodefer_$i.json 2> $OUT/nodefer_$i.err; echo "nodefer rc=$?"
done
bash tools/prof_cmd.sh r5c11/prof
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c11/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'frac', round(d['roofline']['frac'] or 0,3))
    except Exception as e: print(f, 'no line', e)
PY
head -36 gpurun_out/r5c11/prof/timeline.txt
