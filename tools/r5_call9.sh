#!/bin/bash
OUT=gpurun_out/r5c9; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "fp32 or g1_ or g3 or g4 or random_arch or config0 or g8 or g9 or smoke" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="--gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-fp32-parity --no-extras --numerics fp32"
for i in 1 2; do
  timeout 200 python3 bench.py $B > $OUT/f32_overlap_$i.json 2> $OUT/f32_overlap_$i.err; echo "rc=$?"
  timeout 200 python3 bench.py $B --opt f32_adam_overlap=0 > $OUT/f32_nooverlap_$i.json 2> $OUT/f32_nooverlap_$i.err; echo "rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c9/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'TF frac', round(d['step_roofline']['algorithmic_flops_per_step']/(d['ms_per_step']*1e-3)/1e12/157.3,3), d['roofline']['avg_us'])
    except Exception as e: print(f, 'no line', e)
PY
