#!/bin/bash
OUT=gpurun_out/r5c8; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "fp32 or g1_ or g3 or g4 or random_arch or config0 or native_driver" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="--gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-fp32-parity --no-extras --numerics fp32"
for i in 1 2; do
  timeout 200 python3 bench.py $B > $OUT/f32_new_$i.json 2> $OUT/f32_new_$i.err; echo "rc=$?"
  timeout 200 python3 bench.py $B --opt f32_tail_split=0 > $OUT/f32_notail_$i.json 2> $OUT/f32_notail_$i.err; echo "rc=$?"
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof32
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof32 -o p -- python $R/bench.py --steps 40 --warmup 10 --windows 1 --no-cpu-baseline --no-fp32-parity --no-extras --numerics fp32 > $R/$OUT/prof32.log 2>&1
DB=$(find /tmp/prof32 -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/fp32_kernel_stats.txt
python $R/tools/rocprof_summary.py timeline $DB > $R/$OUT/fp32_timeline.txt
cd $R
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'TF frac', round(d['step_roofline']['algorithmic_flops_per_step']/(d['ms_per_step']*1e-3)/1e12/157.3,3))
    except Exception as e: print(f, 'no line', e)
PY
head -16 $OUT/fp32_kernel_stats.txt; head -30 $OUT/fp32_timeline.txt
