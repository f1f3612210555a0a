#!/bin/bash
OUT=gpurun_out/r5dlr; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "bf16 or g8 or ml20m_shape or random_arch or config3 or trained_model or prefetched or deferred or sequence_restart or fast_paths or g11 or cond" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 200 python tools/determinism_check.py 20108 600 200 500 12 3 2>&1 | grep -v amdgpu | head -3
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras"
for i in 1 2 3; do
  timeout 200 python3 bench.py $B > $OUT/row_$i.json 2> $OUT/row_$i.err; echo "row rc=$?"
  RTX_DLOGITS_ROW=0 timeout 200 python3 bench.py $B > $OUT/chunk_$i.json 2> $OUT/chunk_$i.err; echo "chunk rc=$?"
done
bash tools/prof_cmd.sh r5dlr/prof
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5dlr/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], '%.1f us' % (d['ms_per_step']*1e3), [round(x*1e6/d['steps'],1) for x in d['windows']['seconds']], 'loss', round(d['mean_loss'],4))
    except Exception as e: print(f, 'no line', e)
PY
grep -E "dlogits|gemm_nt<unsigned short, 1" gpurun_out/r5dlr/prof/kernel_stats.txt
