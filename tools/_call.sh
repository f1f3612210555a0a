OUT=gpurun_out/r4s; mkdir -p $OUT
timeout 500 python -m pytest tests -q -x -m gpu -k "batch_image_by_scatter or ml20m_shape or g8_epoch or test_g1 or test_g5 or g7 or fast_paths or cmvae or g11" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
for rep in 1 2; do
for v in "scatter" "rewrite --opt gather_scatter=0" "sparse --first-layer sparse"; do
  set -- $v; name=$1; shift
  timeout 100 python bench.py $B "$@" > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
  echo "bench $name $rep rc=$? $(python -c "
import json
d=json.loads(open('$OUT/bench_${name}_$rep.json').read().strip().splitlines()[-1]); print('%.1f us/step; dW avg %.1f us; %s' % (d['ms_per_step']*1e3, d['roofline']['avg_us'], d['config']['first_layer']))" 2>&1 | tail -1)"
done
done
