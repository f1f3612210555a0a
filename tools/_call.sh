OUT=gpurun_out/r4v; mkdir -p $OUT
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
for rep in 1 2 3; do
for v in "events" "values --opt hop_values=1"; do
  set -- $v; name=$1; shift
  timeout 100 python bench.py $B "$@" > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
  echo "bench $name $rep rc=$? $(python -c "
import json
d=json.loads(open('$OUT/bench_${name}_$rep.json').read().strip().splitlines()[-1]); print('%.1f us/step; dW avg %.1f us; loss %.4f' % (d['ms_per_step']*1e3, d['roofline']['avg_us'], d['mean_loss']))" 2>&1 | tail -1)"
done
done
bash tools/prof_cmd.sh r4v/prof_values --opt hop_values=1 > /dev/null 2>&1; head -16 gpurun_out/r4v/prof_values/timeline.txt
