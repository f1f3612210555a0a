OUT=gpurun_out/r4x; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 200 python tools/bench_eval.py 10000 500 > $OUT/bench_eval.json 2> $OUT/bench_eval.err; echo "eval rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_eval.json').read().strip().splitlines()[-1]); print(d['fp32']['users_per_s_device'], d['bf16']['users_per_s_device'], d['bf16_batch_2000'])"
