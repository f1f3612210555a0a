#!/bin/bash
OUT=gpurun_out/r5c12; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "evaluate or topk or metrics or g6 or g8 or ndcg or reference_test" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python tools/bench_eval.py 10000 500 > $OUT/bench_eval.json 2> $OUT/bench_eval.err; echo "eval rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c12/bench_eval.json').read().strip().splitlines()[-1])
print('bf16 B=500 %.2f M users/s | fp32 %.2f M | bf16 B=2000 %.2f M | host loop %.1f K' % (d['bf16']['users_per_s_device']/1e6, d['fp32']['users_per_s_device']/1e6, d['bf16_batch_2000']['users_per_s_device']/1e6, d['bf16']['users_per_s_host_metrics']/1e3))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profev
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profev -o p -- python $R/tools/bench_eval.py 10000 500 > $R/$OUT/prof_eval.log 2>&1
DB=$(find /tmp/profev -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $R/$OUT/eval_kernel_stats.txt
cd $R; head -14 $OUT/eval_kernel_stats.txt
