OUT=$PWD/gpurun_out/r4z; mkdir -p $OUT
R=$PWD
timeout 300 python -m pytest tests -q -x -m gpu -k "evaluate_device or topk or g6 or g8_epoch or ndcg or validfunc or host_logic" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ev
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_ev -o p -- python $R/tools/bench_eval.py 10000 500 > $OUT/bench_eval_prof.log 2>&1
DB=$(find /tmp/prof_ev -name "*.db" | head -1)
python $R/tools/rocprof_summary.py stats $DB > $OUT/eval_kernel_stats.txt
head -8 $OUT/eval_kernel_stats.txt
cd $R
timeout 200 python tools/bench_eval.py 10000 500 > $OUT/bench_eval.json 2> $OUT/bench_eval.err; echo "eval rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_eval.json').read().strip().splitlines()[-1]); print(d['fp32']['users_per_s_device'], d['bf16']['users_per_s_device'], d['bf16_batch_2000'], d['bf16']['device_equals_host_metrics'], d['fp32']['device_equals_host_metrics'])"
