#!/bin/bash
# One GPU-box pass of the round's MEASUREMENTS (the parity suite is `pytest -m gpu`, run separately): the counters (three --pmc
# passes), bench.py (default line citing them), a rocprofv3 kernel trace of bench.py turned into the per-kernel summary and a
# two-step timeline, the first-layer / stream / data-parallel variants, the multi-rank bench path over gloo on one GPU, the
# evaluation / Netflix-shape / EASE / SVAE lines.  Every command has its own timeout.
# Outputs: gpurun_out/$1/.   usage: tools/gpu.sh --timeout 1500 -- 'bash tools/round_check.sh r4final'
OUT=gpurun_out/${1:-check}
mkdir -p $OUT
# the driver's own invocation, verbatim (its line is the graded one): kept as profiles/r6_bench_driver_cmd.json
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver cmd rc=$?"; tail -1 $OUT/bench_driver_cmd.json | cut -c1-200
B="--steps 100 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 150 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%.1f us/step  %.0f users/s  %s  sched=%s  %s' % (d['ms_per_step']*1e3, d['value'], d['config']['parallelism'], d['config'].get('dp_scheduler'), d['config']['first_layer']))
except Exception as e: print('no line', e)
")"; }
run fused
run no_prefetch --no-prefetch
run sparse_first_layer --first-layer sparse
run one_stream --opt two_stream=0
run generic_kernels --opt small_fwd=0 --opt small_bwd=0
run logits_f32 --opt logits16=0
run dp1_rccl_replicated --force-dp --replicated
run dp1_rccl_sharded --force-dp --sharded
run emu8_sharded --emulate-world 8
run emu2_sharded --emulate-world 2
run netflix_b4096_fused --workload netflix
bash tools/prof_cmd.sh ${1:-check}/prof_fused
RTX_DIST_BACKEND=gloo timeout 300 python3 bench.py --gpus 8 --steps 3 --warmup 1 --windows 1 > $OUT/bench_8ranks_gloo_onegpu.json 2> $OUT/bench_8ranks_gloo_onegpu.err; echo "8 gloo ranks (self-launched) rc=$?"
timeout 200 python tools/bench_eval.py 10000 500 > $OUT/bench_eval.json 2> $OUT/bench_eval.err; echo "eval rc=$?"
timeout 200 python tools/bench_ease.py > $OUT/bench_ease.json 2> $OUT/bench_ease.err; echo "ease rc=$?"
timeout 200 python tools/bench_svae.py > $OUT/bench_svae.json 2> $OUT/bench_svae.err; echo "svae rc=$?"
# LAST: the counter passes (a pass that overruns its timeout can leave a profiled process behind -- round 5 lost a whole set of
# measurements to one -- so nothing is measured after them but the line that cites them)
timeout 420 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc/* $OUT/ 2>/dev/null
timeout 300 python bench.py --pmc-json gpurun_out/pmc/pmc_summary.json > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -1 $OUT/bench_default.json | cut -c1-330
