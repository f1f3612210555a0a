#!/bin/bash
OUT=gpurun_out/${1:-r3q}
mkdir -p $OUT
# the N > 1 code path of bench.py end to end: two ranks sharing the one GPU of the box, gloo carrying the collectives the
# engine asks for (RCCL refuses two ranks on one device) -- slow, but every line of the multi-rank path runs
RTX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --windows 1 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2 ranks rc=$?"; tail -1 $OUT/bench_2ranks_gloo.json | cut -c1-700; tail -3 $OUT/bench_2ranks_gloo.err | cut -c1-300
RTX_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --windows 1 --replicated > $OUT/bench_2ranks_gloo_repl.json 2> $OUT/bench_2ranks_gloo_repl.err; echo "2 ranks repl rc=$?"; tail -1 $OUT/bench_2ranks_gloo_repl.json | cut -c1-300
timeout 300 python bench.py --force-dp --sharded --steps 100 --no-cpu-baseline --no-fp32-parity --no-extras > $OUT/dp1.json 2> $OUT/dp1.err; echo "dp1 rc=$?"; tail -1 $OUT/dp1.json | cut -c1-200
