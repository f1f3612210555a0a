#!/bin/bash
OUT=gpurun_out/${1:-r3z}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "svae" > $OUT/pytest_svae.log 2>&1; echo "pytest svae rc=$?"; tail -2 $OUT/pytest_svae.log
timeout 300 python tools/bench_svae.py --cpu-seconds 0 > $OUT/svae_rows.json 2> $OUT/svae_rows.err; tail -1 $OUT/svae_rows.json | cut -c1-400
RTX_SVAE_GRU_ROWS=0 timeout 300 python tools/bench_svae.py --cpu-seconds 0 > $OUT/svae_old.json 2> $OUT/svae_old.err; tail -1 $OUT/svae_old.json | cut -c1-400
timeout 300 python tools/bench_svae.py --cpu-seconds 0 --pack 64 --steps 300 > $OUT/svae_pack64.json 2> $OUT/svae_pack64.err; tail -1 $OUT/svae_pack64.json | cut -c1-300
