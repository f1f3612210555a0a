#!/bin/bash
OUT=gpurun_out/${1:-r3m}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 400 bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; cp gpurun_out/pmc/* $OUT/ 2>/dev/null
timeout 400 python bench.py --pmc-json gpurun_out/pmc/pmc_summary.json > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -1 $OUT/bench_default.json | cut -c1-400
