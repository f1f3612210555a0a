#!/bin/bash
# the round's closing pass: the whole GPU suite, smoke(), then tools/round_check.sh (driver command, counters, variants, profiles)
OUT=gpurun_out/r5final; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/parity_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/parity_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/round_check.sh r5final
