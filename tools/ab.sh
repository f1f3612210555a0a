#!/bin/bash
# A/B of engine knobs in ONE gpurun call, alternating (box-to-box spread is larger than most effects): every argument is one arm's
# bench.py options ("" = defaults); arms run round-robin for R rounds.   usage: tools/ab.sh R "" "--opt small_waves=1" ...
R=$1; shift
B="--steps 100 --warmup 5 --no-cpu-baseline --no-fp32-parity --no-extras"
for r in $(seq 1 $R); do
  for arm in "$@"; do
    out=$(timeout 150 python3 bench.py $B $arm 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us %s' % (d['ms_per_step']*1e3, [round(x*1e3,1) for x in d.get('windows_ms_per_step', d.get('ms_per_step_windows', []))]))" 2>/dev/null)
    echo "round $r [${arm:-default}] $out"
  done
done
