#!/bin/bash
OUT=gpurun_out/r5c13; mkdir -p $OUT
B="--steps 60 --warmup 10 --no-cpu-baseline --no-fp32-parity --no-extras"
run() { name=$1; shift; timeout 150 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', '%.1f us' % (d['ms_per_step']*1e3), {k:(round(v,1) if isinstance(v,float) else v) for k,v in (d['comm'] or {}).get('exchange_us',{}).items() if v}, 'adam', round(d['roofline']['avg_us'] or 0,1))
except Exception as e: print('$name no line', e)
PY
}
run fused
run emu8_fold --emulate-world 8
run emu8_nofold --emulate-world 8 --opt hop_fold=0
run dp1_repl_fold --force-dp --replicated
run dp1_repl_nofold --force-dp --replicated --opt hop_fold=0
RTX_DP_ONE_COMM=1 run dp1_repl_onecomm --force-dp --replicated
bash tools/prof_cmd.sh r5c13/prof_emu8 --emulate-world 8
head -44 gpurun_out/r5c13/prof_emu8/timeline.txt
