OUT=gpurun_out/r4r; mkdir -p $OUT
for i in 1 2 3 4 5 6; do
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29530+i)) tests/dp_world2_onegpu_check.py > $OUT/w2_$i.out 2> $OUT/w2_$i.err; rc=$?; echo "w2 run $i rc=$rc"
if [ $rc -ne 0 ]; then grep -E "AssertionError" $OUT/w2_$i.err | head -1 | cut -c1-250; fi
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29522 tests/dp_world8_onegpu_check.py > $OUT/w8.out 2> $OUT/w8.err; echo "w8 rc=$?"; grep -E "AssertionError|Error" $OUT/w8.err | grep -v elastic | head -3 | cut -c1-600; grep -v Gloo $OUT/w8.out | tail -20
