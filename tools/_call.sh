OUT=gpurun_out/r4u; mkdir -p $OUT
T=build/native/test_gemm
for i in 1 2; do
for v in _nosplit _early0 _early1 "" _early3; do
timeout 40 ${T}$v dwx 0 base > $OUT/v${v}_$i.txt 2>&1; echo "variant '$v' $i rc=$? $(grep 'perf dw' $OUT/v${v}_$i.txt | sed 's/.*K=500: //' | cut -c1-9 | tr '\n' ' ')"
done
done
for v in _early0 _early1; do timeout 40 ${T}$v dwx 0 stamps > $OUT/stamps$v.txt 2>&1; grep -A1 "stamped" $OUT/stamps$v.txt | grep "mean us" | cut -c1-330; done
