#!/bin/bash
# Round-6 probes, one GPU-box pass -> gpurun_out/r6probes/ (copied to profiles/r6_*.txt): where the K = n_items products' time goes
# (ablation), the optimizer state's access pattern alone (2/4/6 streams, tiles vs flat), the skinny products on every kernel, the
# 64-byte-slice logits tile, nt vs plain state loads in the fused weight kernel, the Adam-approximation isolation test, the stress test.
OUT=gpurun_out/r6probes; mkdir -p $OUT
make -C tests/native ../../build/native/test_gemm_ablate > /dev/null 2>&1
timeout 200 build/native/test_gemm_ablate ablate > $OUT/gemm_ablation.txt 2>&1; echo "ablate rc=$?"
timeout 200 build/native/test_gemm streams > $OUT/state_stream.txt 2>&1; echo "streams rc=$?"
timeout 200 build/native/test_gemm skinny > $OUT/skinny_products.txt 2>&1; echo "skinny rc=$?"
timeout 200 build/native/test_gemm logits > $OUT/logits_k32_tile.txt 2>&1; echo "logits rc=$?"
timeout 200 build/native/test_gemm adamiso > $OUT/adam_approx_isolation.txt 2>&1; echo "adamiso rc=$?"
timeout 200 build/native/test_gemm dwperf > $OUT/dw_perf.txt 2>&1; echo "dwperf rc=$?"
timeout 900 python tests/stress_sync_check.py > $OUT/stress_sync.txt 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress_sync.txt
bash tools/ab.sh 3 "" "--opt small_waves=4" "--opt dw_side_pad=12288" > $OUT/ab_small_waves_side_pad.txt 2>&1; cat $OUT/ab_small_waves_side_pad.txt
