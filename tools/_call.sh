OUT=gpurun_out/r4k; mkdir -p $OUT
timeout 200 build/native/test_gemm > $OUT/test_gemm.txt 2>&1; echo "test_gemm rc=$?"; grep -E "f32 adam|FAIL|PASSED|FAILED" $OUT/test_gemm.txt | head
timeout 400 build/native/test_engine > $OUT/test_engine.txt 2>&1; echo "test_engine rc=$?"; tail -1 $OUT/test_engine.txt
timeout 600 python -m pytest tests -q -x -m gpu -k "dp_world or ml20m_shape or test_g1 or test_g2 or test_g3 or test_g4 or config0 or config3 or random_arch or g8_epoch or g9 or custom_op or reference_test" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log
timeout 200 python bench.py --steps 100 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fp32_parity'])"
timeout 100 python bench.py --steps 50 --numerics fp32 --no-cpu-baseline --no-extras --opt fuse_adam_f32=0 > $OUT/bench_fp32_unfused.json 2> $OUT/bench_fp32_unfused.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_fp32_unfused.json').read().strip().splitlines()[-1]); print('fp32 unfused', d['ms_per_step'])"
