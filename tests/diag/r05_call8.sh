T=${1:-r05j}; O=gpurun_out/$T; mkdir -p $O; D=tests/diag
(timeout 400 python $D/gemm_small_m.py deep256=groma_amd/csrc/libgroma_hip.so old=$D/g128_old.so deep512=$D/g128_512.so deep1024=$D/g128_1024.so > $O/gemm_small_m.txt 2>&1)
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_ref_gpu.py -q -x --timeout 600 2>&1 | tail -4 > $O/tests.log)
(timeout 200 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras > $O/bench_b1.json 2>/dev/null)
(timeout 200 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --gemm-plan latency > $O/bench_b1_lat.json 2>/dev/null)
grep -v amdgpu $O/gemm_small_m.txt; tail -3 $O/tests.log; head -c 330 $O/bench_b1.json; echo; head -c 330 $O/bench_b1_lat.json; echo
