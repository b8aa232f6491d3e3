T=${1:-r05l}; O=gpurun_out/$T; mkdir -p $O
(timeout 200 python tests/diag/quant_variants.py reg=groma_amd/csrc/libgroma_hip.so twopass=tests/diag/q_old.so > $O/quant_ab.txt 2>&1)
(timeout 400 python -m pytest tests/test_fp8_gpu.py tests/test_fp8_width_gpu.py -q -x --timeout 600 2>&1 | tail -3 > $O/tests.log)
grep -v amdgpu $O/quant_ab.txt; tail -2 $O/tests.log
