T=${1:-r05i}; O=gpurun_out/$T; mkdir -p $O; D=tests/diag
(timeout 300 python $D/attn_variants.py new=groma_amd/csrc/libgroma_hip.so old=$D/a_old.so sum=$D/a_sum.so ord=$D/a_ord.so g64_4=$D/a_g64_4.so > $O/attn_ilp.txt 2>&1)
(timeout 300 python $D/attn_variants.py --ref new=groma_amd/csrc/libgroma_hip_ref.so old=$D/ar_old.so sum=$D/ar_sum.so g4=$D/ar_g4.so > $O/attn_ilp_ref.txt 2>&1)
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_ref_gpu.py tests/test_e2e_unchained_gpu.py -q -x --timeout 600 2>&1 | tail -6 > $O/tests.log)
grep -v amdgpu $O/attn_ilp.txt; grep -v amdgpu $O/attn_ilp_ref.txt; tail -4 $O/tests.log
