# round 5, GPU call 2: attention build variants, K-loop peel A/B, pair-GEMM tile forms, e4m3 prefill graphs
T=${1:-r05b}; O=gpurun_out/$T; mkdir -p $O
D=tests/diag
(timeout 300 python $D/attn_variants.py base=groma_amd/csrc/libgroma_hip.so prio=$D/att_prio.so stage=$D/att_stage.so noslp=$D/att_noslp.so slot=$D/att_slot.so skew=$D/att_skew.so all=$D/att_all.so > $O/attn_variants.txt 2>&1)
(timeout 300 python $D/attn_variants.py --ref base=groma_amd/csrc/libgroma_hip_ref.so prio=$D/attr_prio.so stage=$D/attr_stage.so noslp=$D/attr_noslp.so all=$D/attr_all.so > $O/attn_variants_ref.txt 2>&1)
(timeout 400 python $D/gemm_variants.py peel=groma_amd/csrc/libgroma_hip.so nopeel=$D/g_nopeel.so > $O/gemm_peel_ab.txt 2>&1)
(timeout 300 python $D/gemm_variants.py --ref peel=groma_amd/csrc/libgroma_hip_ref.so nopeel=$D/gr_nopeel.so > $O/gemm_peel_ab_ref.txt 2>&1)
(timeout 300 python $D/pair_tile_rows.py > $O/pair_tile_rows.txt 2>&1)
(timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_fp8_width_gpu.py tests/test_kernels_gpu.py tests/test_ref_gpu.py -q -x --timeout 600 2>&1 | tail -15 > $O/tests.log)
(timeout 300 python bench.py --dtype fp8 --no-extras --no-cpu-baseline --no-traffic --steps 5 --warmup 3 > $O/bench_fp8_graphs.json 2>$O/bench_fp8.err)
(timeout 300 python bench.py --dtype fp8 --no-extras --no-cpu-baseline --no-traffic --steps 5 --warmup 3 --no-prefill-graphs > $O/bench_fp8_eager.json 2>>$O/bench_fp8.err)
cat $O/attn_variants.txt $O/attn_variants_ref.txt | grep -v "^$" | tail -40; tail -12 $O/gemm_peel_ab.txt; tail -6 $O/gemm_peel_ab_ref.txt; tail -6 $O/pair_tile_rows.txt; tail -4 $O/tests.log; head -c 300 $O/bench_fp8_graphs.json; echo; head -c 300 $O/bench_fp8_eager.json; echo
