# round 5, GPU call 3: full -m gpu suite on the new host code, pair-attention A/B, e4m3 K-loop peel A/B, default bench + timeline
T=${1:-r05c}; O=gpurun_out/$T; mkdir -p $O
D=tests/diag
(timeout 300 python $D/attn_variants.py --ref new=groma_amd/csrc/libgroma_hip_ref.so r04=$D/attr_r04.so g4=$D/attr_g4.so g1=$D/attr_g1.so > $O/attn_pair_ab.txt 2>&1)
(timeout 300 python $D/gemm_variants.py --fp8 peel=groma_amd/csrc/libgroma_hip.so nopeel=$D/g8_nopeel.so > $O/gemm_peel_ab_fp8.txt 2>&1)
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -40 > $O/gpu_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
t0=$(date +%s)
(timeout 700 python bench.py --gemm-breakdown $O/gemm_shapes_b14.txt > $O/bench.json 2> $O/bench.err)
echo "wall seconds: $(( $(date +%s) - t0 ))" > $O/bench_wall.txt
bash $D/timeline_run.sh $T hybrid
grep -v amdgpu $O/attn_pair_ab.txt; tail -8 $O/gemm_peel_ab_fp8.txt; tail -5 $O/gpu_tests.log; tail -3 $O/smoke.log; head -c 900 $O/bench.json; echo; cat $O/bench_wall.txt; head -14 $O/timeline_hybrid.txt
