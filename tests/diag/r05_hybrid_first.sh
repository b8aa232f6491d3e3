# round 5, GPU call 1: the hybrid build -- parity (unchained), smoke, default bench line, kernel stats / timeline
T=${1:-r05a}
mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_e2e_unchained_gpu.py tests/test_fulldepth_parity_gpu.py::test_full_depth_hybrid_precision_unchained -q -x -s --timeout 600 2>&1 | grep -v "^$" | tail -80 > gpurun_out/$T/hybrid_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.log 2>&1)
t0=$(date +%s)
(timeout 700 python bench.py --gemm-breakdown gpurun_out/$T/gemm_shapes_b14.txt > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err)
echo "wall seconds: $(( $(date +%s) - t0 ))" > gpurun_out/$T/bench_wall.txt
bash tests/diag/timeline_run.sh $T hybrid
bash tests/diag/timeline_run.sh $T bf16vit --vit-operands same
tail -3 gpurun_out/$T/hybrid_tests.log; tail -3 gpurun_out/$T/smoke.log; head -c 600 gpurun_out/$T/bench.json; echo; cat gpurun_out/$T/bench_wall.txt; head -30 gpurun_out/$T/timeline_hybrid.txt
