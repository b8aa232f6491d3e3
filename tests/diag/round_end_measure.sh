# Round-end measurement on the GPU box: the full -m gpu suite, smoke(), the driver-form bench line (+ extras), rocprofv3 kernel
# stats / timelines of the forward and the generate step, per-shape GEMM tables at 14 / 4 / 1 images per call, serving throughput.
# Everything lands under gpurun_out/<tag>/ ; the summaries worth keeping are copied into profiles/ by hand.
T=${1:-r04z}
mkdir -p gpurun_out/$T
(timeout 1300 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -60 > gpurun_out/$T/gpu_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.log 2>&1)
(timeout 700 python bench.py --gemm-breakdown gpurun_out/$T/gemm_shapes_b14.txt > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err)
(timeout 200 python bench.py --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --gemm-breakdown gpurun_out/$T/gemm_shapes_b4.txt > gpurun_out/$T/bench_b4.json 2>/dev/null)
(timeout 200 python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras --gemm-breakdown gpurun_out/$T/gemm_shapes_b1.txt > gpurun_out/$T/bench_b1.json 2>/dev/null)
(timeout 300 python tests/serve_bench.py > gpurun_out/$T/serve_bench.txt 2>&1)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_fwd -o fwd -- python $R/bench.py --no-cpu-baseline --no-traffic --no-extras --steps 5 --warmup 3 > $R/gpurun_out/$T/bench_prof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_gen -o gen -- python $R/bench.py --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/$T/bench_gen_prof.json 2>/dev/null
cd $R
F=$(find gpurun_out/$T/prof_fwd -name "*kernel_trace.csv" | head -1); G=$(find gpurun_out/$T/prof_gen -name "*kernel_trace.csv" | head -1)
python tests/diag/timeline.py $F > gpurun_out/$T/timeline_b14.txt 2>&1
python tests/diag/decode_trace.py $G > gpurun_out/$T/decode_step.txt 2>&1
cp $(find gpurun_out/$T/prof_fwd -name "*kernel_stats.csv" | head -1) gpurun_out/$T/fwd_kernel_stats.csv
cp $(find gpurun_out/$T/prof_gen -name "*kernel_stats.csv" | head -1) gpurun_out/$T/gen_kernel_stats.csv
rm -rf gpurun_out/$T/prof_fwd gpurun_out/$T/prof_gen
grep -n "passed\|failed" gpurun_out/$T/gpu_tests.log | tail -2; tail -2 gpurun_out/$T/smoke.log; head -c 200 gpurun_out/$T/bench.json; echo; cat gpurun_out/$T/serve_bench.txt | tail -3; tail -2 gpurun_out/$T/decode_step.txt
