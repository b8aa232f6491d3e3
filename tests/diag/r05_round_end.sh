# round 5 measurement pass: full -m gpu suite, smoke, the driver-form bench line (+ extras), kernel stats / timeline of the forward,
# generate benches (bf16-behind-the-ViT and e4m3) with decode traces, serving throughput.   bash tests/diag/r05_round_end.sh <tag>
T=${1:-r05z}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -30 > $O/gpu_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
t0=$(date +%s)
(timeout 900 python bench.py --gemm-breakdown $O/gemm_shapes_b14.txt > $O/bench.json 2> $O/bench.err)
echo "wall seconds: $(( $(date +%s) - t0 ))" > $O/bench_wall.txt
bash tests/diag/timeline_run.sh $T hybrid
(timeout 300 python bench.py --mode generate --batch 4 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_gen.json 2>$O/bench_gen.err)
(timeout 300 python tests/serve_bench.py > $O/serve_bench.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gen -o gen -- python $R/bench.py --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/$O/bench_gen_prof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gen8 -o gen -- python $R/bench.py --dtype fp8 --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/$O/bench_gen8_prof.json 2>/dev/null
cd $R
python tests/diag/decode_trace.py $(find $O/prof_gen -name "*kernel_trace.csv" | head -1) > $O/decode_step.txt 2>&1
python tests/diag/decode_trace.py $(find $O/prof_gen8 -name "*kernel_trace.csv" | head -1) > $O/decode_step_fp8.txt 2>&1
rm -rf $O/prof_gen $O/prof_gen8
tail -4 $O/gpu_tests.log; tail -3 $O/smoke.log; head -c 1100 $O/bench.json; echo; cat $O/bench_wall.txt; head -12 $O/timeline_hybrid.txt; tail -3 $O/serve_bench.txt; tail -2 $O/decode_step.txt; tail -2 $O/decode_step_fp8.txt
