mkdir -p gpurun_out/r04f
(timeout 1300 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -60 > gpurun_out/r04f/gpu_tests.log)
(timeout 600 python bench.py > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f/prof_fwd -o fwd -- python $R/bench.py --no-cpu-baseline --no-traffic --no-extras --steps 5 --warmup 3 > $R/gpurun_out/r04f/bench_prof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f/prof_gen -o gen -- python $R/bench.py --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/r04f/bench_gen_prof.json 2>/dev/null
cd $R
F=$(find gpurun_out/r04f/prof_fwd -name "*kernel_trace.csv" | head -1); G=$(find gpurun_out/r04f/prof_gen -name "*kernel_trace.csv" | head -1)
python tests/diag/timeline.py $F > gpurun_out/r04f/timeline_b14.txt 2>&1
python tests/diag/decode_trace.py $G > gpurun_out/r04f/decode_step.txt 2>&1
cp $(find gpurun_out/r04f/prof_fwd -name "*kernel_stats.csv" | head -1) gpurun_out/r04f/fwd_kernel_stats.csv
cp $(find gpurun_out/r04f/prof_gen -name "*kernel_stats.csv" | head -1) gpurun_out/r04f/gen_kernel_stats.csv
rm -rf gpurun_out/r04f/prof_fwd gpurun_out/r04f/prof_gen
tail -4 gpurun_out/r04f/gpu_tests.log; head -c 300 gpurun_out/r04f/bench.json; echo; tail -3 gpurun_out/r04f/decode_step.txt; head -12 gpurun_out/r04f/timeline_b14.txt
