T=${1:-r05e}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
(timeout 200 python tests/diag/gemv_w8_bench.py 4 w8 > $O/gemv_w8.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gen -o gen -- python $R/bench.py --dtype fp8 --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/$O/bench_gen_prof.json 2>/dev/null
cd $R
G=$(find $O/prof_gen -name "*kernel_trace.csv" | head -1)
python tests/diag/decode_trace.py $G > $O/decode_step_fp8.txt 2>&1
rm -rf $O/prof_gen
grep -v amdgpu $O/gemv_w8.txt; cat $O/decode_step_fp8.txt
