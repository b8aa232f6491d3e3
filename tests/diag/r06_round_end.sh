# round 6 measurement pass: smoke, the driver-form bench line (+ extras, parity block), rocprofv3 kernel stats / timeline of the same
# command, SQ PMC counters per kernel, generate bench, serving throughput at 4..32 rows.   bash tests/diag/r06_round_end.sh <tag> [suite]
T=${1:-r06z}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
if [ "$2" = "suite" ]; then (timeout 1700 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -30 > $O/gpu_tests.log); fi
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
t0=$(date +%s)
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --gemm-breakdown $O/gemm_shapes_b14.txt > $O/bench.json 2> $O/bench.err)
echo "wall seconds: $(( $(date +%s) - t0 ))" > $O/bench_wall.txt
(timeout 400 python tests/serve_bench.py > $O/serve_bench.txt 2>&1)
(timeout 400 python tests/serve_bench.py --rows 8,16,32 --requests 64 --ragged > $O/serve_bench_ragged.txt 2>&1)
(timeout 300 python bench.py --mode generate --batch 4 --steps 3 --warmup 2 --no-cpu-baseline --no-parity > $O/bench_gen.json 2>$O/bench_gen.err)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fwd -o fwd -- python $R/bench.py --no-cpu-baseline --no-traffic --no-extras --no-parity --steps 5 --warmup 3 > $R/$O/bench_prof.json 2>/dev/null
cd $R
F=$(find $O/prof_fwd -name "*kernel_trace.csv" | head -1)
python tests/diag/timeline.py $F > $O/timeline_b14.txt 2>&1
cp $(find $O/prof_fwd -name "*kernel_stats.csv" | head -1) $O/fwd_kernel_stats.csv
rm -rf $O/prof_fwd
bash tests/diag/pmc_sq.sh $T --no-parity > $O/pmc_sq_b14.txt 2>&1
tail -4 $O/gpu_tests.log 2>/dev/null; tail -4 $O/smoke.log; head -c 1500 $O/bench.json; echo; cat $O/bench_wall.txt; head -14 $O/timeline_b14.txt; tail -5 $O/serve_bench.txt
