# round 5: e4m3 decode streams -- unit + model tests, per-shape bench, generate bench (fp8 / bf16), decode trace
T=${1:-r05d}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_fp8_width_gpu.py -q -x -s --timeout 600 2>&1 | tail -25 > $O/tests.log)
(timeout 200 python tests/diag/gemv_w8_bench.py 4 mfma > $O/gemv_w8.txt 2>&1)
(timeout 200 python tests/diag/gemv_w8_bench.py 8 mfma >> $O/gemv_w8.txt 2>&1)
(timeout 300 python bench.py --dtype fp8 --mode generate --batch 4 --steps 3 --warmup 2 --no-traffic --no-cpu-baseline > $O/bench_gen_fp8.json 2>$O/bench_gen_fp8.err)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gen -o gen -- python $R/bench.py --dtype fp8 --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic > $R/$O/bench_gen_prof.json 2>/dev/null
cd $R
G=$(find $O/prof_gen -name "*kernel_trace.csv" | head -1)
python tests/diag/decode_trace.py $G > $O/decode_step_fp8.txt 2>&1
rm -rf $O/prof_gen
tail -12 $O/tests.log; grep -v amdgpu $O/gemv_w8.txt; cat $O/decode_step_fp8.txt; python - <<PY
import json
d = json.loads(open("$O/bench_gen_fp8.json").read().strip().splitlines()[-1])
print("fp8 generate img/s", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "launches_per_step", "avg_launch_us", "bytes_per_launch")})
PY
