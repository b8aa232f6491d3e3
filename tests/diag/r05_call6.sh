T=${1:-r05g}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests/test_fp8_gpu.py -q -x --timeout 600 2>&1 | tail -5 > $O/tests.log)
(timeout 200 python tests/diag/gemv_w8_bench.py 4 mfma2 > $O/gemv_w8.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc -- python $R/tests/diag/gemv_w8_bench.py 4 pmc > /dev/null 2>&1
cd $R
python tests/diag/pmc_fetch.py $O/pmc gemv > $O/pmc_fetch.txt 2>&1; rm -rf $O/pmc
(timeout 300 python bench.py --dtype fp8 --mode generate --batch 4 --steps 3 --warmup 2 --no-traffic --no-cpu-baseline > $O/bench_gen_fp8.json 2>$O/bench_gen_fp8.err)
tail -3 $O/tests.log; grep -v amdgpu $O/gemv_w8.txt; cat $O/pmc_fetch.txt; python - <<PY
import json
d = json.loads(open("$O/bench_gen_fp8.json").read().strip().splitlines()[-1])
print("fp8 generate img/s", d["value"], "ms/step", d["ms_per_step"])
PY
