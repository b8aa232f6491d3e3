# the driver-form default run with its wall time: bash tests/diag/default_bench.sh <tag>
T=$1; mkdir -p gpurun_out/$T
t0=$(date +%s)
python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
echo "wall seconds: $(( $(date +%s) - t0 ))" | tee gpurun_out/$T/bench_wall.txt
python - <<PY
import json
d = json.loads(open("gpurun_out/$T/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["graph_captures_in_timed_region"], d["graph_setup_steps"], d["cpu_baseline"]["value"])
for k, v in d["extras"].items():
    if isinstance(v, dict):
        print(k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"))
PY
