# round 5: the region extraction queued before the NMS counts reach the host -- A/B (interleaved, same box) at 14 / 4 / 1 images per
# call, then the full -m gpu suite, smoke, the default bench line and the timeline of the step.   bash tests/diag/r05_spec_extract.sh <tag>
T=${1:-r05n}; O=gpurun_out/$T; mkdir -p $O
B="--no-cpu-baseline --no-traffic --no-extras --steps 20 --warmup 5"
for r in 1; do for b in 14 1; do for s in 0 1; do
  echo "round $r batch $b speculative_extract=$s $(timeout 200 python tests/diag/bench_spec_extract.py $s $B --batch $b 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")')" >> $O/spec_extract_ab.txt
done; done; done
cat $O/spec_extract_ab.txt
(timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -30 > $O/gpu_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1)
t0=$(date +%s)
(timeout 900 python bench.py --gemm-breakdown $O/gemm_shapes_b14.txt > $O/bench.json 2> $O/bench.err)
echo "wall seconds: $(( $(date +%s) - t0 ))" > $O/bench_wall.txt
bash tests/diag/timeline_run.sh $T hybrid
tail -4 $O/gpu_tests.log; tail -3 $O/smoke.log; head -c 1100 $O/bench.json; echo; cat $O/bench_wall.txt; head -12 $O/timeline_hybrid.txt; grep -A6 "largest idle" $O/timeline_hybrid.txt
