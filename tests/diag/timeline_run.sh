# rocprofv3 kernel trace + tests/diag/timeline.py of one bench configuration:  bash tests/diag/timeline_run.sh <tag> <name> [bench args]
T=$1; NAME=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_$NAME -o p -- python $R/bench.py --no-cpu-baseline --no-traffic --no-extras --steps 5 --warmup 3 "$@" > $R/gpurun_out/$T/bench_$NAME.json 2>/dev/null
cd $R
F=$(find gpurun_out/$T/prof_$NAME -name "*kernel_trace.csv" | head -1)
python tests/diag/timeline.py $F > gpurun_out/$T/timeline_$NAME.txt 2>&1
cp $(find gpurun_out/$T/prof_$NAME -name "*kernel_stats.csv" | head -1) gpurun_out/$T/stats_$NAME.csv
gzip -c $F > gpurun_out/$T/trace_$NAME.csv.gz   # (a few MB: kept for offline analysis)
rm -rf gpurun_out/$T/prof_$NAME
