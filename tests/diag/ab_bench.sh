#!/bin/bash
# one-box A/B of two library builds through bench.py:  bash tests/diag/ab_bench.sh <libA.so> <libB.so> <rounds> [bench args]
A=$1; B=$2; R=$3; shift 3
for i in $(seq 1 $R); do for L in $A $B; do
  python tests/diag/bench_variant.py $L --no-cpu-baseline --no-traffic --no-extras --steps 10 --warmup 3 "$@" 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$(basename $L .so)', '$*', round(d['value'], 2), 'img/s', round(d['ms_per_step'], 2), 'ms')"
done; done
