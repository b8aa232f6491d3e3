# one-box A/B of a python-side flag through bench_variant.py:  bash tests/diag/ab_flag.sh <flag> <tag> [bench args]
FLAG=$1; T=$2; shift 2
mkdir -p gpurun_out/$T
L=groma_amd/csrc/libgroma_hip.so
B="--no-cpu-baseline --no-traffic --no-extras --steps 20 --warmup 4"
for r in 1 2; do for v in 0 1; do
  python tests/diag/bench_variant.py $L $FLAG=$v $B "$@" > gpurun_out/$T/o.json 2> gpurun_out/$T/err_$v.txt
  python -c "import json; d=json.loads(open('gpurun_out/$T/o.json').read()); print('$FLAG=$v', '$*', d['value'], d['ms_per_step'])" 2>&1 | tail -1
done; done | tee -a gpurun_out/$T/ab.txt
