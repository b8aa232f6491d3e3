# SQ counters of the headline step's kernels (one pass, 8 SQ slots), summarised per kernel: bash tests/diag/pmc_sq.sh <tag> [bench args]
T=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $R/gpurun_out/$T/pmc_sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --no-extras --no-prefill-graphs "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/$T/pmc_sq/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); n[k] += 1
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:10]
print("kernel                                    launches  wave_cycles  wait_any%  wait_inst%  active_inst%  mfma_busy/busy  lds_conflict/wave_cycles")
for k, c in rows:
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k:40s} {n[k]:8d} {wc:12.3e} {100*c.get('SQ_WAIT_ANY',0)/wc:9.1f} {100*c.get('SQ_WAIT_INST_ANY',0)/wc:10.1f} {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:12.1f} "
          f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(c.get('SQ_BUSY_CYCLES',0),1):14.3f} {c.get('SQ_LDS_BANK_CONFLICT',0)/wc:12.4f}")
PY
rm -rf gpurun_out/$T/pmc_sq
