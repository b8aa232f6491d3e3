"""mean FETCH_SIZE / WRITE_SIZE per kernel name from a rocprofv3 --pmc run:  python tests/diag/pmc_fetch.py <dir> [name filter]
(FETCH_SIZE in KB at the L2 <-> fabric boundary; gfx950 counts a wide streaming read at half its bytes -> x2, MI355X_MICROARCH.md)"""
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in agg.items():
    if flt in k:
        print(k, {c: (len(x), round(sum(x) / len(x) * (2 if c == "FETCH_SIZE" else 1) * 1024 / 1e6, 2)) for c, x in v.items()}, "(calls, MB per launch)")
