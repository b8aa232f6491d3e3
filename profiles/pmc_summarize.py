"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -- python bench.py --steps 1 --warmup 1 --batch B --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -- python bench.py --steps 1 --warmup 1 --batch B --no-cpu-baseline

into profiles/<round>_pmc_traffic_b<B>.json: per kernel, average counter value per launch (KB) and bytes per launch with the
guide's gfx950 correction (FETCH_SIZE counts a wide streaming read at half its bytes -> x2; WRITE_SIZE taken as is).
usage: python profiles/pmc_summarize.py <dir> <batch> [round tag, default r02]"""
import csv, glob, json, os, sys, collections

d, batch = sys.argv[1], int(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else "r02"


def load(sub, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


fe, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
out = {"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 --batch {batch} "
                  "--no-cpu-baseline (two separate passes)",
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide-stream correction); counters are L2<->fabric requests "
               "and include Infinity-Cache hits", "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    f = fe.get(k, [0.0, 0]); w = wr.get(k, [0.0, 0])
    fk = f[0] / max(f[1], 1); wk = w[0] / max(w[1], 1)
    out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": fk, "launches": max(f[1], w[1]), "WRITE_SIZE_KB_per_launch": wk,
                         "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{tag}_pmc_traffic_b{batch}.json")
json.dump(out, open(p, "w"), indent=1)
print(p, len(out["kernels"]), "kernels")
