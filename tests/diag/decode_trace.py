"""Aggregate the LAST decode step of a rocprofv3 kernel trace (tests/decode_prof.py) by kernel and grid size."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(('sample_rows', 'argmax'))]  # the step's sampler
a, b = idx[-2], idx[-1]
agg = collections.OrderedDict()
t0 = int(rows[a + 1]['Start_Timestamp']); t1 = int(rows[b]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    k = (r['Kernel_Name'][:44], r['Grid_Size_X'], r['Workgroup_Size_X'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    e = agg.setdefault(k, [0, 0]); e[0] += 1; e[1] += d
tot = 0
for k, (c, d) in agg.items():
    print(f"{k[0]:46s} grid {k[1]:>8s} x{k[2]:>5s}  calls {c:3d}  avg {d / c / 1e3:6.1f} us  total {d / 1e3:7.1f} us"); tot += d
print(f"sum of kernel time {tot / 1e3:.1f} us, wall {(t1 - t0) / 1e3:.1f} us, launches {sum(c for c, _ in agg.values())}")
