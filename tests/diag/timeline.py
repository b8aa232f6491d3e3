"""Where a step's WALL time goes: reads a rocprofv3 --kernel-trace CSV of bench.py and, for each forward (patchify .. next
patchify), reports the GPU-idle gaps (nothing running), the time with exactly one kernel running attributed to that kernel, and
the overlapped time.  usage: python tests/diag/timeline.py <kernel_trace.csv> [step_index_from_end=1]"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "")
    for cut in ("(", "<"):
        if cut in n and not n.startswith("at::"):
            n = n.split(cut)[0]
    return n[:60]


def main(path, back=1):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("patchify_kernel")]
    if len(marks) < back + 1:
        raise SystemExit("not enough forwards in the trace")
    lo, hi = marks[-back - 1], marks[-back]
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], rows[hi][0]
    print(f"step window: {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels")
    # sweep
    ev = []
    for s, e, n, q, st in seg:
        ev.append((s, 1, n)); ev.append((min(e, t1), -1, n))
    ev.sort(key=lambda x: (x[0], -x[1]))
    active = defaultdict(int)
    nact, last = 0, t0
    idle, solo, multi = 0, defaultdict(int), 0
    gaps = []
    prev_end_name = None
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            if nact == 0:
                idle += dt
                gaps.append((dt, last - t0, prev_end_name, n))
            elif nact == 1:
                solo[next(k for k, v in active.items() if v > 0)] += dt
            else:
                multi += dt
        last = t
        active[n] += d
        nact += d
        if d < 0:
            prev_end_name = n
    tot = t1 - t0
    print(f"idle (no kernel running): {idle / 1e6:.3f} ms = {100 * idle / tot:.1f} %   overlapped (>=2 kernels): {multi / 1e6:.3f} ms")
    agg = defaultdict(int)
    for k, v in solo.items():
        agg[short(k)] += v
    print("time with exactly ONE kernel running, by kernel:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:28]:
        print(f"  {v / 1e6:8.3f} ms  {100 * v / tot:5.1f} %  {k}")
    print("largest idle gaps (dur us, at ms, after -> before):")
    for dt, at, a, b in sorted(gaps, reverse=True)[:15]:
        print(f"  {dt / 1e3:8.1f} us at {at / 1e6:8.3f} ms   {short(a or '-')} -> {short(b)}")
    small = sum(dt for dt, *_ in gaps if dt < 20000)
    print(f"gaps < 20 us: {sum(1 for g in gaps if g[0] < 20000)} totalling {small / 1e6:.3f} ms; >= 20 us: {sum(1 for g in gaps if g[0] >= 20000)} totalling {(idle - small) / 1e6:.3f} ms")
    # totals per kernel (sum of durations, incl. overlap)
    dur = defaultdict(lambda: [0, 0])
    for s, e, n, q, st in seg:
        dur[short(n)][0] += e - s; dur[short(n)][1] += 1
    print("sum of kernel durations in the step (overlap counted twice):")
    for k, (v, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"  {v / 1e6:8.3f} ms  x{c:4d}  avg {v / c / 1e3:8.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
