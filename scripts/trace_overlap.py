"""Reads a rocprofv3 kernel_trace.csv and reports, for the last `window_ms` of the trace, how much kernels overlap:
sum of durations / union of busy time, and a per-kernel table."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
end = rows[-1][1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
sel = [r for r in rows if r[0] >= end - win]
tot = sum(e - s for s, e, *_ in sel)
busy = 0; cur_s = cur_e = None
for s, e, *_ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{len(sel)} dispatches in the last {win/1e6:.0f} ms: sum of durations {tot/1e6:.2f} ms, busy (union) {busy/1e6:.2f} ms, wall {(sel[-1][1]-sel[0][0])/1e6:.2f} ms, overlap factor {tot/busy:.2f}")
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, q, st in sel:
    per[(n, q)][0] += 1; per[(n, q)][1] += e - s
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k[0]:42s} queue {k[1]:>3s}  n {v[0]:5d}  avg {v[1]/v[0]/1e3:8.1f} us")
