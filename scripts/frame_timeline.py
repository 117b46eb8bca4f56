"""Diagnostic: the launches of ONE frame in dispatch order, from a rocprofv3 --kernel-trace run (kernel, duration, gap to the
previous launch).  usage: python scripts/frame_timeline.py <rocprof dir> [which frame, counted from the end, default 1]"""
import csv, glob, re, sys
d0 = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = [r for fn in glob.glob(f"{d0}/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(fn))]
ev = []
for r in rows:
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"])).replace("rt::", "")
    if name.startswith("k_"):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_raygen")]
lo = starts[-back]
hi = starts[-back + 1] if back > 1 else len(ev)
t0 = ev[lo][0]
prev_end = t0
for s, e, name in ev[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    prev_end = e
print(f"frame: {(ev[hi - 1][1] - t0) / 1e3:.1f} us")
