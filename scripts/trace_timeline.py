"""Prints the dispatch timeline (start offset, duration, queue, kernel, grid) of the last frame in a rocprofv3 kernel trace."""
import csv, sys, glob
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", "")[:34], r["Queue_Id"], int(r["Grid_Size_X"])))
rows.sort()
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2.2e6
end = rows[-1][1]
sel = [r for r in rows if r[0] >= end - span]
t0 = sel[0][0]
for s, e, n, q, g in sel:
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  q{q}  {n:34s} grid {g}")
