"""Per-kernel launch durations of a rocprofv3 --kernel-trace run of bench.py, next to the HIP-event figure bench.py
printed in that same run (roofline.avg_launch_ms): the two clocks must agree.
usage: python scripts/launch_durations.py <rocprof dir> <bench json of the same run> <out.json>"""
import collections, csv, glob, json, re, sys

d0, bench_json, out = sys.argv[1:4]
rows = [r for fn in glob.glob(f"{d0}/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(fn))]
acc = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"<.*", "", re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))).replace("rt::", "")
    if name.startswith("k_"):
        acc[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
res = {"_source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline; kernel_trace.csv of that run",
       "bench_same_run": {"ms_per_step": bench["ms_per_step"], "roofline_kernel": bench["roofline"]["kernel"],
                          "avg_launch_ms_hip_events": bench["roofline"]["avg_launch_ms"],
                          "launches_per_frame": bench["roofline"]["launches_per_frame"]}}
for k, v in sorted(acc.items()):
    durs = [d for _, d in v]
    res[k] = {"dispatches": len(durs), "avg_us": round(sum(durs) / len(durs) / 1e3, 2), "min_us": round(min(durs) / 1e3, 2),
              "max_us": round(max(durs) / 1e3, 2)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
