"""Turns rocprofv3 --pmc passes of SQ counters (separate runs, <= 8 SQ counters each) into profiles/r01e_sq_counters.json:
per kernel of the render path, averaged per launch — instruction mix (VALU / f64 add, mul, fma, transcendental / SALU /
SMEM / LDS / vector loads), and how busy the SIMDs were issuing VALU work while wavefronts were resident.

usage: python scripts/pmc_sq.py <out.json> <pass_dir> [<pass_dir> ...]"""
import collections, csv, json, re, sys

def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d0 in dirs:
        for r in csv.DictReader(open(f"{d0}/p_counter_collection.csv")):
            name = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
            name = re.sub(r"<.*", "", name).replace("rt::", "")
            if not name.startswith("k_"):
                continue
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    res = {}
    for k, cs in sorted(acc.items()):
        row = {c: round(v[0] / max(v[1], 1), 1) for c, v in sorted(cs.items())}
        row["launches_sampled"] = max(v[1] for v in cs.values())
        if row.get("SQ_WAVE_CYCLES") and row.get("SQ_ACTIVE_INST_VALU"):
            # both are summed over wavefronts / SIMDs; their ratio is the share of resident-wavefront time spent issuing VALU
            row["valu_issue_share_of_wave_cycles"] = round(row["SQ_ACTIVE_INST_VALU"] / row["SQ_WAVE_CYCLES"], 4)
        if row.get("SQ_INSTS_VALU"):
            f64 = sum(row.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
            row["f64_share_of_valu_insts"] = round(f64 / row["SQ_INSTS_VALU"], 4)
        res[k] = row
    res["_method"] = ("rocprofv3 --pmc <counters> --kernel-trace in separate runs of `bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
                      "(one pass per counter group, at most 8 SQ counters each); values are per-launch averages over every launch "
                      "of the kernel in the run (all recursion levels); instruction counters count wavefront instructions")
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k.startswith("k_"):
            print(k, {c: v[c] for c in v if c.startswith("SQ_INSTS") or c.endswith("share_of_wave_cycles") or c.endswith("valu_insts")})

if __name__ == "__main__":
    main()
