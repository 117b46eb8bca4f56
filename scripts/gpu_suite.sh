#!/bin/bash
# usage: scripts/gpu_suite.sh <tag> [pytest args...]   (GPU box): the -m gpu suite, then the default bench line.
# Every GPU step under its own time limit, chained: a failure ends the call.
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
python - <<P
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print("ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "host", d["rt_render_host"], "\nc5g4", d["c5_grid4_8k"] and d["c5_grid4_8k"]["ms_per_step"], "c5g13", d["c5_grid13_8k"] and d["c5_grid13_8k"]["ms_per_step"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {k:v["ms_per_frame"] for k,v in d["kernels"].items() if isinstance(v,dict)})
P
