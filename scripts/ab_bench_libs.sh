#!/bin/bash
# usage: scripts/ab_bench_libs.sh <variant lib.so>   (GPU box): bench lines of the headline and of C5 at the survey's scale,
# in-tree build and the variant library in turn, twice
var=$PWD/$1
run() {  # label, env assignment or "X=1"
  env $2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-c5 --traffic file --steps 100 > gpurun_out/ab_h.json 2>/dev/null || return 1
  env $2 timeout -k 10 300 python bench.py --scene grid --grid 13 --width 7680 --height 4320 --depth 8 --no-cpu-baseline --no-c5 --traffic file --steps 10 --warmup 2 > gpurun_out/ab_c.json 2>/dev/null || return 1
  python - "$1" <<'P'
import json, sys
h = json.load(open("gpurun_out/ab_h.json")); c = json.load(open("gpurun_out/ab_c.json"))
print("%-8s headline %.4f / %.4f ms   C5 grid 13 %.3f / %.3f ms   k_trace %.3f ms" % (sys.argv[1], h["ms_per_step"], h["one_frame_at_a_time"]["ms_per_frame"],
      c["ms_per_step"], c["one_frame_at_a_time"]["ms_per_frame"], c["kernels"]["k_trace"]["ms_per_frame"]), flush=True)
P
}
for round in 1 2; do
  run in-tree X=1 || exit 1
  run variant RT_AMD_LIB=$var || exit 1
done
