#!/bin/bash
# usage: scripts/c5_profile.sh <tag>     (on the GPU box, from the repo root)
# C5 at the survey's scale (grid 13, 8K, depth 8): the two PMC traffic passes and the bench line that quotes them.
set -e
tag=$1
bash scripts/pmc_collect.sh ${tag}_c5g13 7680 4320 --scene grid --grid 13 --width 7680 --height 4320 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline --no-c5
cp gpurun_out/pmc_${tag}_c5g13.json profiles/r03_pmc_traffic_c5_grid13.json
timeout -k 10 300 python bench.py --scene grid --grid 13 --width 7680 --height 4320 --depth 8 --no-cpu-baseline --no-c5 --steps 10 --warmup 2 > gpurun_out/${tag}_bench_c5_grid13.json 2> gpurun_out/${tag}_bench_c5_grid13.err
