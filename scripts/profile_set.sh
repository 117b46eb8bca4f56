#!/bin/bash
# usage: scripts/profile_set.sh <tag>      (on the GPU box, from the repo root; every GPU step under its own time limit)
# The per-round profile set behind profiles/<tag>_*: rocprofv3 kernel stats of the default bench command, the two PMC
# traffic passes, the default bench line (with cpu_baseline) and the other BASELINE configs.
set -e
tag=$1
root=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $root/gpurun_out/${tag}_prof
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_prof -- python3 $root/bench.py --no-cpu-baseline --no-c5 --traffic file --in-flight 1 > $root/gpurun_out/${tag}_bench_under_rocprof.json 2> $root/gpurun_out/${tag}_prof.err
cd $root
bash scripts/pmc_collect.sh $tag 1920 1080 --steps 2 --warmup 1 --no-cpu-baseline --no-c5
cp gpurun_out/pmc_${tag}.json profiles/pmc_traffic.json   # what the bench line below quotes as roofline.traffic
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
bash scripts/all_configs.sh
find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
python scripts/launch_durations.py gpurun_out/${tag}_prof gpurun_out/${tag}_bench_under_rocprof.json gpurun_out/${tag}_launch_durations.json
