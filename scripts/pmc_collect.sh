#!/bin/bash
# usage: scripts/pmc_collect.sh <tag> <hsize> <vsize> <bench.py arguments...>   (on the GPU box, from the repo root)
# Two rocprofv3 --pmc passes, one counter each (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes; then
# scripts/pmc_traffic.py turns them into gpurun_out/pmc_<tag>.json.
set -e
tag=$1; hs=$2; vs=$3; shift 3
root=$PWD
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $root/gpurun_out/pmc_${tag}_$c
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_${tag}_$c -- python3 $root/bench.py --traffic file --in-flight 1 "$@" > $root/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $root
python scripts/pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}.json "$*" $hs $vs
