set -e
tag=r03d
root=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf $root/gpurun_out/${tag}_prof
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_prof -- python3 $root/bench.py --no-cpu-baseline --no-c5 --traffic file --in-flight 1 > $root/gpurun_out/${tag}_bench_under_rocprof.json 2> $root/gpurun_out/${tag}_prof.err
cd $root
find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
python scripts/launch_durations.py gpurun_out/${tag}_prof gpurun_out/${tag}_bench_under_rocprof.json gpurun_out/${tag}_launch_durations.json | tail -30
