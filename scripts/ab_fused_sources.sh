#!/bin/bash
# usage: scripts/ab_fused_sources.sh   (GPU box): ms/frame of the BASELINE configs with k_trace making the rays of levels
# 1.. itself (spawn_ray), with k_spawn launches instead (RT_AMD_FUSE_SPAWN=0), and with ab_prev/'s build if there is one;
# twice, after one discarded warm-up process
timeout -k 10 200 python scripts/ab_configs.py > /dev/null || exit 1
for round in 1 2; do
  echo "== k_trace spawns"; timeout -k 10 200 python scripts/ab_configs.py || exit 1
  echo "== k_spawn launches"; RT_AMD_FUSE_SPAWN=0 timeout -k 10 200 python scripts/ab_configs.py || exit 1
  if [ -f ab_prev/libraytracer_amd.so ]; then
    echo "== previous build"; RT_AMD_LIB=$PWD/ab_prev/libraytracer_amd.so timeout -k 10 200 python scripts/ab_configs.py || exit 1
  fi
done
