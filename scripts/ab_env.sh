#!/bin/bash
# usage: scripts/ab_env.sh "<VAR=val ...>" ...   (GPU box): ms/frame of the BASELINE configs once per environment given
# (an argument may name RT_AMD_LIB=<path> too), two rounds, after one discarded warm-up process
timeout -k 10 200 python scripts/ab_configs.py > /dev/null || exit 1
for round in 1 2; do
  for e in "$@"; do echo "== $e"; env $e timeout -k 10 200 python scripts/ab_configs.py || exit 1; done
done
