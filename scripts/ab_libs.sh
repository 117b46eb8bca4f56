#!/bin/bash
# usage: scripts/ab_libs.sh <lib.so> [<lib.so> ...]   (GPU box): ms/frame of the BASELINE configs, in-tree build and each
# variant library in turn, twice, after one discarded warm-up process (the first process on a fresh box runs ~3 % faster)
timeout -k 10 200 python scripts/ab_configs.py > /dev/null || exit 1
for round in 1 2; do
  timeout -k 10 200 python scripts/ab_configs.py || exit 1
  for lib in "$@"; do echo "== $lib"; RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python scripts/ab_configs.py || exit 1; done
done
