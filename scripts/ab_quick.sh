#!/bin/bash
# usage: scripts/ab_quick.sh <pytest -k expression or ""> <lib.so> [<lib.so> ...]   (GPU box)
# a parity subset on the in-tree build, then ms/frame of the BASELINE configs: in-tree and each variant library, twice
sel=$1; shift
mkdir -p gpurun_out
if [ -n "$sel" ]; then
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py -m gpu -x -q --timeout 300 --timeout-method thread -k "$sel" > gpurun_out/ab_quick_pytest.log 2>&1
  rc=$?; tail -4 gpurun_out/ab_quick_pytest.log; [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 200 python scripts/ab_configs.py > /dev/null || exit 1
for round in 1 2; do
  timeout -k 10 200 python scripts/ab_configs.py || exit 1
  for lib in "$@"; do echo "== $lib"; RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python scripts/ab_configs.py || exit 1; done
done
