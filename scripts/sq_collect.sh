#!/bin/bash
# usage: scripts/sq_collect.sh <tag>   (on the GPU box, from the repo root)
# Three rocprofv3 --pmc passes of SQ counters (at most 8 per pass, each pass a run of its own, no trace domains), then
# scripts/pmc_sq.py turns them into gpurun_out/<tag>_sq_counters.json.
set -e
tag=$1
root=$PWD
export TMPDIR=/tmp
cd /tmp
p1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
p2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"
p3="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
dirs=""
for p in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  d=$root/gpurun_out/${tag}_sq_$i
  rm -rf $d
  timeout -s KILL 170 rocprofv3 --pmc $p --output-format csv -d $d -o p -- python3 $root/bench.py --traffic file --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-c5 > $d.log 2>&1
  dirs="$dirs $(dirname $(find $d -name 'p_counter_collection.csv' | head -1))"
done
cd $root
python scripts/pmc_sq.py gpurun_out/${tag}_sq_counters.json $dirs
