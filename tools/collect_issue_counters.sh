#!/bin/bash
# Run ON THE GPU BOX from the repository root: instruction-issue and instruction-cache counters of the bench kernel.
set -e -o pipefail
TAG=${1:-r01}
KERNEL=${2:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/issue_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/icache -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/icache.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/ifetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/ifetch.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/misc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/misc.log 2>&1 || true
find $OUT -name "*counter_collection.csv" | head
