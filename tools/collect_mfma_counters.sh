#!/bin/bash
# Run ON THE GPU BOX from the repository root: MFMA utilisation of the streaming kernel (bench workload, --kernel 1).
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --kernel 1 > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --kernel 1 > $OUT/bench_stats.log 2>&1
find $OUT -name "*.csv" | head
