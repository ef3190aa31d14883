#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root.  Collects, for the bench workload:
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel average duration
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -> HBM-side bytes, each in its own pass (TCC slots), plus the same two
#      passes over tools/micro/hbm_calib (known 1 GiB per launch, same dword-per-lane access shape) for the correction
# Results land in gpurun_out/prof_<tag>/ ; tools/summarize_profiles.py turns them into profiles/*.csv|json.
set -e -o pipefail
TAG=${1:-r01}
KERNEL=${2:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/bench_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/bench_$C.log 2>&1
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- $ROOT/build/hbm_calib > $OUT/calib_$C.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-closed-loop --kernel $KERNEL > $OUT/bench_sq.log 2>&1 || true
grep '^{' $OUT/bench_stats.log | tail -1 | cut -c1-400
find $OUT -name "*.csv" | head -30
