#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root:
#   bash tools/collect_kernel_counters.sh <tag> <workload> [<workload> ...]
# For every workload of tools/prof_workload.py: one rocprofv3 --kernel-trace --stats pass (average launch duration), two SQ
# counter passes (8 slots each), one FETCH_SIZE and one WRITE_SIZE pass (TCC slots do not fit together), each pass a
# process of its own as MI355X_MICROARCH.md prescribes.  tools/summarize_kernel_counters.py turns gpurun_out/kc_<tag>/ into
# profiles/<tag>_<workload>_counters.json.
set -e -o pipefail
TAG=$1; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
test -x $ROOT/build/hbm_calib && for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $ROOT/gpurun_out/kc_$TAG/calib
  rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/kc_$TAG/calib/$C -- $ROOT/build/hbm_calib > $ROOT/gpurun_out/kc_$TAG/calib/$C.log 2>&1 || true
done
for W in "$@"; do
  OUT=$ROOT/gpurun_out/kc_$TAG/$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/prof_workload.py $W --launches 5 > $OUT/stats.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 $ROOT/tools/prof_workload.py $W --launches 2 > $OUT/sq1.log 2>&1 || true
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/prof_workload.py $W --launches 2 > $OUT/sq2.log 2>&1 || true
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/sq3 -- python3 $ROOT/tools/prof_workload.py $W --launches 2 > $OUT/sq3.log 2>&1 || true
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $ROOT/tools/prof_workload.py $W --launches 2 > $OUT/$C.log 2>&1 || true
  done
  echo "$W: $(grep '^{' $OUT/stats.log | tail -1 | cut -c1-300)"
done
