#!/bin/bash
# A/B of the headline kernel on one box: the shipped build, then a rebuild with TINYMPC_T16_FLAGS="$1", each timed with tools/pi_time.py-like loop
set -e
mkdir -p gpurun_out
python tools/t16_time.py "shipped" >> gpurun_out/ab_t16.txt
TINYMPC_T16_FLAGS="$1" python -c "
import accelerated_tinympc_amd as T
T.build.build(force=False)" >/dev/null 2>&1 || true
touch accelerated-tinympc_amd/csrc/admm_tile16.hip
TINYMPC_T16_FLAGS="$1" python -c "
import accelerated_tinympc_amd as T
T.build.build()"
TINYMPC_T16_FLAGS="$1" python tools/t16_time.py "$1" >> gpurun_out/ab_t16.txt
