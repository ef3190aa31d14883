#!/bin/bash
# Run HERE (the build container), from the repository root: everything the round-end driver runs, plus the differential
# drivers, in the order "cheap first".  GPU steps go through gpurun (one box each); logs land in gpurun_out/.
#   bash tools/run_all_checks.sh [fuzz-seconds-per-driver, default 60]
set -e -o pipefail
FUZZ=${1:-60}
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests/ -x -q -m "not gpu"
/usr/local/graft/bin/gpurun --timeout 1200 -- "timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/checks_gputest.log 2>&1; tail -2 gpurun_out/checks_gputest.log; \
  timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1"
/usr/local/graft/bin/gpurun --timeout 1200 -- "for f in fuzz_parity fuzz_parity64 fuzz_api fuzz_mpc fuzz_native fuzz_fast_families fuzz_stream_consistency fuzz_dispatch; do \
  timeout -k 10 $((FUZZ + 90)) python tests/fuzz/\$f.py $FUZZ 1 > gpurun_out/checks_\$f.log 2>&1; echo \"\$f: \$(tail -1 gpurun_out/checks_\$f.log | cut -c1-160)\"; done"
/usr/local/graft/bin/gpurun --timeout 900 -- "timeout -k 10 400 python bench.py > gpurun_out/checks_bench.log 2>&1; tail -1 gpurun_out/checks_bench.log | cut -c1-400; \
  timeout -k 10 400 python bench.py --config random32 > gpurun_out/checks_bench_random32.log 2>&1; tail -1 gpurun_out/checks_bench_random32.log | cut -c1-300"
echo "all checks ran; after a change under accelerated-tinympc_amd/csrc/ re-bind the committed counter figures to the new sources:"
echo "  gpurun -- 'bash tools/collect_profiles.sh r03 0' && python tools/summarize_profiles.py r03        # the bench run itself (roofline.traffic)"
echo "  gpurun -- 'bash tools/collect_kernel_counters.sh r03 <workloads of tools/prof_workload.py>' && python tools/summarize_kernel_counters.py r03"
