#!/bin/bash
# Run HERE (the build container), from the repository root, after any change under accelerated-tinympc_amd/csrc/:
#   bash tools/refresh_profiles.sh r02
# re-collects the rocprofv3 kernel stats, the FETCH_SIZE / WRITE_SIZE passes (with the hbm_calib correction) and the issue
# counters of the default bench run on an MI355X through gpurun, then rewrites profiles/<tag>_* and profiles/hbm_traffic.json,
# whose entries are bound to the hash of the kernel sources (bench.py prints roofline.traffic only while that hash matches).
set -e -o pipefail
TAG=${1:-r02}
test -x build/hbm_calib || hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_calib.hip -o build/hbm_calib
python -c "import __graft_entry__ as g; g.build()"
rm -rf gpurun_out/prof_$TAG gpurun_out/issue_$TAG
/usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/collect_profiles.sh $TAG 0 > gpurun_out/collect_$TAG.log 2>&1; tail -1 gpurun_out/collect_$TAG.log; bash tools/collect_issue_counters.sh $TAG 0 > gpurun_out/collect_issue_$TAG.log 2>&1; tail -1 gpurun_out/collect_issue_$TAG.log"
python tools/summarize_profiles.py $TAG > /dev/null
python - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
t = json.load(open('profiles/hbm_traffic.json'))
print('kernel sources', bench.kernel_source_sha())
for k, v in t.items():
    print(f"  {k}: {v['bytes'] / 1e6:.1f} MB per launch, bound to {v['csrc_sha']}")
PY
