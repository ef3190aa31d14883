#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
   <tag>_<kernel>_kernel_stats.csv, <tag>_<kernel>_pmc.json and the hbm_traffic.json that bench.py reads."""
import csv, glob, json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}"
prof = ROOT / "profiles"


def counter_rows(d):
    # gpurun merges every call's files into the same directory: only the newest collection of a pass counts
    files = sorted(glob.glob(str(src / d / "**" / "*counter_collection.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    return list(csv.DictReader(open(files[-1]))) if files else []


def per_kernel(rows, counter):
    acc = {}
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return acc


out = {}
CALIB_BYTES = float(1 << 30)
factors = {}
for c, kname in (("FETCH_SIZE", "calib_read"), ("WRITE_SIZE", "calib_write")):
    cal = per_kernel(counter_rows(f"calib_{c}"), c)
    vals = [v for k, vs in cal.items() if kname in k for v in vs]
    factors[c] = CALIB_BYTES / (sum(vals) / len(vals)) if vals else None
out["calibration"] = {"bytes_per_launch": CALIB_BYTES, "bytes_per_count": factors,
                      "note": "bytes per counter unit measured on tools/micro/hbm_calib (dword per lane, 256 B per wave-instruction)"}
kern = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, vs in per_kernel(counter_rows(f"pmc_{c}"), c).items():
        if "admm_" in k:
            kern.setdefault(k, {})[c] = sum(vs) / len(vs)
for k, d in kern.items():
    rd = d.get("FETCH_SIZE", 0) * (factors["FETCH_SIZE"] or 0)
    wr = d.get("WRITE_SIZE", 0) * (factors["WRITE_SIZE"] or 0)
    d.update(read_bytes=rd, write_bytes=wr, hbm_bytes=rd + wr)
out["kernels"] = kern
sq = {}
for r in counter_rows("pmc_sq"):
    if "admm_" in r["Kernel_Name"]:
        sq.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out["sq"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in sq.items()}
line = None
for l in open(src / "bench_stats.log"):
    if l.startswith("{"):
        line = json.loads(l)
out["bench_line_under_profiler"] = line
kshort = (line["roofline"]["kernel"] if line else "kernel").replace("<", "_").replace(">", "").replace(",", "_")
(prof / f"{tag}_{kshort}_pmc.json").write_text(json.dumps(out, indent=1))
stats = sorted(glob.glob(str(src / "stats" / "**" / "*kernel_stats.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
if stats:
    (prof / f"{tag}_{kshort}_kernel_stats.csv").write_text(open(stats[-1]).read())
# the table bench.py reads: "<kernel name>:<mode>:<batch>" -> HBM bytes per launch
tf = prof / "hbm_traffic.json"
table = json.loads(tf.read_text()) if tf.exists() else {}
table = {k: v for k, v in table.items() if not k.startswith("void ")}
def label(kname):
    import re
    # template arguments: NX, NU, N, EXACT, H16, MPC (closed loop on chip), BPI (per-instance bounds), D32 (fp32 duals)
    m = re.search(r"admm_rowlane_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)((?:, (?:true|false))*)>", kname)
    if m:
        rest = [t.strip() == "true" for t in m.group(6).split(",") if t.strip()] + [False] * 3
        if not rest[0] and not rest[1]:   # the closed-loop and per-instance-bounds instantiations are different workloads
            sto = (",h16d" if rest[2] else ",h16") if m.group(5) == "true" else ""
            return f"rowlane<{m.group(1)},{m.group(2)},{m.group(3)},{'exact' if m.group(4) == 'true' else 'fast'}{sto}>"
    m = re.search(r"admm_tile16_kernel<(\d+), (true|false), (true|false)((?:, (?:true|false))*)>", kname)   # N, EXACT, COLD[, MPC, BR, XR] (nx = 12, nu = 4)
    if m:
        rest = [t.strip() == "true" for t in m.group(4).split(",") if t.strip()] + [False] * 3
        if not rest[0]:   # the closed-loop instantiation is a different workload
            return f"tile16<12,4,{m.group(1)},{'exact' if m.group(2) == 'true' else 'fast'}{',pi' if rest[1] or rest[2] else ''}>"
    m = re.search(r"admm_stream_kernel<(\d+), (\d+)>", kname)
    if m:
        return f"stream<{m.group(1)},{m.group(2)}>"
    m = re.search(r"admm_(wave\w*)_kernel<(\d+), (\d+)(?:, (true|false))?", kname)
    if m:
        return f"{'waveres' if m.group(1) == 'waveres' else 'wavestream'}<{m.group(2)},{m.group(3)},{'fast' if m.group(4) == 'false' else 'exact'}>"
    return kname
if line:
    sys.path.insert(0, str(ROOT))
    from bench import kernel_source_sha  # binds the figure to the kernel sources it was measured on (bench.py reports null once they change)
    for k, d in kern.items():
        table[f"{label(k)}:early_exit:{line['config']['instances_per_gpu']}"] = {"bytes": d["hbm_bytes"], "isa_sha": __import__("accelerated_tinympc_amd").build.kernel_isa_sha(label(k)), "csrc_sha": kernel_source_sha(), "profile": tag}
tf.write_text(json.dumps(table, indent=1))
print(json.dumps({k: v for k, v in out.items() if k != "bench_line_under_profiler"}, indent=1)[:3000])


# ---- instruction-issue counters (tools/collect_issue_counters.sh -> gpurun_out/issue_<tag>/) -> profiles/<tag>_rowlane_issue_counters.json
import collections, csv as _csv, glob as _glob
issue = {}
for f in _glob.glob(str(ROOT / "gpurun_out" / f"issue_{tag}" / "*" / "*" / "*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in _csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "admm_" not in k: continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in per:
        issue.setdefault(k, {}).update({c: v / len(cnt[k]) for c, v in per[k].items()})
if issue:
    dst = prof / f"{tag}_rowlane_issue_counters.json"
    note = json.loads(dst.read_text()).get("note", "") if dst.exists() else ""
    res = {"note": note or "rocprofv3 --pmc passes of tools/collect_issue_counters.sh over the default bench run; averages per launch; "
                          "SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 shader clocks", "kernels": {}}
    for k, d in issue.items():
        dd = dict(d)
        if all(c in d for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS")):
            waves = 16384.0  # 65 536 instances, four per wave
            dd["derived"] = {"valu_per_wave": d["SQ_INSTS_VALU"] / waves, "salu_per_wave": d["SQ_INSTS_SALU"] / waves, "lds_per_wave": d["SQ_INSTS_LDS"] / waves,
                             "clocks_per_valu_per_simd": 4 * d["SQ_WAVE_CYCLES"] / 2 / d["SQ_INSTS_VALU"],
                             "wait_inst_any_frac_of_wave_cycles": d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]}
            if d.get("SQC_ICACHE_REQ"): dd["derived"]["icache_miss_rate"] = d.get("SQC_ICACHE_MISSES", 0.0) / d["SQC_ICACHE_REQ"]
        res["kernels"][k] = dd
    dst.write_text(json.dumps(res, indent=1))
