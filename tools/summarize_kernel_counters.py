#!/usr/bin/env python3
"""gpurun_out/kc_<tag>/<workload>/ (tools/collect_kernel_counters.sh) -> profiles/<tag>_<workload>_counters.json.

One JSON per workload: rocprofv3 kernel-stats average duration of the solve kernel, every SQ counter averaged per launch,
FETCH_SIZE / WRITE_SIZE converted to bytes with the factor measured on tools/micro/hbm_calib in the same call
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half the bytes of a coalesced read on gfx950 — the calibration run
measures the factor instead of assuming it), derived figures (clocks per VALU instruction and SIMD, stall fractions, LDS
share, roofline fraction) and the hash of the kernel sources the figures were taken on.

    python tools/summarize_kernel_counters.py <tag> [<workload> ...]
"""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import kernel_source_sha  # noqa: E402

tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"kc_{tag}"
prof = ROOT / "profiles"
CALIB_BYTES = float(1 << 30)
SOLVE_KERNELS = ("admm_", )


def newest(pattern):
    files = sorted(glob.glob(str(pattern), recursive=True), key=lambda f: Path(f).stat().st_mtime)
    return files[-1] if files else None


def counters(d):
    f = newest(src / d / "**" / "*counter_collection.csv")
    if not f:
        return {}, {}
    per, disp, meta = {}, {}, {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        per.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp.setdefault(k, set()).add(r["Dispatch_Id"])
        meta[k] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds_bytes=int(r["LDS_Block_Size"]),
                       scratch_bytes_per_lane=int(r["Scratch_Size"]), workgroup=int(r["Workgroup_Size"]), grid=int(r["Grid_Size"]))
    return {k: {c: v / len(disp[k]) for c, v in d.items()} for k, d in per.items()}, meta


def calib_factor(c, kname):
    per, _ = counters(f"calib/{c}")
    vals = [d[c] for k, d in per.items() if kname in k and c in d]
    return CALIB_BYTES / (sum(vals) / len(vals)) if vals else None


factors = {"FETCH_SIZE": calib_factor("FETCH_SIZE", "calib_read"), "WRITE_SIZE": calib_factor("WRITE_SIZE", "calib_write")}
workloads = sys.argv[2:] or sorted(p.name for p in src.iterdir() if p.is_dir() and p.name != "calib")
for w in workloads:
    line = None
    for l in open(src / w / "stats.log"):
        if l.startswith("{"):
            line = json.loads(l)
    out = {"workload": w, "tag": tag, "csrc_sha": kernel_source_sha(), "run": line,
           "how": "tools/collect_kernel_counters.sh: one rocprofv3 process per pass (kernel-trace stats; three SQ passes; FETCH_SIZE; WRITE_SIZE); "
                  "per-launch averages; SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (x4 = shader clocks)"}
    f = newest(src / w / "stats" / "**" / "*kernel_stats.csv")
    kname = None
    if f:
        for r in csv.DictReader(open(f)):
            if any(s in r["Name"] for s in SOLVE_KERNELS) and "dispatch_key" not in r["Name"]:
                kname = r["Name"]
                out["kernel_stats"] = dict(name=r["Name"], calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6,
                                           max_ms=float(r["MaxNs"]) / 1e6)
                break
        (prof / f"{tag}_{w}_kernel_stats.csv").write_text(open(f).read())
    sq, meta = {}, {}
    for p in ("sq1", "sq2", "sq3"):
        per, m = counters(f"{w}/{p}")
        for k, d in per.items():
            if kname and k == kname:
                sq.update(d)
                meta = m[k]
    out["launch"] = meta
    out["sq"] = sq
    hbm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per, _ = counters(f"{w}/{c}")
        for k, d in per.items():
            if kname and k == kname and c in d:
                hbm[c] = d[c]
                hbm[c + "_bytes"] = d[c] * factors[c] if factors[c] else None
    if hbm:
        hbm["bytes_per_count"] = factors
        if hbm.get("FETCH_SIZE_bytes") is not None and hbm.get("WRITE_SIZE_bytes") is not None:
            hbm["hbm_bytes_per_launch"] = hbm["FETCH_SIZE_bytes"] + hbm["WRITE_SIZE_bytes"]
    out["hbm"] = hbm
    d = {}
    if sq.get("SQ_WAVES") and sq.get("SQ_WAVE_CYCLES"):
        waves = sq["SQ_WAVES"]
        d["waves"] = waves
        d["shader_clocks_per_wave"] = 4 * sq["SQ_WAVE_CYCLES"] / waves
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM"):
            if c in sq:
                d[c.lower().replace("sq_insts_", "") + "_insts_per_wave"] = sq[c] / waves
        if sq.get("SQ_INSTS_VALU"):
            d["wave_clocks_per_valu_inst"] = 4 * sq["SQ_WAVE_CYCLES"] / sq["SQ_INSTS_VALU"]
        for c, n in (("SQ_ACTIVE_INST_VALU", "valu_issue"), ("SQ_ACTIVE_INST_ANY", "any_issue"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "parked_waitcnt_or_barrier"),
                     ("SQ_WAIT_INST_LDS", "lds_issue_stall"), ("SQ_ACTIVE_INST_LDS", "lds_issue"), ("SQ_ACTIVE_INST_VMEM", "vmem_issue"), ("SQ_INST_CYCLES_VMEM", "vmem_inst_cycles"),
                     ("SQ_ACTIVE_INST_SCA", "scalar_issue")):
            if c in sq:
                d[n + "_frac_of_wave_cycles"] = sq[c] / sq["SQ_WAVE_CYCLES"]
        if sq.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and sq.get("SQ_BUSY_CYCLES"):
            d["mfma_busy_cycles"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"]
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"]
    if line and out.get("kernel_stats"):
        ms = out["kernel_stats"]["avg_ms"]
        d["roof_frac_from_rocprof_avg"] = line["alg_flops_per_launch"] / (ms * 1e-3) / 1e12 / line["roof_tflops"]
        d["alg_hbm_GBs"] = line["alg_bytes_per_launch"] / (ms * 1e-3) / 1e9
        if hbm.get("hbm_bytes_per_launch"):
            d["counter_hbm_GBs"] = hbm["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
            d["traffic_over_algorithmic_bytes"] = hbm["hbm_bytes_per_launch"] / line["alg_bytes_per_launch"]
    out["derived"] = d
    (prof / f"{tag}_{w}_counters.json").write_text(json.dumps(out, indent=1))
    print(w, json.dumps(d, indent=1))
    # the table bench.py reads for roofline.traffic: "<kernel name>:<mode>:<batch>" -> HBM bytes per launch, bound to the kernel sources
    if line and hbm.get("hbm_bytes_per_launch"):
        tf = prof / "hbm_traffic.json"
        table = json.loads(tf.read_text()) if tf.exists() else {}
        import accelerated_tinympc_amd as _T  # the figure is bound to the kernel's DEVICE CODE (build.kernel_isa_sha): a comment edit does not make it stale
        table[f"{line['kernel']}:early_exit:{line['batch']}"] = {"bytes": hbm["hbm_bytes_per_launch"], "isa_sha": _T.build.kernel_isa_sha(line["kernel"]),
                                                                   "csrc_sha": out["csrc_sha"], "profile": f"{tag}_{w}_counters.json"}
        tf.write_text(json.dumps(table, indent=1))
