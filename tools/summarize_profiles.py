"""Turns the rocprofv3 output of tools/collect_profiles.sh (merged into gpurun_out/prof) into the committed summaries
under profiles/: kernel stats (average duration per kernel) and HBM traffic per launch of the search kernel."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def find(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None  # (gpurun merges new output into the old directory: newest wins)


summary = {"tag": tag, "command": "python bench.py --steps 200 --warmup 20 --no-cpu-baseline (rocprofv3 --kernel-trace --stats; separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes)"}
ks = find("stats/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(DST, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write(open(ks).read())
    for r in rows:
        if "pdmpc_search" in r["Name"]:
            summary["kernel_stats"] = {k: r[k] for k in r}
kt = find("stats/**/*kernel_trace.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "pdmpc_search" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    if d:
        timed = d[-200:] if len(d) >= 200 else d
        summary["kernel_trace"] = {"launches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
                                   "timed_region_launches": len(timed), "timed_region_avg_ms": sum(timed) / len(timed),
                                   "lds_block_size": rows[0].get("LDS_Block_Size"), "vgpr": rows[0].get("VGPR_Count"), "sgpr": rows[0].get("SGPR_Count"),
                                   "grid": rows[0].get("Grid_Size"), "workgroup": rows[0].get("Workgroup_Size")}
traffic = {}
for name in ("fetch", "write"):
    cc = find("%s/**/*counter_collection.csv" % name)
    if not cc:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(cc)) if "pdmpc_search" in r["Kernel_Name"]]
    if vals:
        traffic[name] = {"counter": name.upper() + "_SIZE", "unit": "KiB as reported", "per_launch_avg": sum(vals) / len(vals), "launches": len(vals)}
if traffic:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
    # coalesced reads -> doubled (upper bound for this kernel, whose reads are mostly 16 B/lane copies and 64 B records)
    fetch = traffic.get("fetch", {}).get("per_launch_avg", 0.0) * 1024 * 2
    write = traffic.get("write", {}).get("per_launch_avg", 0.0) * 1024
    summary["traffic"] = traffic
    summary["hbm_bytes_per_launch"] = fetch + write
    summary["hbm_bytes_note"] = "FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024, averaged over the launches of the profiled bench run"
    json.dump({"hbm_bytes_per_launch": fetch + write, "source": "%s_summary.json" % tag}, open(os.path.join(DST, "pmc_traffic.json"), "w"))
for name in ("plain", "stats"):
    p = os.path.join(SRC, "bench_%s.json" % name)
    if os.path.exists(p):
        try:
            summary["bench_" + name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            summary["bench_" + name] = "unparsed: %s" % e
json.dump(summary, open(os.path.join(DST, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k not in ("bench_plain", "bench_stats")}, indent=1))
