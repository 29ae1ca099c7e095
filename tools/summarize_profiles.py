"""Turns the rocprofv3 output of tools/collect_profiles.sh (merged into gpurun_out/prof) into the committed summaries
under profiles/: kernel stats (average duration per kernel) and HBM traffic per launch of the search kernel."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNEL = "pdmpc_frontier"  # the kernel the default bench launches (search_kernel.hip's pdmpc_search_* only with PDMPC_KERNEL=serial)
TIMED = int(sys.argv[2]) if len(sys.argv) > 2 else 200  # launches of the timed region ...
TAIL = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # ... followed by this many launches (the native closed loop of `host_inclusive`, steps + skip)
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
        if KERNEL in r["Name"]:
            summary["kernel_stats"] = {k: r[k] for k in r}
kt = find("stats/**/*kernel_trace.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    if d:
        timed = d[len(d) - TAIL - TIMED : len(d) - TAIL] if len(d) >= TIMED + TAIL else d
        summary["kernel_trace"] = {"launches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
                                   "timed_region_launches": len(timed), "timed_region_avg_ms": sum(timed) / len(timed),
                                   "lds_block_size": rows[0].get("LDS_Block_Size"), "vgpr": rows[0].get("VGPR_Count"), "sgpr": rows[0].get("SGPR_Count"),
                                   "grid": rows[0].get("Grid_Size"), "workgroup": rows[0].get("Workgroup_Size")}
traffic = {}
helper_traffic = {}
HELPER = "pdmpc_helper"
for name in ("fetch", "write"):
    cc = find("%s/**/*counter_collection.csv" % name)
    if not cc:
        continue
    rows_cc = list(csv.DictReader(open(cc)))
    vals = [float(r["Counter_Value"]) for r in rows_cc if KERNEL in r["Kernel_Name"]]
    hvals = [float(r["Counter_Value"]) for r in rows_cc if HELPER in r["Kernel_Name"]]  # the helper kernel that runs next to every search launch
    if hvals:
        ht = hvals[len(hvals) - TAIL - TIMED : len(hvals) - TAIL] if len(hvals) >= TIMED + TAIL else hvals
        helper_traffic[name] = sum(ht) / len(ht)
    if vals:
        timed = vals[len(vals) - TAIL - TIMED : len(vals) - TAIL] if len(vals) >= TIMED + TAIL else vals  # the timed launches only (recording and warm-up launches come first)
        traffic[name] = {"counter": name.upper() + "_SIZE", "unit": "KiB as reported", "per_launch_avg_timed": sum(timed) / len(timed), "launches_timed": len(timed),
                         "per_launch_avg_all": sum(vals) / len(vals), "launches_all": len(vals)}
if traffic:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of WIDE
    # COALESCED STREAMING reads (16 B per lane).  This kernel's reads are node records fetched by all lanes of a wave at once
    # (one 64-byte line per record) and 8-byte-per-lane list scans, not 16 B/lane streams, so the raw value is reported and the
    # doubled one only as an upper bound.
    fetch = traffic.get("fetch", {}).get("per_launch_avg_timed", 0.0) * 1024
    write = traffic.get("write", {}).get("per_launch_avg_timed", 0.0) * 1024
    hfetch = helper_traffic.get("fetch", 0.0) * 1024
    hwrite = helper_traffic.get("write", 0.0) * 1024
    summary["traffic"] = traffic
    summary["search_kernel_bytes_per_launch"] = fetch + write
    summary["helper_kernel_bytes_per_launch"] = hfetch + hwrite
    summary["helper_kernel_fetch_write_KiB"] = [helper_traffic.get("fetch", 0.0), helper_traffic.get("write", 0.0)]
    fetch += hfetch  # both kernels of a step: the helpers read records and soups and write children, verdicts
    write += hwrite
    summary["hbm_bytes_per_launch"] = fetch + write
    summary["hbm_bytes_per_launch_upper_bound"] = 2 * fetch + write
    summary["hbm_bytes_note"] = "FETCH_SIZE*1024 + WRITE_SIZE*1024 averaged over the timed launches; upper bound = FETCH_SIZE doubled (gfx950 wide-read correction, not applicable to this access pattern)"
    json.dump({"hbm_bytes_per_launch": fetch + write, "hbm_bytes_per_launch_upper_bound": 2 * fetch + write, "source": "%s_summary.json" % tag, "launches": traffic.get("fetch", traffic.get("write"))["launches_timed"]},
              open(os.path.join(DST, "%s_pmc_traffic.json" % tag), "w"))
for name in ("plain", "stats"):
    p = os.path.join(SRC, "bench_%s.json" % name)
    if os.path.exists(p):
        try:
            summary["bench_" + name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            summary["bench_" + name] = "unparsed: %s" % e
json.dump(summary, open(os.path.join(DST, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k not in ("bench_plain", "bench_stats")}, indent=1))
