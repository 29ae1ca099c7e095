"""Turns the rocprofv3 output of tools/collect_profiles.sh into the committed summaries under profiles/:
kernel stats (average duration per kernel), HBM traffic per launch (FETCH_SIZE / WRITE_SIZE passes) and the SQ counters.

    python tools/summarize_profiles.py <tag, e.g. r03_c2> <timed launches> [<directory with the passes>]

The profiled command is `python3 bench.py --workload W --steps S --warmup U --no-cpu-baseline --no-host-inclusive
--no-scaling-reference`: its launches are, in order, the recording closed loop, one launch per resident bank (packing), U
warm-up launches, the S timed launches and one more launch per bank (the replay check).  The timed region is therefore the S
launches in front of the last <banks> ones.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_c2"
workload = tag.split("_")[-1]
TIMED = int(sys.argv[2]) if len(sys.argv) > 2 else 200
SRC = os.path.join(ROOT, sys.argv[3]) if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "prof_" + workload)
KERNEL, HELPER = "pdmpc_bulk_kernel", "pdmpc_helper"  # (the helpers are workgroups of the search launch itself: there is no helper kernel any more, the second family stays empty)
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def find(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None


def bench_line(name):
    p = os.path.join(SRC, "bench_%s.json" % name)
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p).read().splitlines() if l.startswith('{"metric')]
    return json.loads(lines[-1]) if lines else None


plain = bench_line("plain")
banks = 20 if workload == "c2" else 8
if plain:
    m = re.search(r"(\d+) recorded closed-loop steps", plain["config"]["workload"])
    banks = int(m.group(1)) if m else banks
TAIL = banks  # the replay check's launches behind the timed region


def timed(vals):
    return vals[len(vals) - TAIL - TIMED : len(vals) - TAIL] if len(vals) >= TIMED + TAIL else vals


args = open(os.path.join(SRC, "args.txt")).read().strip() if os.path.exists(os.path.join(SRC, "args.txt")) else ""
summary = {"tag": tag, "command": "python3 bench.py " + args,
           "passes": "rocprofv3 --kernel-trace --stats; then one --kernel-trace --pmc pass per counter group (FETCH_SIZE; WRITE_SIZE; four SQ groups)",
           "timed_launches": TIMED, "launches_behind_the_timed_region": TAIL}
ks = find("stats/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(DST, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write(open(ks).read())
    summary["kernel_stats"] = {r["Name"]: {k: r[k] for k in r if k != "Name"} for r in rows if KERNEL in r["Name"] or HELPER in r["Name"]}
kt = find("stats/**/*kernel_trace.csv")
if kt:
    for kname, key in ((KERNEL, "kernel_trace"), (HELPER, "helper_kernel_trace")):
        rows = [r for r in csv.DictReader(open(kt)) if kname in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        if not d:
            continue
        t = timed(d)
        summary[key] = {"launches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d), "timed_region_launches": len(t),
                        "timed_region_avg_ms": sum(t) / len(t), "grid": rows[-1].get("Grid_Size"), "workgroup": rows[-1].get("Workgroup_Size"),
                        # (the trace's own LDS_Block_Size / VGPR_Count / SGPR_Count columns are not copied: they describe the code object in the
                        # trace's units — static LDS, allocation granules — and read like evidence they are not; the launch's dynamic LDS and the
                        # compiler's register / scratch numbers are in `resources` below)
                        }


def counters(passname):
    """{counter: {kernel family: average over the timed launches}} of one --pmc pass."""
    cc = find("%s/**/*counter_collection.csv" % passname)
    out = {}
    if not cc:
        return out
    per = {}
    for r in csv.DictReader(open(cc)):
        fam = KERNEL if KERNEL in r["Kernel_Name"] else (HELPER if HELPER in r["Kernel_Name"] else None)
        if fam:
            per.setdefault((r["Counter_Name"], fam), []).append(float(r["Counter_Value"]))
    for (cname, fam), vals in per.items():
        t = timed(vals)
        out.setdefault(cname, {})[fam] = {"per_launch_avg_timed": sum(t) / len(t), "launches_timed": len(t), "launches_all": len(vals)}
    return out


traffic = {}
for name in ("fetch", "write"):
    traffic.update(counters(name))
if traffic:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of WIDE COALESCED
    # STREAMING reads (16 B per lane).  This kernel's reads are node records fetched by all lanes of a wave at once (one 64-byte line
    # per record) and 8-byte-per-lane list scans, not 16 B/lane streams: the raw value is reported, the doubled one as an upper bound.
    def kib(counter, fam):
        return traffic.get(counter, {}).get(fam, {}).get("per_launch_avg_timed", 0.0) * 1024

    fetch, write = kib("FETCH_SIZE", KERNEL), kib("WRITE_SIZE", KERNEL)
    hfetch, hwrite = kib("FETCH_SIZE", HELPER), kib("WRITE_SIZE", HELPER)
    summary["traffic"] = traffic
    summary["search_kernel_bytes_per_launch"] = fetch + write
    summary["helper_kernel_bytes_per_launch"] = hfetch + hwrite
    total = fetch + write + hfetch + hwrite
    summary["hbm_bytes_per_launch"] = total
    summary["hbm_bytes_per_launch_upper_bound"] = 2 * (fetch + hfetch) + write + hwrite
    summary["hbm_bytes_note"] = "FETCH_SIZE*1024 + WRITE_SIZE*1024 of the search kernel and of the helper kernel next to it, averaged over the timed launches; upper bound = FETCH_SIZE doubled (gfx950 wide-read correction, not applicable to this access pattern)"
    if plain:
        alg = plain["roofline"]["algorithmic_bytes_per_launch"]
        summary["algorithmic_bytes_per_launch"] = alg
        summary["traffic_over_algorithmic"] = total / alg if alg else None
    json.dump({"hbm_bytes_per_launch": total, "hbm_bytes_per_launch_upper_bound": summary["hbm_bytes_per_launch_upper_bound"], "source": "%s_summary.json" % tag,
               "launches": TIMED}, open(os.path.join(DST, "%s_pmc_traffic_%s.json" % (tag.rsplit("_", 1)[0], workload)), "w"))
sq = {}
for name in ("sq1", "sq2", "sq3", "sq4"):
    sq.update(counters(name))
if sq:
    summary["sq_counters_per_launch"] = sq

    def v(c, fam=KERNEL):
        return sq.get(c, {}).get(fam, {}).get("per_launch_avg_timed")

    derived = {}
    if v("SQ_WAVE_CYCLES") and v("SQ_ACTIVE_INST_VALU") is not None:
        derived["valu_active_share_of_wave_cycles"] = v("SQ_ACTIVE_INST_VALU") / v("SQ_WAVE_CYCLES")
    if v("SQ_WAVE_CYCLES") and v("SQ_WAIT_ANY") is not None:
        derived["wait_any_share_of_wave_cycles"] = v("SQ_WAIT_ANY") / v("SQ_WAVE_CYCLES")
    if v("SQ_WAVE_CYCLES") and v("SQ_WAIT_INST_ANY") is not None:
        derived["wait_inst_any_share_of_wave_cycles"] = v("SQ_WAIT_INST_ANY") / v("SQ_WAVE_CYCLES")
    if v("SQ_LDS_IDX_ACTIVE") and v("SQ_LDS_BANK_CONFLICT") is not None:
        derived["lds_bank_conflict_share_of_lds_cycles"] = v("SQ_LDS_BANK_CONFLICT") / v("SQ_LDS_IDX_ACTIVE")
    if v("SQ_INSTS_VALU") and v("SQ_INSTS_FLAT") is not None:
        derived["flat_instructions_per_valu_instruction"] = v("SQ_INSTS_FLAT") / v("SQ_INSTS_VALU")
    summary["sq_derived_search_kernel"] = derived
res = os.path.join(DST, "%s_resource_usage.txt" % tag.split("_")[0])
if os.path.exists(res):
    txt = open(res).read()
    m = re.search(r"Function Name: %s\b.*?VGPRs: (\d+)" % KERNEL + r".*?ScratchSize \[bytes/lane\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", txt, re.S)
    if m:
        summary["resources"] = {"source": "profiles/%s_resource_usage.txt (hipcc" % tag.split("_")[0] + "  -Rpass-analysis=kernel-resource-usage)", "vgprs": int(m.group(1)), "scratch_bytes_per_lane": int(m.group(2)),
                                "sgpr_spills": int(m.group(3)), "vgpr_spills": int(m.group(4)),
                                "dynamic_lds_bytes_per_workgroup": plain["roofline"]["lds_bytes_per_workgroup"] if plain else None}
for name in ("plain", "stats"):
    b = bench_line(name)
    if b:
        summary["bench_" + name] = b
json.dump(summary, open(os.path.join(DST, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k not in ("bench_plain", "bench_stats", "traffic", "sq_counters_per_launch")}, indent=1))
