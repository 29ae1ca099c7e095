"""Compact print of bench.py JSON lines: value, ms/step, p50 / p99, roofline fraction, the self-checks, the side rates."""
import json
import sys

for path in sys.argv[1:]:
    try:
        b = json.loads([l for l in open(path).read().splitlines() if l.startswith('{"metric')][-1])
        print(path.split("/")[-1], round(b["value"], 1), "steps/s", round(b["ms_per_step"], 3), "ms  p50", round(b["p50_latency_ms"], 2), "p99", round(b["p99_latency_ms"], 2),
              "frac %.2e" % b["roofline"]["frac"], "parity", b.get("parity_checked"), b.get("parity_mismatches"), "replay", b.get("replay_mismatches"),
              "host-inclusive", b.get("value_host_inclusive"), "literal", b.get("value_run_optimizer_literal"))
    except Exception as e:  # noqa: BLE001
        print(path, "ERR", e)
