"""Where a time step's time goes in the frontier kernel: per vehicle rounds, nodes processed vs popped, and the 100 MHz
tick counters of the round phases and of the helper workgroups (PDMPC_TUNING=debug_tail=1, set here); PROFILE_CHAIN=1: when every
vehicle's areas went out; PROFILE_ROUNDS=1: the round sizes of the heaviest search; PROFILE_TOP=n: the n slowest vehicles per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
os.environ["PDMPC_TUNING"] = ",".join(x for x in (os.environ.get("PDMPC_TUNING", ""), "debug_tail=2" if os.environ.get("PROFILE_SEATS") else ("debug_tail=3" if os.environ.get("PROFILE_PASSES") else "debug_tail=1")) if x)
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 20; args.hp = 8; args.mpa = "single_speed"; args.instances = 1; args.workload = "c2"; args.max_nodes = 1 << 17; args.seed = 1; args.max_levels = 99; args.priorities = "constant"
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    args.vehicles = 128; args.workload = "c3"; args.max_levels = 2; args.priorities = "coloring"; args.max_nodes = 1 << 16
if len(sys.argv) > 1 and sys.argv[1] == "c4":
    args.vehicles = 512; args.workload = "c4"; args.hp = 10; args.priorities = "coloring"; args.max_nodes = 1 << 16
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    args.workload = "c5"; args.instances = 64; args.max_nodes = 1 << 15
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
probs = bench.record_steps(options, mpa, ctl, opt, 20 if args.workload in ("c2", "c5") else 4, 6 if args.workload not in ("c4", "c5") else 3, explore_instances=64 if args.workload == "c5" else 0)
for b, prob in enumerate(probs):
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb, weights=None if os.environ.get("PROFILE_NO_WEIGHTS") else prob.get("prev_pops"))  # (slots by priority on the work of the step before, as bench.py packs its banks)
    h.reset_stats()
    h.launch(); recs = h.fetch(len(prob["iters"])); st = h.stats()
    print("step", b, "kernel ms %.3f" % st["kernel_ms"], "levels", len(prob["level_sizes"]))
    dc = h.debug_counters()
    if dc[13]:
        print("   helpers: %d tiles, per tile: claim -> soup %.1f us, records %.1f us, checks %.1f us, verdicts + report %.1f us; idle in all %.0f us; shared rounds %d; lifetimes in all %.0f us" % (
            dc[13], dc[9] / 100.0 / dc[13], dc[10] / 100.0 / dc[13], dc[11] / 100.0 / dc[13], dc[12] / 100.0 / dc[13], dc[8] / 100.0, dc[4], dc[7] / 100.0))
    if dc[15]:
        print("   arrivals: %d of %d (step, predecessor) areas differ from the expected ones" % (dc[14], dc[15]))
    rows = []
    for v in range(len(recs)):
        t = np.asarray(recs[v]["path_nodes"])
        rows.append((t[16][7] / 100.0, v, prob["levels"][v], int(recs[v]["n_popped"]), int(t[16][1]), int(t[16][2]), int(t[16][0]), t[15][0] / 100.0, t[15][1] / 100.0, t[15][2] / 100.0, t[15][3] / 100.0, len(prob["preds"][v]),
                     t[14][0] / 100.0, t[14][1] / 100.0, t[14][2] / 100.0, t[14][3] / 100.0, t[14][4] / 100.0, t[14][5] / 100.0))
    if os.environ.get("PROFILE_ROUNDS"):  # the sizes of the first thirty-two rounds of the step's largest search (bulk kernel; only while Hp <= 8)
        v = max(range(len(recs)), key=lambda i: np.asarray(recs[i]["path_nodes"])[16][1])
        t = np.asarray(recs[v]["path_nodes"])
        print("   round sizes of veh %d:" % v, [int(x) for x in t[9:13].reshape(-1) if x > 0])
    if os.environ.get("PROFILE_CHAIN"):  # when every vehicle's areas went out (done flag) and ended, on the device's 100 MHz clock from the first workgroup's start; hop = after its last predecessor's flag
        T = [np.asarray(r["path_nodes"]) for r in recs]
        origin = min(t[15][7] for t in T)
        pub = [((t[15][5] if t[15][5] > 0 else t[15][6]) - origin) / 100.0 for t in T]
        end = [(t[15][6] - origin) / 100.0 for t in T]
        for v in sorted(range(len(recs)), key=lambda i: (prob["levels"][i], i)):
            pr = prob["preds"][v]
            last = max([pub[q] for q in pr], default=0.0)
            la = (T[v][14][7] - origin) / 100.0 if T[v][14][7] > 0 else 0.0
            print("   chain veh %2d level %2d preds %2d | busy %4.0f us | areas out %4.0f us (hop %4.0f) | end %4.0f us | last arrival into the running search %4.0f us at round %2d of %2d | %2d verifications %3.0f us (copy %.0f re-check %.0f parked %.0f candidates %.0f record+flag %.0f; of the re-check: gathering %.0f)" % (
                v, prob["levels"][v], len(pr), T[v][14][0] / 100.0 + T[v][15][0] / 100.0 + T[v][15][2] / 100.0, pub[v], pub[v] - last, end[v], la, int(T[v][14][6]), int(T[v][16][0]), int(T[v][16][6]), T[v][15][1] / 100.0, T[v][13][0] / 100.0, T[v][13][1] / 100.0, T[v][13][2] / 100.0, T[v][13][3] / 100.0, T[v][13][4] / 100.0, T[v][13][5] / 100.0))
    if os.environ.get("PROFILE_SEATS"):  # the P1 passes of the step's largest search by the number of seated helpers
        v = max(range(len(recs)), key=lambda i: np.asarray(recs[i]["path_nodes"])[16][1])
        t = np.asarray(recs[v]["path_nodes"])
        for bkt, name in enumerate(("not shared", "1-7 seats", "8-31 seats", "32-64 seats")):
            n = t[11][bkt]
            if n:
                print("   veh %d P1 %-11s: %4d rounds, %6.1f us each, %6.0f entries of which the owner checks %5.0f" % (v, name, n, t[11][4 + bkt] / 100.0 / n, t[12][bkt] / n, t[12][4 + bkt] / n))
    if os.environ.get("PROFILE_SEATS"):
        held = sorted(((int(np.asarray(r["path_nodes"])[13][7]), v) for v, r in enumerate(recs)), reverse=True)
        print("   seats given out: %d searches, %d seats; the largest holders (seats: vehicle, nodes processed, us waiting): %s" % (
            sum(1 for s_, _ in held if s_), sum(s_ for s_, _ in held),
            ", ".join("%d: veh %d %d %.0f" % (s_, v, int(np.asarray(recs[v]["path_nodes"])[16][1]), np.asarray(recs[v]["path_nodes"])[15][3] / 100.0) for s_, v in held[:14] if s_)))
    if os.environ.get("PROFILE_PASSES"):  # a round's passes taken apart, per round, for every vehicle of the step (us)
        for v in range(len(recs)):
            t = np.asarray(recs[v]["path_nodes"]); n = max(1.0, t[16][0])
            print("   veh %2d rounds %3d, per round: share decision %.2f check items %.2f sincos %.2f P1 barrier %.2f | boundary -> selection %.2f selection %.2f us" % (
                v, int(t[16][0]), *[t[13][i] / 100.0 / n for i in range(6)]))
    top = int(os.environ.get("PROFILE_TOP", "8"))
    org = min(np.asarray(r["path_nodes"])[15][7] for r in recs)
    for r in sorted(rows, reverse=True)[:top]:
        print("   [workgroup started %5.0f us into the kernel]" % ((np.asarray(recs[r[1]]["path_nodes"])[15][7] - org) / 100.0), end="")
        print("   total %.0f us veh %d level %d popped %d processed %d nodes %d rounds %d | work %.0f arrival %.0f select %.0f wait %.0f us | preds %d | prologue %.0f check %.0f verdict %.0f expand %.0f phaseB %.0f refill %.0f us" % r)
