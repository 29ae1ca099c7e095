"""Extracts the lanelet geometry and topology of the reference's CPM-lab map into a small data fixture.

Input  (this container only): /root/reference/scenarios/road_network/lanelets/offline_road_data/LabMapCommonRoad.xml
Output (committed): p-dmpc_amd/pdmpc/data/labmap.npz
    bounds    float64 [n_lanelets, 2 (left/right), P_max, 2 (x/y)]  NaN padded
    n_points  int32   [n_lanelets]
    pred/succ int32   [n_lanelets, 4]  lanelet ids (1-based), 0 padded
    adj       int32   [n_lanelets, 2 (left/right), 2 (id, same_direction)]
The file holds map DATA (point coordinates and references between lanelets), no reference source code.
Run:  python tests/golden/make_labmap_fixture.py
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

SRC = "/root/reference/scenarios/road_network/lanelets/offline_road_data/LabMapCommonRoad.xml"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DST = os.path.join(ROOT, "p-dmpc_amd", "pdmpc", "data", "labmap.npz")


def main():
    root = ET.parse(SRC).getroot()
    lanelets = root.findall("lanelet")
    ids = [int(l.get("id")) for l in lanelets]
    assert ids == list(range(1, len(ids) + 1)), "lanelet ids are expected to be 1..n in order"
    n = len(ids)
    pts = []
    for l in lanelets:
        lb = [(float(p.find("x").text), float(p.find("y").text)) for p in l.find("leftBound").findall("point")]
        rb = [(float(p.find("x").text), float(p.find("y").text)) for p in l.find("rightBound").findall("point")]
        assert len(lb) == len(rb)
        pts.append((lb, rb))
    pmax = max(len(lb) for lb, _ in pts)
    bounds = np.full((n, 2, pmax, 2), np.nan)
    n_points = np.zeros(n, dtype=np.int32)
    pred = np.zeros((n, 4), dtype=np.int32)
    succ = np.zeros((n, 4), dtype=np.int32)
    adj = np.zeros((n, 2, 2), dtype=np.int32)
    for i, l in enumerate(lanelets):
        lb, rb = pts[i]
        n_points[i] = len(lb)
        bounds[i, 0, : len(lb)] = lb
        bounds[i, 1, : len(rb)] = rb
        for q, e in enumerate(l.findall("predecessor")):
            pred[i, q] = int(e.get("ref"))
        for q, e in enumerate(l.findall("successor")):
            succ[i, q] = int(e.get("ref"))
        for side, tag in enumerate(("adjacentLeft", "adjacentRight")):
            e = l.find(tag)
            if e is not None:
                adj[i, side] = (int(e.get("ref")), 1 if e.get("drivingDir") == "same" else 0)
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    np.savez_compressed(DST, bounds=bounds, n_points=n_points, pred=pred, succ=succ, adj=adj)
    print("wrote", DST, os.path.getsize(DST), "bytes;", n, "lanelets, up to", pmax, "points per bound")


if __name__ == "__main__":
    sys.exit(main())
