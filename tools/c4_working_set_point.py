#!/usr/bin/env python3
"""One line of the C4 working-set sweep (tools/c4_working_set_sweep.sh): the bench line's figures beside the PMC passes' of the TIMED launch
(the last launch of er_stream_kernel<false, ...> in each pass: the warm-up launch runs on the other deal and counts work per tile)."""
import csv
import json
import os
import sys


def last_launch(path, needle="er_stream_kernel<false"):
    per = {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            d = per.setdefault(int(r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per[max(per)] if per else {}


def main():
    d, tris = sys.argv[1], int(sys.argv[2])
    b = json.loads(open(os.path.join(d, "bench.log")).read().strip().splitlines()[-1])
    r, a = b["roofline"], b["accel"]
    c = {p: last_launch(os.path.join(d, p, "run_counter_collection.csv")) for p in ("tcc", "fetch", "write", "tcp2", "tcp3")}
    steps = b["steps"]
    hit = c["tcc"]["TCC_HIT_sum"] / max(1.0, c["tcc"]["TCC_HIT_sum"] + c["tcc"]["TCC_MISS_sum"])
    fabric = (2.0 * c["fetch"]["FETCH_SIZE"] + c["write"]["WRITE_SIZE"]) * 1024.0
    fabric_gbps = fabric / (c["fetch"]["ns"] / 1e9) / 1e9
    lat = c["tcp3"]["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, c["tcp2"]["TCP_TCC_READ_REQ_sum"])
    tcp_lat = c["tcp3"]["TCP_TCP_LATENCY_sum"] / max(1.0, c["tcp3"]["TCP_TA_TCP_STATE_READ_sum"])
    l1_miss = c["tcp2"]["TCP_TCC_READ_REQ_sum"] / max(1.0, c["tcp2"]["TCP_TOTAL_CACHE_ACCESSES_sum"])
    ws_mb = (a["node_bytes"] * a["nodes"] + 48.0 * 10000 * tris) / 1e6
    alg = r["achieved"] * 1e9 * (b["ms_per_step"] * 1e-3)      # algorithmic traversal bytes per sample pass
    print(f"{tris:5d} triangles per blob: nodes + triangle records {ws_mb:7.1f} MB | {b['value']:7.1f} Msamples/s, frac {r['frac']:.3f}, "
          f"{r['node_visits_per_ray']:.2f} visits + {r['tri_tests_per_ray']:.2f} tests per ray | L2 hit {hit:.3f}, L1 miss share {l1_miss:.3f}, "
          f"{lat:.0f} cycles beyond the L1, {tcp_lat:.0f} in the L1 | fabric {fabric / steps / 1e9:.1f} GB per pass = {fabric / steps / max(alg, 1.0):.2f} x algorithmic, {fabric_gbps:.0f} GB/s")


if __name__ == "__main__":
    main()
