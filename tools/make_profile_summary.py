"""Turn a tools/record_run.sh output directory into the small tracked files under profiles/.
Usage: python tools/make_profile_summary.py gpurun_out/record_r01 r01"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "bench.log"), os.path.join(dst, f"{tag}_bench.log"))
shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


rows = []
traffic = {}
fetch = per_kernel(os.path.join(src, "pmc_fetch", "run_counter_collection.csv"))
write = per_kernel(os.path.join(src, "pmc_write", "run_counter_collection.csv"))
for k in sorted(set(fetch) | set(write)):
    fc, fs = fetch.get(k, {}).get("FETCH_SIZE", [0, 0.0])
    wc, ws = write.get(k, {}).get("WRITE_SIZE", [0, 0.0])
    # MI355X_MICROARCH.md, HBM: both counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> x2
    fetch_b = 2.0 * fs * 1024.0 / max(fc, 1)
    write_b = ws * 1024.0 / max(wc, 1)
    rows.append([k, fc, round(fs / max(fc, 1), 2), wc, round(ws / max(wc, 1), 2), round(fetch_b + write_b)])
    traffic[k] = round(fetch_b + write_b)
with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_fetch_pass", "FETCH_SIZE_KB_per_launch_raw", "launches_write_pass", "WRITE_SIZE_KB_per_launch",
                "hbm_bytes_per_launch (2*FETCH + WRITE)"])
    w.writerows(rows)

valu = per_kernel(os.path.join(src, "pmc_valu", "run_counter_collection.csv"))
with open(os.path.join(dst, f"{tag}_pmc_valu.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE(sum of 8 XCDs)",
                "valu_busy = 4*ACTIVE_INST_VALU/1024 SIMDs/(GUI_ACTIVE/8)", "lane_utilisation = THREAD_CYCLES/(64*ACTIVE_INST_VALU)"])
    for k, d in sorted(valu.items()):
        n = d["SQ_INSTS_VALU"][0]
        iv, av, tc, ga = (d[c][1] for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"))
        busy = 4.0 * av / 1024.0 / (ga / 8.0) if ga else 0.0
        util = tc / (64.0 * av) if av else 0.0
        w.writerow([k, n, int(iv), int(av), int(tc), int(ga), round(busy, 3), round(util, 3)])

l2_path = os.path.join(src, "pmc_l2", "run_counter_collection.csv")
if os.path.exists(l2_path):
    l2 = per_kernel(l2_path)
    with open(os.path.join(dst, f"{tag}_pmc_l2.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "TCC_HIT_sum", "TCC_MISS_sum", "l2_hit_rate = HIT / (HIT + MISS)"])
        for k, d in sorted(l2.items()):
            h, m = d["TCC_HIT_sum"][1], d["TCC_MISS_sum"][1]
            w.writerow([k, d["TCC_HIT_sum"][0], int(h), int(m), round(h / (h + m), 4) if h + m else 0.0])

key = [k for k in traffic if k.startswith("er_wf_trace<false>")]
# the streaming schedule is one kernel per call: tools/record_run.sh's PMC passes make one call of 1 step (warm-up) and one of 4
# steps with the uninstrumented variant, so bytes per step = the variant's total / 5
PMC_STEPS = 5
skey = [k for k in traffic if k.startswith("er_stream_kernel<false")]
stream_per_step = None
if skey:
    fs = fetch.get(skey[0], {}).get("FETCH_SIZE", [0, 0.0])[1]
    ws = write.get(skey[0], {}).get("WRITE_SIZE", [0, 0.0])[1]
    stream_per_step = round((2.0 * fs + ws) * 1024.0 / PMC_STEPS)
out = {"er_wf_trace_hbm_bytes_per_launch": traffic[key[0]] if key else None,
       "er_stream_kernel_hbm_bytes_per_step": stream_per_step,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = 2*FETCH_SIZE(KB)*1024 + WRITE_SIZE(KB)*1024 "
               "(MI355X_MICROARCH.md HBM section); per launch of one slot pool (a third of the frame's rays)",
       "all_kernels": traffic}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
