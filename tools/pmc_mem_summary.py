#!/usr/bin/env python3
"""Sum the counters of tools/pmc_mem.sh per kernel: python tools/pmc_mem_summary.py gpurun_out/pmc_mem_r02 [out.csv]"""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    m = re.match(r"(?:void )?(?:erd::)?(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]

def main():
    root = sys.argv[1]
    rows = []
    for d in sorted(glob.glob(os.path.join(root, "*/"))):
        pas = os.path.basename(d.rstrip("/"))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: [0.0, 0])
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = (short(r["Kernel_Name"]), r["Counter_Name"])
                    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
            for (kern, ctr), (s, n) in sorted(acc.items()):
                rows.append((pas, kern, ctr, n, s, s / n))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["pass", "kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
    for r in rows:
        if r[1].startswith("er_wf_trace") or r[1].startswith("er_wf_shade") or r[1].startswith("er_stream") or r[1].startswith("er_fused"):
            w.writerow([r[0], r[1], r[2], r[3], f"{r[4]:.6g}", f"{r[5]:.6g}"])

main()
