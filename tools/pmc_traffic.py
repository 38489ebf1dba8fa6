#!/usr/bin/env python3
"""Fabric bytes per sample pass of the streaming kernel from the `fetch` and `write` passes of tools/pmc_passes.sh, merged into
profiles/<tag>_pmc_traffic.json under the config's name (bench.py reads `roofline.traffic` from there and says "replayed").

    python tools/pmc_traffic.py gpurun_out/pmc_r05_c4 r05 C4 [steps of the timed call = 6]

bytes = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024 (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes; 99.8 % of these kernels' read requests are 128-byte ones, profiles/r02_pmc_request_sizes.csv).
A pass of tools/pmc_passes.sh runs `--warmup 2 --steps 6 --repeats 1`: two launches of the uninstrumented kernel, of which the second --
the timed call, 6 sample passes -- is counted (the instrumented replay is another template instance)."""
import csv
import json
import os
import sys


def total(path, counter, needle):
    """Counter sum, launch count and duration of the LAST launch of the kernel in the pass: bench.py's timed call.  (Its first launch is
    the warm-up call, which since round 5 runs on the default deal of tiles while the kernel counts work per tile; the timed call runs on
    the deal the library then chose -- the steady state of a render.)"""
    per = {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
            d = per.setdefault(int(r["Dispatch_Id"]), [0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
            d[0] += float(r["Counter_Value"])
    if not per:
        return 0.0, 0, 0
    last = per[max(per)]
    return last[0], len(per), last[1]


def main():
    src, tag, config = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    needle = "er_stream_kernel<false"
    f, nf, ns_f = total(os.path.join(src, "fetch", "run_counter_collection.csv"), "FETCH_SIZE", needle)
    w, nw, ns_w = total(os.path.join(src, "write", "run_counter_collection.csv"), "WRITE_SIZE", needle)
    if not nf or not nw:
        raise SystemExit("no er_stream_kernel<false, ...> rows in the fetch / write passes")
    per_step = (2.0 * f + w) * 1024.0 / steps
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_pmc_traffic.json")
    doc = json.load(open(dst)) if os.path.exists(dst) else {"note": __doc__.split("\n\n")[2]}
    doc[config] = {"er_stream_kernel_hbm_bytes_per_step": round(per_step), "read_bytes_per_step": round(2.0 * f * 1024.0 / steps),
                   "write_bytes_per_step": round(w * 1024.0 / steps), "steps_per_pass": steps, "launches": nf,
                   "kernel_ms_fetch_pass": round(ns_f / 1e6, 3), "kernel_ms_write_pass": round(ns_w / 1e6, 3),
                   "GBps_over_the_fetch_pass": round(per_step * steps / (ns_f / 1e9) / 1e9, 1), "source": src}
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps(doc[config], indent=1))


if __name__ == "__main__":
    main()
