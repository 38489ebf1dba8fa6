#!/usr/bin/env python3
"""Per-launch durations of one kernel from rocprofv3's kernel trace (`--kernel-trace --stats --output-format csv`).
The stats file averages every launch of a name -- bench.py's warm-up call together with its timed repeats -- so the figure that must
agree with the bench line's `roofline.avg_launch_ms` / `repeats.region_ms` is read from the trace itself.

    python tools/kernel_launches.py gpurun_out/record_r05/stats/run_kernel_trace.csv er_stream_kernel"""
import csv
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_name = {}
    for r in rows:
        by_name.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for name, ms in by_name.items():
        print(f"{name}: {len(ms)} launches, ms each: {[round(m, 3) for m in ms]}")
        if len(ms) > 1:
            rest = ms[1:]      # (the first launch of bench.py is its warm-up call, of --warmup steps)
            print(f"  after the first: mean {sum(rest) / len(rest):.3f} ms, median {sorted(rest)[(len(rest) - 1) // 2]:.3f} ms, min {min(rest):.3f}, max {max(rest):.3f}")


if __name__ == "__main__":
    main()
