#!/usr/bin/env python3
"""One table per rocprofv3 --pmc pass directory set (tools/pmc_passes.sh): the counters of one kernel summed
over its launches, with the ratios DESIGN.md quotes.  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles
(MI355X_MICROARCH.md); a wave is, at any time, issuing (ACTIVE_INST_ANY), stalled at issue (WAIT_INST_ANY) or parked in a wait
(WAIT_ANY).

    python tools/pmc_table.py gpurun_out/pmc_stream_r03base [kernel-substring] > profiles/r03_pmc_stream_kernel.txt
"""
import collections
import csv
import glob
import os
import sys


def collect(base, needle):
    acc = collections.OrderedDict()
    meta = {}
    for d in sorted(glob.glob(os.path.join(base, "*", "run_counter_collection.csv"))):
        name = d.split(os.sep)[-2]
        seen = set()
        for r in csv.DictReader(open(d)):
            if needle not in r["Kernel_Name"]:
                continue
            acc[(name, r["Counter_Name"])] = acc.get((name, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                m = meta.setdefault(name, {"launches": 0, "ns": 0, "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"]})
                m["launches"] += 1
                m["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc, meta


def main():
    base = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else "er_stream_kernel"
    acc, meta = collect(base, needle)
    if not acc:
        raise SystemExit("no rows for " + needle)
    print(f"# {needle}: rocprofv3 --pmc passes under {base} (sums over the launches of each pass)")
    for name, m in meta.items():
        print(f"# pass {name}: {m['launches']} launches, {m['ns'] / 1e6:.2f} ms, VGPR {m['vgpr']}, LDS {m['lds']} B, scratch {m['scratch']} B")
    print(f"{'pass':10s} {'counter':40s} {'sum':>16s}")
    for (name, ctr), v in acc.items():
        print(f"{name:10s} {ctr:40s} {v:16.6g}")
    g = lambda n, c: acc.get((n, c))
    print("\n# derived")

    def ratio(label, a, b, scale=1.0, fmt="{:.4f}"):
        if a is not None and b:
            print(f"{label:70s} " + fmt.format(a / b * scale))

    for p in ("sq", "sq3"):
        wc = g(p, "SQ_WAVE_CYCLES")
        if wc:
            ratio(f"[{p}] wave time issuing an instruction   (ACTIVE_INST_ANY / WAVE_CYCLES)", g(p, "SQ_ACTIVE_INST_ANY"), wc)
            ratio(f"[{p}] wave time stalled at issue         (WAIT_INST_ANY / WAVE_CYCLES)", g(p, "SQ_WAIT_INST_ANY"), wc)
            ratio(f"[{p}] wave time parked in a wait         (WAIT_ANY / WAVE_CYCLES)", g(p, "SQ_WAIT_ANY") or g("sq2", "SQ_WAIT_ANY"), wc)
            ratio(f"[{p}] wave time issuing VALU             (ACTIVE_INST_VALU / WAVE_CYCLES)", g(p, "SQ_ACTIVE_INST_VALU"), wc)
            ratio(f"[{p}] wave time issuing scalar           (ACTIVE_INST_SCA / WAVE_CYCLES)", g(p, "SQ_ACTIVE_INST_SCA"), wc)
            ratio(f"[{p}] wave time issuing LDS              (ACTIVE_INST_LDS / WAVE_CYCLES)", g(p, "SQ_ACTIVE_INST_LDS"), wc)
    ratio("VALU lane utilisation   (THREAD_CYCLES_VALU / (64 * INSTS_VALU))", g("sq2", "SQ_THREAD_CYCLES_VALU"), (g("sq2", "SQ_INSTS_VALU") or 0) * 64)
    ratio("VALU instructions per vector-memory read instruction", g("sq2", "SQ_INSTS_VALU"), g("sq2", "SQ_INSTS_VMEM_RD"), fmt="{:.1f}")
    ratio("scalar instructions per VALU instruction", g("sq3", "SQ_INSTS_SALU") or g("sq2", "SQ_INSTS_SALU"), g("sq2", "SQ_INSTS_VALU"))
    ratio("instruction-cache miss rate   (SQC_ICACHE_MISSES / SQC_ICACHE_REQ)", g("icache", "SQC_ICACHE_MISSES"), g("icache", "SQC_ICACHE_REQ"), fmt="{:.6f}")
    ratio("instruction fetches in flight per wave   (SQ_IFETCH_LEVEL / WAVE_CYCLES)", g("icache", "SQ_IFETCH_LEVEL"), g("icache", "SQ_WAVE_CYCLES"))
    hit, miss = g("tcc", "TCC_HIT_sum"), g("tcc", "TCC_MISS_sum")
    if hit is not None and miss is not None:
        ratio("L2 hit rate   (TCC_HIT / (TCC_HIT + TCC_MISS))", hit, hit + miss)
    for c in ("TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"):
        ga = g("tcp_stall", "GRBM_GUI_ACTIVE")
        if g("tcp_stall", c) is not None and ga:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; the TCP counters sum the 256 CUs
            ratio(f"{c} per CU / kernel cycles", g("tcp_stall", c) / 256.0, ga / 8.0)
    # vector-memory front end (the ta*/tcp*/td* passes of tools/pmc_passes.sh): *_sum counters add the 256 CUs, GRBM_GUI_ACTIVE adds the 8 XCDs
    def per_cu_cycle(p, c):
        return (g(p, c) / 256.0, (g(p, "GRBM_GUI_ACTIVE") or 0) / 8.0) if g(p, c) is not None else (None, None)
    for label, p, c in (("TA busy   (TA_TA_BUSY per CU / kernel cycles)", "ta1", "TA_TA_BUSY_sum"),
                        ("TA address stalled by the TCP / kernel cycles", "ta2", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"),
                        ("TA data stalled by the TCP / kernel cycles", "ta2", "TA_DATA_STALLED_BY_TC_CYCLES_sum"),
                        ("vector-memory wave-instructions per CU per cycle", "ta1", "TA_TOTAL_WAVEFRONTS_sum"),
                        ("TCP 64-byte cache accesses per CU per cycle", "tcp2", "TCP_TOTAL_CACHE_ACCESSES_sum"),
                        ("TCP lane accesses per CU per cycle", "tcp4", "TCP_TOTAL_ACCESSES_sum"),
                        ("TCP -> L2 read requests per CU per cycle", "tcp2", "TCP_TCC_READ_REQ_sum")):
        a, b = per_cu_cycle(p, c)
        ratio(label, a, b)
    ratio("TA busy cycles per vector-memory wave-instruction", g("ta1", "TA_TA_BUSY_sum"), g("ta1", "TA_TOTAL_WAVEFRONTS_sum"), fmt="{:.1f}")
    ratio("L1 miss share   (TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)", g("tcp2", "TCP_TCC_READ_REQ_sum"), g("tcp2", "TCP_TOTAL_CACHE_ACCESSES_sum"))
    ratio("L2 read latency seen by the TCP, cycles   (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)", g("tcp3", "TCP_TCC_READ_REQ_LATENCY_sum"), g("tcp2", "TCP_TCC_READ_REQ_sum"), fmt="{:.0f}")
    ratio("time of a wave-instruction in the TCP, cycles   (TCP_TCP_LATENCY / TCP_TA_TCP_STATE_READ)", g("tcp3", "TCP_TCP_LATENCY_sum"), g("tcp3", "TCP_TA_TCP_STATE_READ_sum"), fmt="{:.0f}")
    for label, p, c in (("TD busy   (TD_TD_BUSY per CU / kernel cycles)", "td1", "TD_TD_BUSY_sum"),
                        ("TD stalled by the TCP / kernel cycles", "td1", "TD_TC_STALL_sum")):
        a, b = per_cu_cycle(p, c)
        ratio(label, a, b)
    ratio("TD busy cycles per load wave-instruction", g("td2", "TD_TD_BUSY_sum"), g("td2", "TD_LOAD_WAVEFRONT_sum"), fmt="{:.1f}")
    ratio("LDS bank-conflict cycles / LDS active cycles", g("lds", "SQ_LDS_BANK_CONFLICT"), g("lds", "SQ_LDS_IDX_ACTIVE"))
    f, w = g("fetch", "FETCH_SIZE"), g("write", "WRITE_SIZE")
    if f is not None and w is not None:
        ns = meta["fetch"]["ns"]
        # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B on this ROCm (guide: FETCH_SIZE x 2 for 128-byte requests on gfx950)
        print(f"{'fabric bytes: 2 x FETCH_SIZE + WRITE_SIZE (KB), per ms of kernel':70s} {(2 * f + w) / (ns / 1e6):.1f} KB/ms = {(2 * f + w) * 1024 / (ns / 1e9) / 1e9:.1f} GB/s")


if __name__ == "__main__":
    main()
