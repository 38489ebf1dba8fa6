"""Sum rocprofv3 --pmc counters per kernel from a results database (rocprofv3 ... -d DIR -o NAME writes
DIR/NAME_results.db).  Usage: python tools/pmc_summary.py DB [kernel-substring]"""
import sqlite3
import sys


def summarize(path, needle=""):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pick = lambda key: [t for t in tabs if key in t][0]
    pmc, info, kd, ks = pick("pmc_event"), pick("info_pmc"), pick("kernel_dispatch"), pick("info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct k.id) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {kd} k on e.event_id=k.event_id join {ks} s on k.kernel_id=s.id group by 1,2")
    out = {}
    for name, ctr, val, n in db.execute(q):
        if needle in name:
            out.setdefault(name, {})[ctr] = (val, n)
    return out


if __name__ == "__main__":
    res = summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for k, d in res.items():
        for c, (v, n) in sorted(d.items()):
            print(f"{k[:48]:48s} {c:28s} {v:14.5g}  launches={n}")
