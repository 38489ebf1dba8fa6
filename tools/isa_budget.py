#!/usr/bin/env python3
"""Static instruction budget of er_stream_kernel from the device assembly.

Compiles elevenrender_amd/csrc/er_stream.hip with `hipcc -S --cuda-device-only -DER_ISA_MARKS` (the product flags otherwise):
every ER_MARK("name") of the source leaves an assembler comment in the instruction stream, and the instructions between two
consecutive marks are counted by class.  Regions are stretches of the TEXT of the kernel in layout order: a cold block the
compiler moved elsewhere is counted where it lies, so read the figures as the budget of the straight-line path plus whatever
cold code shares the stretch.  The unmarked build's totals are printed beside the marked build's so that the marks can be
seen not to change the code (volatile asm comments only pin the order of memory operations around them).

    python tools/isa_budget.py [--kernel 'ILb0ELb0E'] [--out profiles/r03_isa_budget.txt]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elevenrender_amd", "csrc")
FLAGS = ["-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith(("v_readlane", "v_writelane")):
        return "lane_spill_move"
    if op.startswith("v_readfirstlane"):
        return "valu"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_cvt_"):
        return "valu_cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_load_", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


COLS = ["valu", "valu_pk", "valu_cvt", "valu_trans", "salu", "smem", "lds", "vmem", "scratch", "lane_spill_move", "branch", "wait", "other"]


def kernel_text(asm, needle):
    lines = asm.splitlines()
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z16er_stream_kernel" + needle + r".*:", l):
            start = i
        elif start is not None and ".end_amdhsa_kernel" in l:
            end = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + needle)
    return lines[start:end]


def budget(lines):
    regions = collections.OrderedDict()
    cur = "(prologue)"
    regions[cur] = collections.Counter()
    for l in lines:
        s = l.strip()
        m = re.match(r"^; ER_MARK (\S+)", s)
        if m:
            cur = m.group(1)
            n = 2
            base = cur
            while cur in regions:          # a mark the compiler duplicated (unrolled / cloned block)
                cur = f"{base}#{n}"
                n += 1
            regions[cur] = collections.Counter()
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        regions[cur][classify(op)] += 1
    return regions


def meta(lines_all, needle):
    out = {}
    txt = "\n".join(lines_all)
    for key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        m = re.search(r"\.name:\s+_Z16er_stream_kernel" + needle + r".*?\." + key + r":\s+(\d+)", txt, re.S)
        # metadata order is alphabetical inside one kernel's map, so search the whole map instead
        out[key] = None
    # the metadata map of one kernel: from '.name: <kernel>' back to the previous '- .agpr_count' and forward to the next
    blocks = re.split(r"\n  - \.agpr_count", txt)
    for b in blocks:
        if re.search(r"\.name:\s+_Z16er_stream_kernel" + needle, b):
            for key in out:
                m = re.search(r"\." + key + r":\s+(\d+)", b)
                if m:
                    out[key] = int(m.group(1))
    return out


def compile_asm(marks):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "er_stream.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-DER_ISA_MARKS"] if marks else []) + ["er_stream.hip", "-o", out]
        subprocess.check_call(cmd, cwd=CSRC, stderr=subprocess.DEVNULL)
        return open(out).read()


def fmt_table(regions):
    hdr = f"{'region':28s}" + "".join(f"{c[:9]:>10s}" for c in COLS) + f"{'total':>8s}"
    rows = [hdr]
    tot = collections.Counter()
    for name, c in regions.items():
        rows.append(f"{name:28s}" + "".join(f"{c[k]:10d}" for k in COLS) + f"{sum(c.values()):8d}")
        tot.update(c)
    rows.append(f"{'(whole kernel)':28s}" + "".join(f"{tot[k]:10d}" for k in COLS) + f"{sum(tot.values()):8d}")
    return "\n".join(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ILb0ELb0E", help="mangled template arguments: ILb<COUNT>ELb<EXT>E")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    marked = compile_asm(True)
    plain = compile_asm(False)
    km = kernel_text(marked, a.kernel)
    kp = kernel_text(plain, a.kernel)
    rm = budget(km)
    rp = budget(kp)
    text = []
    text.append(f"er_stream_kernel<{a.kernel}>: static instruction counts by region (tools/isa_budget.py; hipcc -S, product flags)")
    text.append("")
    text.append(fmt_table(rm))
    text.append("")
    tp = collections.Counter()
    for c in rp.values():
        tp.update(c)
    text.append("unmarked (product) build, whole kernel: " + ", ".join(f"{k} {tp[k]}" for k in COLS if tp[k]) + f", total {sum(tp.values())}")
    text.append("metadata marked:   " + str(meta(marked.splitlines(), a.kernel)))
    text.append("metadata unmarked: " + str(meta(plain.splitlines(), a.kernel)))
    # tracer loop vs shader loop: spill traffic sites
    tr = collections.Counter()
    sh = collections.Counter()
    for name, c in rm.items():
        base = name.split("#")[0]
        if base in ("tracer_loop_end", "pool_loop_end", "shader_loop_end", "epilogue", "(prologue)"):
            continue
        if base.startswith(("tracer_", "pool_", "tri_block", "node_block", "apply_end")):
            tr.update(c)
        elif base.startswith("shader_"):
            sh.update(c)
    for label, c in (("tracer loop", tr), ("shader loop", sh)):
        text.append(f"{label}: scratch sites {c['scratch']}, v_readlane/v_writelane sites {c['lane_spill_move']}, "
                    f"VALU {c['valu'] + c['valu_pk'] + c['valu_cvt'] + c['valu_trans']} (packed {c['valu_pk']}, cvt {c['valu_cvt']}, transcendental {c['valu_trans']}), "
                    f"SALU {c['salu']}, LDS {c['lds']}, VMEM {c['vmem']}, branches {c['branch']}, waits {c['wait']}")
    s = "\n".join(text)
    print(s)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
