#!/usr/bin/env python3
"""Static budget of the streaming kernel's SHADING STEP by source function.

Compiles elevenrender_amd/csrc/er_stream.hip with the product flags + `-g -S -DER_ISA_MARKS`, takes the instructions of
er_stream_kernel<false, false, 1024> between the marks `shader_step` and `shader_pixel_ring` (the shading step proper: slot record,
resolve, er_bounce.inc, stores) and charges each instruction to the source function its `.loc` line lies in (inlined code keeps the
line of the function it was written in).  Per function: instructions by class and an estimate of the vector-pipe time they take,
priced with the measured table of tools/microbench/valu_cost_bench.hip (profiles/r04_microbench_valu_cost.log, 4 waves per SIMD, cycles
at the nominal clock: v_add/mul/sub/fmac/and/mov 2.5, v_fma_f32 3.0, other single-rate-looking VOP2/VOP3 and every compare, convert,
select, min/max 4.3, packed f32 and f64 arithmetic 4.5, transcendental 8.4).  STATIC: a function inlined at three places counts three
times, cold paths (the exact re-trace, the two-candidate resolve) count in full.

    python tools/shader_function_budget.py [--out profiles/r04_shader_function_budget.txt]
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
FLAGS = ["-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-g", "-DER_ISA_MARKS"]
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_mov_b64", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_accvgpr")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def price(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return 2.5
    if op.startswith(TRANS):
        return 8.4
    if op.startswith("v_pk_") or op.endswith("_f64") or "_f64_" in op:
        return 4.5
    if op.startswith("v_fma_f32"):
        return 3.0
    if op.startswith(FAST):
        return 2.5
    return 4.3


def function_ranges(path):
    """[(first line, last line, name)] of the function definitions of a source file (brace counting from the definition line)."""
    src = open(path).read().split("\n")
    out = []
    head = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:ERD|ER_HD|ER_RING_FN|ERM|__device__|__host__|static|inline|__global__)[\w\s:<>\*&,]*?\b([A-Za-z_]\w*)\s*\([^;]*$")
    i = 0
    while i < len(src):
        m = head.match(src[i])
        if m and not src[i].strip().startswith(("//", "#", "return")):
            name = m.group(1)
            depth, j, seen = 0, i, False
            while j < len(src):
                depth += src[j].count("{") - src[j].count("}")
                seen = seen or "{" in src[j]
                if seen and depth <= 0:
                    break
                if not seen and src[j].rstrip().endswith(";"):
                    break
                j += 1
            if seen:
                out.append((i + 1, j + 1, name))
                i = j
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "er_stream.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, "er_stream.hip"), "-o", asm], cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z16er_stream_kernelILb0ELb0ELj1024"))
    a = next(i for i, l in enumerate(lines[start:], start) if "ER_MARK shader_step" in l)
    b = next(i for i, l in enumerate(lines[start:], start) if "ER_MARK shader_pixel_ring" in l)
    ranges = {}
    for f in set(files.values()):
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            ranges[f] = function_ranges(p)
    # the .loc in force at the start of the region
    cur = ("?", 0)
    for l in lines[start:a]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
    acc = collections.defaultdict(lambda: collections.Counter())
    for l in lines[a:b]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = t[0]
        f, ln = cur
        name = None
        for lo, hi, n in ranges.get(f, []):
            if lo <= ln <= hi:
                name = n
        if f == "er_bounce.inc":
            name = "er_bounce.inc (the bounce's own statements)"
        key = f"{f}:{name}" if name else f"{f} (kernel body)"
        c = acc[key]
        if op.startswith(("v_readlane", "v_writelane")):
            c["lane"] += 1
            c["cyc"] += price(op)
        elif op.startswith("v_"):
            c["valu"] += 1
            c["cyc"] += price(op)
            if op.startswith(TRANS): c["trans"] += 1
            if op.endswith("_f64") or "_f64_" in op: c["f64"] += 1
            if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")): c["div"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    tot = collections.Counter()
    for c in acc.values():
        tot.update(c)
    rows = sorted(acc.items(), key=lambda kv: -kv[1]["cyc"])
    out = [f"# static budget of the shading step of er_stream_kernel<false, false, 1024> by source function ({b - a} assembly lines between the marks shader_step and shader_pixel_ring)",
           "# cyc = vector-pipe cycles of the function's instructions at the prices of profiles/r04_microbench_valu_cost.log; div = v_div_scale / fmas / fixup (3 per IEEE divide)",
           f"{'function':62s} {'cyc':>8s} {'share':>6s} {'VALU':>6s} {'f64':>5s} {'trans':>5s} {'div':>5s} {'SALU':>6s} {'nop':>5s} {'VMEM':>5s} {'scratch':>7s} {'LDS':>4s} {'lane':>5s}"]
    for k, c in rows:
        out.append(f"{k[:62]:62s} {c['cyc']:8.0f} {c['cyc'] / max(tot['cyc'], 1):6.3f} {c['valu']:6d} {c['f64']:5d} {c['trans']:5d} {c['div']:5d} {c['salu']:6d} {c['nop']:5d} {c['vmem']:5d} {c['scratch']:7d} {c['lds']:4d} {c['lane']:5d}")
    out.append(f"{'total':62s} {tot['cyc']:8.0f} {1.0:6.3f} {tot['valu']:6d} {tot['f64']:5d} {tot['trans']:5d} {tot['div']:5d} {tot['salu']:6d} {tot['nop']:5d} {tot['vmem']:5d} {tot['scratch']:7d} {tot['lds']:4d} {tot['lane']:5d}")
    text = "\n".join(out)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
