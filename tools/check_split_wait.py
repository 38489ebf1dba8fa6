#!/usr/bin/env python3
"""Checks the device assembly of er_stream.hip built with -DER_STREAM_SPLIT_WAIT=1: between the statement that issues the
loads of a traversal step ("; ER_SPLIT issue tri=T node=N") and the statements that wait for them ("wait_tri": the T triangle pieces,
"wait_node": the N node pieces) no instruction may name a destination register of a load that has not been waited for -- the compiler does not
know those loads are in flight (cdna_hip_programming.md 5.7 item 1), so a copy, spill or reuse there would be silent corruption.

    python tools/check_split_wait.py file.s        exit code 0 = every occurrence is clean (and there is at least one)
The Makefile runs it on every build of er_stream.o and deletes the object when it fails.
"""
import re
import sys


def regs_of(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def named(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    n_sites, bad = 0, 0
    i = 0
    while i < len(lines):
        if "ER_SPLIT issue" not in lines[i]:
            i += 1
            continue
        n_sites += 1
        j = i + 1
        loads = []
        while "global_load_dwordx4" in lines[j]:
            loads.append(regs_of(lines[j].split()[1].rstrip(",")))
            j += 1
        # the marker says how many loads of each kind the statement holds ("tri=6 node=4"; eleven loads, six first, without it)
        m = re.search(r"tri=(\d+) node=(\d+)", lines[i])
        n_tri, n_node = (int(m.group(1)), int(m.group(2))) if m else (6, 5)
        assert len(loads) == n_tri + n_node, (i, len(loads), n_tri, n_node)
        tri, node = set().union(*loads[:n_tri]), set().union(*loads[n_tri:])
        pending = tri | node
        k = j
        state = "tri"
        while True:
            l = lines[k].strip()
            if "ER_SPLIT wait_tri" in l:
                pending = set(node)
                state = "node"
            elif "ER_SPLIT wait_node" in l:
                break
            elif l and not l.startswith((";", ".", "s_waitcnt")) and not l.endswith(":"):
                hit = named(l) & pending
                if hit:
                    bad += 1
                    print(f"line {k + 1}: `{l}` names v{sorted(hit)} while its load is in flight (waiting for {state})")
            k += 1
            if k - j > 3000:
                raise SystemExit(f"site at line {i + 1}: no wait_node within 3000 lines")
        i = k
    print(f"{n_sites} split-wait sites, {bad} offending instructions")
    return 1 if bad or not n_sites else 0


if __name__ == "__main__":
    sys.exit(main())
