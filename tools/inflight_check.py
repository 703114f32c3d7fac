#!/usr/bin/env python3
"""Diagnostic: do compiler-generated instructions touch registers that an untracked (inline-asm) load has in flight?

The step kernels issue their prefetches as inline-asm global loads and wait for them explicitly; the compiler
does not know the registers are in flight and is free to copy or spill them right after the asm statement
(it did: a scratch spill of a register pair an iteration before its data arrived).  This script scans the ISA
of one kernel linearly: registers written by an asm load (between ;;#ASMSTART / ;;#ASMEND) are in flight until
the next `s_waitcnt vmcnt(n)` retires them (all but the youngest n loads); any instruction OUTSIDE asm blocks that names one of them in between is reported.
Linear scan = no control flow: a report is a place to look at, not a proof either way.

usage: inflight_check.py file.s kernel_symbol_substring
"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") or (sym in l and re.match(r"^_Z\S+:", l)))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
regs_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(s):
    out = set()
    for m in regs_re.finditer(s):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


inflight, in_asm, hits = {}, False, 0
for i in range(start, end):
    l = text[i].split(";")[0].strip() if not text[i].strip().startswith(";;#") else text[i].strip()
    if l.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if l.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
    if m:
        # one in-order counter: all but the youngest n operations are done.  Stores are not modelled, so
        # "the youngest n asm loads stay in flight" over-states what is in flight (conservative).
        n = int(m.group(1))
        keep = sorted(set(inflight.values()))[-n:] if n else []
        inflight = {r: at for r, at in inflight.items() if at in keep}
        continue
    if in_asm:
        if l.startswith("global_load") or (l.startswith("global_atomic") and " sc0" in l):  # (returning atomic)
            dst = l.split()[1].rstrip(",")
            for r in regs(dst):
                inflight[r] = i + 1
        continue
    touched = regs(l) & set(inflight)
    if touched:
        hits += 1
        if hits <= 25:
            print("line %d: %s   <- in flight since line %d: v%s" % (i + 1 - start, l, inflight[min(touched)] - start, sorted(touched)))
print("%s: %d suspicious instruction(s)" % (sym, hits))
