"""tools/isa_phase_counts.py -- static instruction counts per phase of a kernel, from an ISA listing built
with -DTWX_MARK (the STAMP(i) boundaries become "; MARK Pi" comments; twixt_device_common.h).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -DTWX_MARK -S \\
        --cuda-device-only -o /tmp/engine_mark.s twixt_for_open_spiel_amd/csrc/twixt_engine.hip
  python tools/isa_phase_counts.py /tmp/engine_mark.s [kernel-regex = step_window1_kernelILi24E] [phase-to-dump]

Static counts: rare paths (auto-reset, swaps, on-the-spot floods) weigh as much as the main line.  The column
"moves" counts v_mov / v_readlane / v_writelane / v_accvgpr (register traffic, mostly SGPR spills and notes)."""
import re
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    kre = sys.argv[2] if len(sys.argv) > 2 else "step_window1_kernelILi24E"
    dump = sys.argv[3] if len(sys.argv) > 3 else None
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + kre + r"\S*:", l)][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    cur, cnt, order, ops = "entry", {}, [], Counter()
    for l in lines[start:end]:
        m = re.search(r"; MARK P(\d+)", l)
        if m:
            cur = "after P" + m.group(1)
            continue
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if dump is not None and cur == "after P" + dump:
            print(t)
        op = t.split()[0]
        kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        d = cnt.setdefault(cur, Counter())
        d[kind] += 1
        if op.startswith(("v_mov_b32", "v_readlane", "v_writelane", "v_accvgpr", "v_readfirstlane")):
            d["moves"] += 1
        ops[op] += 1
        if cur not in order:
            order.append(cur)
    if dump is not None:
        return
    tot = Counter()
    print("%-10s %6s %6s %6s %6s %6s" % ("phase", "valu", "moves", "salu", "lds", "vmem"))
    for c in order:
        d = cnt[c]
        tot.update(d)
        print("%-10s %6d %6d %6d %6d %6d" % (c, d["valu"], d["moves"], d["salu"], d["lds"], d["vmem"]))
    print("%-10s %6d %6d %6d %6d %6d" % ("total", tot["valu"], tot["moves"], tot["salu"], tot["lds"], tot["vmem"]))
    print("most frequent:", ", ".join("%s %d" % kv for kv in ops.most_common(16)))


if __name__ == "__main__":
    main()
