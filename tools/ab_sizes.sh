#!/bin/bash
# A/B (or A/B/C) of step-kernel libraries per board size, one process per size with interleaved rounds
# (tools/ab_bench.py), B = 2^20.  usage: AB_SIZES="24 19 12" tools/ab_sizes.sh libA.so[:KNOB=V] libB.so [libC.so]
# (round 2 used it for: journal rows vs chunk journal, ds_or_b64 pairing, per-slot journal lines, lazy
# opponent count -- profiles/r02_*_ab.txt; the libraries compared were builds of the commits named there)
for n in ${AB_SIZES:-24 24 21 19 16 13 12 10 8 6}; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py "$@" 2>/dev/null | grep median; done
