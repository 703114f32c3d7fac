#!/bin/bash
# A/B: opponent's free-cell count taken only when a board is nearly full, against the build before
A=tools/bin/libtwixt_hip_slots.so
B=tools/bin/libtwixt_hip_cntopp.so
for n in ${AB_SIZES:-24 24 21 19 16 13 12 10 8 6}; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py $A $B 2>/dev/null | grep median; done
