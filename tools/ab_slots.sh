#!/bin/bash
# A/B: per-slot journal line loads (and plain instead of non-temporal record stores) against the build before
A=tools/bin/libtwixt_hip_or2.so
B=tools/bin/libtwixt_hip_slots.so
C=tools/bin/libtwixt_hip_slots_recplain.so
for n in 24 24 21 19 16 13 12 10; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py $A $B $C 2>/dev/null | grep median; done
