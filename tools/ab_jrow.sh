#!/bin/bash
# A/B of the [slot][board] journal layout against the tagged round-2 stable library, per board size
S=tools/bin/libtwixt_hip_stable.so
L=tools/bin/libtwixt_hip_jrow.so
for n in 24 24 19 16 12 10 8 6; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py $S $L 2>/dev/null | grep median; done
