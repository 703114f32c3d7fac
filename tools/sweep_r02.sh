#!/bin/bash
# board-size sweep: the r01 library (tools/bin/libtwixt_hip_base.so) against this build, B = 2^20
L=twixt_for_open_spiel_amd/libtwixt_hip.so
B=tools/bin/libtwixt_hip_base.so
for n in ${SWEEP_N:-5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24}; do
  echo "== N=$n"; AB_N=$n python tools/ab_bench.py $B $L 2>/dev/null | grep median
done
