#!/bin/bash
# resident waves per CU of the one-lane kernel, per board size (TWX_WAVES_PER_CU; the default is what the
# occupancy API reports): picks preferred_waves_per_cu() in twixt_engine.hip
L=twixt_for_open_spiel_amd/libtwixt_hip.so
for n in ${SWEEP_N:-5 6 7 8 9 10 11 12 13 14 15 16 17 18}; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py $L $L:TWX_WAVES_PER_CU=4 $L:TWX_WAVES_PER_CU=6 $L:TWX_WAVES_PER_CU=8 $L:TWX_WAVES_PER_CU=10 $L:TWX_WAVES_PER_CU=12 $L:TWX_WAVES_PER_CU=16 2>/dev/null | grep median | sed 's/.*libtwixt_hip.so//'; done
