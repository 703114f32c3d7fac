#!/bin/bash
# journal period on the small and middle boards (step_kernel64_small): picks default_period()
L=twixt_for_open_spiel_amd/libtwixt_hip.so
for n in 5 6 7 8 9 10 11 12 13 14 15 16 17 18; do echo "== N=$n"; AB_N=$n python tools/ab_bench.py $L:TWX_JOURNAL_PERIOD=1 $L:TWX_JOURNAL_PERIOD=2 $L:TWX_JOURNAL_PERIOD=4 $L:TWX_JOURNAL_PERIOD=8 2>/dev/null | grep median | sed 's/.*\(lib[a-z_]*.so:[A-Z_=0-9]*\)/\1/'; done
