#!/bin/bash
# A/B of the ds_or_b64 pairing: the row-journal library against this build, same journal period
S=tools/bin/libtwixt_hip_jrow.so
L=twixt_for_open_spiel_amd/libtwixt_hip.so
for np in 24:8 24:8 21:8 19:8 18:8 16:4 14:4 12:2 10:2 8:1 6:1; do n=${np%%:*}; p=${np##*:}; echo "== N=$n period $p"; AB_N=$n python tools/ab_bench.py $S:TWX_JOURNAL_PERIOD=$p $L:TWX_JOURNAL_PERIOD=$p 2>/dev/null | grep median; done
