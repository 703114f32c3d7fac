#!/bin/bash
# one-off diagnostic batch (round 2): per-launch times and ablations of both step kernels
L=twixt_for_open_spiel_amd/libtwixt_hip.so
{
TWX_PAIR=1 python tools/launch_times.py $L
TWX_PAIR=0 python tools/launch_times.py $L
TWX_PAIR=0 python tools/launch_times.py tools/bin/libtwixt_hip_skel.so
TWX_PAIR=1 python tools/launch_times.py tools/bin/libtwixt_abl16.so
TWX_PAIR=0 python tools/launch_times.py tools/bin/libtwixt_abl16.so
} > gpurun_out/r02_launch_times.txt 2>&1
python tools/ab_bench.py $L:TWX_PAIR=1 $L:TWX_PAIR=1,TWX_PAIR_WAVES_PER_CU=4 $L:TWX_PAIR=1,TWX_PAIR_WAVES_PER_CU=6 $L:TWX_PAIR=1,TWX_DYNAMIC=0 \
  tools/bin/libtwixt_abl1.so:TWX_PAIR=1 tools/bin/libtwixt_abl2.so:TWX_PAIR=1 tools/bin/libtwixt_abl3.so:TWX_PAIR=1 tools/bin/libtwixt_abl4.so:TWX_PAIR=1 \
  tools/bin/libtwixt_abl8.so:TWX_PAIR=1 tools/bin/libtwixt_abl16.so:TWX_PAIR=1 tools/bin/libtwixt_abl31.so:TWX_PAIR=1 > gpurun_out/r02_ablate2.txt 2>&1
cat gpurun_out/r02_launch_times.txt gpurun_out/r02_ablate2.txt
