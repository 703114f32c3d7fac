#!/bin/bash
# one-off diagnostic batch (round 2): where do the stores' costs come from?
L=twixt_for_open_spiel_amd/libtwixt_hip.so
{
python tools/launch_times.py $L
python tools/launch_times.py tools/bin/libtwixt_abl512.so
python tools/launch_times.py tools/bin/libtwixt_abl514.so
python tools/launch_times.py tools/bin/libtwixt_abl16.so
} > gpurun_out/r02_launch_times2.txt 2>&1
cat gpurun_out/r02_launch_times2.txt
