#!/bin/bash
# L2 (TCC) hits and misses of the step kernel per launch: how much of a launch is served by what the
# launch before left in the XCDs' L2s.  usage (GPU box): bash tools/pmc_tcc.sh [TWX_PINGPONG value]
export TMPDIR=/tmp
O=gpurun_out/pmc_tcc
rm -rf $O; mkdir -p $O
[ -n "$1" ] && export TWX_PINGPONG=$1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ -d $O/k -- python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-rollout --no-observation > $O/k.log 2>&1
python3 tools/pmc_counters.py $O/tcc.json $O/k
rm -rf $O/k
