#!/bin/bash
# Profiles of the shipped binary for profiles/<tag>_*: rocprofv3 kernel stats of the default bench
# run, HBM traffic of the step kernel (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ counters.
# usage (on the GPU box, repo root): bash tools/profile_round.sh r02_final
set -e
tag=$1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-rollout --no-observation"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/stats -- $B --steps 500 --warmup 20 > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/fetch -- $B --steps 64 --warmup 0 > $out/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/write -- $B --steps 64 --warmup 0 > $out/write.log 2>&1
python3 tools/pmc_traffic.py $out/fetch $out/write $out/$tag
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $out/sqa -- $B --steps 64 --warmup 0 > $out/sqa.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/sqb -- $B --steps 64 --warmup 0 > $out/sqb.log 2>&1
python3 tools/pmc_counters.py $out/${tag}_sq_counters.json $out/sqa $out/sqb > /dev/null
# the bench line of the same binary, with the traffic just measured (bench.py only believes a
# traffic file that carries this build's hash)
cp $out/${tag}_traffic.json profiles/traffic_step_kernel_n24.json
python3 bench.py --steps 500 --warmup 20 > $out/${tag}_bench.json 2> $out/bench.err
rm -rf $out/stats $out/fetch $out/write $out/sqa $out/sqb
ls -la $out
