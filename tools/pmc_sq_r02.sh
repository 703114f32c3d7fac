#!/bin/bash
# SQ counters of the step kernel in two rocprofv3 --output-format csv --kernel-trace --pmc passes (8 SQ slots each); usage: pmc_sq_r02.sh NAME [ENV=V ...]
name=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
B="python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-rollout --no-observation"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmc_$name/a -- $B > gpurun_out/pmc_$name.a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_$name/b -- $B > gpurun_out/pmc_$name.b.log 2>&1
python3 tools/pmc_counters.py gpurun_out/pmc_$name.json gpurun_out/pmc_$name/a gpurun_out/pmc_$name/b > gpurun_out/pmc_$name.txt 2>&1
rm -rf gpurun_out/pmc_$name   # raw CSVs are large; the reduction is kept
cat gpurun_out/pmc_$name.txt
