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
# the same trace reduced to the TIMED launches (the stats average also covers the prologue, whose boards are
# young and fetch fewer windows, and the launches timed one by one afterwards): which launches were timed is in the
# bench line of the traced run (prologue, warm-up, steps x blocks)
python3 - "$(find $out/stats -name '*kernel_trace.csv' | head -1)" $out/bench_under_rocprof.json $out/${tag}_kernel_timed.json <<'PY'
import csv, json, sys
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
first = b["config"]["prologue_steps"] + b["warmup"]
count = b["steps"] * b["timed_blocks"]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_" in r["Kernel_Name"] and "<24>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
timed = dur[first:first + count]
json.dump({"launches_in_trace": len(rows), "first_timed_launch": first, "timed_launches": len(timed),
           "mean_duration_us_timed": sum(timed) / len(timed), "mean_duration_us_all": sum(dur) / len(dur),
           "hip_event_kernel_ms_of_the_same_run": b["roofline"]["kernel_ms"],
           "note": "rocprofv3 --kernel-trace of bench.py --steps 500 --warmup 20; the HIP-event figure of the same run "
                   "also holds the gaps between launches"},
          open(sys.argv[3], "w"), indent=1)
PY
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
