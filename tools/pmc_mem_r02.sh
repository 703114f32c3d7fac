#!/bin/bash
# memory-side counters of the step kernel and of the streaming probe (round-2 diagnostic)
export TMPDIR=/tmp
O=gpurun_out/pmc_mem
mkdir -p $O
B="python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-rollout --no-observation"
P="./tools/bin/window_probe2"
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_INST_LEVEL_LDS" \
           "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL" \
           "TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_THRASHING_STALL" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_REQ"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/k$i -- $B > $O/k$i.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/p$i -- $P > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag, pat, kern in (("step_pair_kernel", "gpurun_out/pmc_mem/k*", "step_pair_kernel<24>"), ("probe_stream_all", "gpurun_out/pmc_mem/p*", "probe<1, 4>")):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                per[(row["Counter_Name"], int(row["Dispatch_Id"]))] += float(row["Counter_Value"])
        byname = collections.defaultdict(list)
        for (name, d), v in sorted(per.items(), key=lambda kv: kv[0][1]):
            byname[name].append(v)
        for name, vals in byname.items():
            vals = vals[-40:]
            acc[name] = sum(vals) / len(vals)
    out[tag] = dict(acc)
json.dump(out, open("gpurun_out/r02_pmc_mem.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O
