#!/usr/bin/env python3
"""Reduce two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of
`bench.py --steps K --warmup 0 --no-cpu-baseline --no-rollout`) to the HBM bytes per launch
and per env step of the step kernel: the `roofline.traffic` figure of bench.py.

usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT_PREFIX [K=64] [N=24] [B=1048576]
writes OUT_PREFIX_pmc_fetch.csv, OUT_PREFIX_pmc_write.csv (one row per timed launch) and
OUT_PREFIX_traffic.json.  Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE
reports half the bytes of a wide coalesced (16 B per lane) read stream and is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KB."""
import csv
import glob
import json
import os
import sys


def per_dispatch(directory, counter, kernel_tag):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under " + directory)
    acc = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter or kernel_tag not in row["Kernel_Name"]:
                    continue
                d = int(row["Dispatch_Id"])
                acc[d] = acc.get(d, 0.0) + float(row["Counter_Value"])  # summed over XCD rows
    return sorted(acc.items())


def main():
    fetch_dir, write_dir, prefix = sys.argv[1:4]
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 24
    b = int(sys.argv[6]) if len(sys.argv) > 6 else 1 << 20
    # whichever whole-batch step kernels the library runs at this board size: the big boards' journal
    # launches and compactions are two kernels (step_window1_kernel / step_window_full_kernel); the tag
    # "step_" + "<N" catches every whole-batch step kernel of the size, in launch order
    tags = ["step_window1_kernel<%d>" % n, "step_window_full_kernel<%d>" % n, "step_window_kernel<%d>" % n,
            "step_pair_kernel<%d>" % n, "step_kernel64<%d, false>" % n, "step_kernel64_small<%d, false>" % n]
    present = [t for t in tags if per_dispatch(fetch_dir, "FETCH_SIZE", t)]
    if os.environ.get("PMC_KERNEL"):
        present = [os.environ["PMC_KERNEL"]]
    tag = present[0] if present else tags[0]

    def rows_of(directory, name):  # the timed launches of all step kernels, in dispatch order
        rows = []
        for t in present:
            rows += [(d, v, t) for d, v in per_dispatch(directory, name, t)]
        return sorted(rows)[-k:]
    out, per_kernel = {}, {}
    for name, directory in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows3 = rows_of(directory, name)
        rows = [(d, v) for d, v, _ in rows3]
        for t in present:
            vals = [v for _, v, tt in rows3 if tt == t]
            if vals:
                per_kernel.setdefault(t, {})[name] = sum(vals) / len(vals)
                per_kernel[t]["launches"] = len(vals)
        if len(rows) < k:
            raise SystemExit("%s: only %d launches of %s" % (name, len(rows), tag))
        with open("%s_pmc_%s.csv" % (prefix, name.split("_")[0].lower()), "w") as fh:
            fh.write("Dispatch_Id,Counter_Name,Counter_Value_KB\n")
            for d, v in rows:
                fh.write("%d,%s,%.1f\n" % (d, name, v))
        out[name] = sum(v for _, v in rows) / len(rows)
    fetch = 2.0 * out["FETCH_SIZE"] * 1024.0
    write = out["WRITE_SIZE"] * 1024.0
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from twixt_for_open_spiel_amd import build as bld
    js = {
        # what the measurement is valid for: bench.py reports it only for this build and workload
        "source_stamp": bld.source_stamp(), "lib_sha256": bld.lib_stamp(), "board_size": n, "batch": b,
        "knobs": sorted(k for k in os.environ if k.startswith("TWX_")),
        "FETCH_SIZE_KB_per_launch_mean": out["FETCH_SIZE"],
        "WRITE_SIZE_KB_per_launch_mean": out["WRITE_SIZE"],
        "kernel": tag,
        # per kernel: mean bytes of ITS launches (FETCH_SIZE doubled as above)
        "per_kernel": {t: {"launches": v.get("launches"),
                           "fetch_bytes_corrected_per_launch": 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0,
                           "write_bytes_per_launch": v.get("WRITE_SIZE", 0.0) * 1024.0,
                           "hbm_bytes_per_launch": 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0 + v.get("WRITE_SIZE", 0.0) * 1024.0}
                       for t, v in per_kernel.items()},
        "workload": "bench.py --steps %d --warmup 0 (N=%d, B=%d): last %d launches" % (k, n, b, k),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                  "(MI355X_MICROARCH.md HBM section); FETCH_SIZE doubled: on gfx950 it reports 1/2 of the "
                  "bytes of a wide coalesced (16 B/lane) read stream; WRITE_SIZE exact for 16-B-per-lane stores",
        "fetch_bytes_corrected_per_launch": fetch,
        "write_bytes_per_launch": write,
        "hbm_bytes_per_launch": fetch + write,
        "hbm_bytes_per_env_step": (fetch + write) / b,
    }
    js["hbm_GBps_note"] = "divide hbm_bytes_per_launch by the kernel's average duration in *_kernel_stats.csv"
    with open(prefix + "_traffic.json", "w") as fh:
        json.dump(js, fh, indent=1)
    print(json.dumps(js))


if __name__ == "__main__":
    main()
