#!/usr/bin/env python
"""HBM bytes per launch of the bench step's activation-sized GEMM launches from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE: separate runs, as the TCC block cannot hold both), written as profiles/pmc_traffic.json.
usage: tools/pmc_traffic.py FETCH_DIR WRITE_DIR COMMIT > pmc_traffic.json

gemm_big    = NT launches of >= 900 workgroups (the activation-sized launches, on either tile configuration)
gemm_big_tn = TN launches of >= 900 workgroups
Counters are KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the bytes of
wide coalesced streaming reads, so the read side is doubled (fetch_correction); the calibration entry checks both sides on a
kernel whose byte count is known exactly (adam_kernel: 3 reads + 3 writes... of the flat bucket)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def rows(root):
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            yield r


def collect(root, counter):
    acc = defaultdict(list)
    for r in rows(root):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        key = None
        if "gemm_nt_kernel" in name and wgs >= 900:
            key = "gemm_big"
        elif "gemm_tn_kernel" in name and wgs >= 900:
            key = "gemm_big_tn"
        elif "adam_kernel" in name:
            key = "adam"
        if key:
            acc[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
commit = sys.argv[3] if len(sys.argv) > 3 else "unknown"
out = {"_source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                  f"--no-profile --no-series --no-side-modes` on the build of commit {commit} (tools/collect_r06.sh); mean over the "
                  "dispatches of the class, KiB",
       "_comment": "fetch_correction 2.0: FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section)"}
for k in ("gemm_big", "gemm_big_tn"):
    if k in fetch and k in write:
        out[k] = {"fetch_kib": fetch[k][0], "write_kib": write[k][0], "fetch_correction": 2.0, "dispatches": fetch[k][1]}
if "adam" in fetch and "adam" in write:
    # the flat bucket is 256-byte slotted (dist.FlatTrainer pads every parameter to a slot): 16 052 KiB at configs[1], i.e.
    # 4 109 312 floats -- not the 3 726 848 live values (14 558 KiB) an earlier version of this note compared with
    bucket_kib = float(sys.argv[4]) if len(sys.argv) > 4 else 16052.0
    out["_calibration"] = {"kernel": "adam_kernel (4 reads + 3 writes of the flat fp32 bucket, 16 B per lane)",
                           "fetch_kib_x2": 2.0 * fetch["adam"][0], "write_kib": write["adam"][0], "bucket_kib": bucket_kib,
                           "fetch_ratio": 2.0 * fetch["adam"][0] / (4.0 * bucket_kib), "write_ratio": write["adam"][0] / (3.0 * bucket_kib),
                           "note": "ratios against 4 x (reads) and 3 x (writes) the 256-byte-slotted flat bucket, 16 052 KiB at configs[1] "
                                   "(bench.py: collective.allreduce_bytes_per_step_per_rank under --collective library)"}
print(json.dumps(out, indent=2))
