#!/bin/bash
# Per-dispatch timeline of ONE bench step (rocprofv3 kernel trace), with the queue each kernel ran on.  usage: tools/trace_r6.sh TAG [bench args]
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
rm -rf /tmp/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$TAG -- python bench.py --steps 3 --warmup 2 --no-profile --no-cpu-baseline --no-series --no-side-modes "$@" > $O/trace_bench_$TAG.json 2> $O/trace_$TAG.err
python - <<P > $O/trace_$TAG.txt
import csv, glob
f = glob.glob("/tmp/trace_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
prev_end = t0
qs = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gh::", "")[:60]
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%9.1f us  +%6.1f gap  %8.1f us  q%d grid %6d  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, g, name))
    tot += e - s
    prev_end = max(prev_end, e)
print("step span %.1f us, kernel time %.1f us, %d dispatches" % ((prev_end - t0) / 1e3, tot / 1e3, b - a))
P
tail -1 $O/trace_$TAG.txt
