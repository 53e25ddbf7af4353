#!/bin/bash
# per-dispatch kernel durations of one C2 frame, one path group (rocprofv3 --kernel-trace): usage scripts/dispatch_trace.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/disp_$tag; mkdir -p $out
SPP=64 REPS=2 PBRHIP_STREAMS=1 rocprofv3 --kernel-trace -f csv -d $out/p -o p -- python3 scripts/render_once.py > $out/log.txt 2>&1
python3 - $out <<'PY'
import sys, glob, csv
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/p/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pb::", "")))
rows.sort()
# last frame only: from the last k_generate on
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_generate"))
rows = rows[last:]
t0 = rows[0][0]
with open(out + "/dispatches.txt", "w") as fo:
    prev_end = t0
    for s, e, k in rows:
        line = "%8.3f ms  +%7.3f  gap %6.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, k[:40])
        prev_end = e
        print(line); fo.write(line + "\n")
PY
rm -rf $out/p
