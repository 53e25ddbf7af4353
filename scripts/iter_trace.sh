#!/bin/bash
# per-iteration kernel durations of one render (rocprofv3 kernel trace).  usage: scripts/iter_trace.sh <variant> <spp>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/it && VARIANT=$1 SPP=$2 rocprofv3 --kernel-trace -f csv -d gpurun_out/it -o it -- python3 scripts/render_once.py > gpurun_out/it.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/it/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
it = []; cur = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pb::", "").split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if k == "k_trace_closest" and cur: it.append(cur); cur = {}
    cur[k] = cur.get(k, 0) + d
it.append(cur)
print("iterations", len(it))
keys = ["k_trace_closest", "k_classify", "k_shade_principled", "k_shade_hair", "k_sss_step", "k_compact", "k_trace_shadow"]
for i, c in enumerate(it):
    if i < 30 or i % 20 == 0: print(i, " ".join("%s=%.0f" % (k[2:], c.get(k, 0)) for k in keys))
tot = collections.Counter()
for c in it:
    for k, v in c.items(): tot[k] += v
print({k: round(v / 1e3, 1) for k, v in tot.items()})
PY
rm -rf gpurun_out/it
