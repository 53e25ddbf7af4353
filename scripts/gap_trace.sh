#!/bin/bash
# kernel timeline of one rank's share of a render: busy time vs gaps.  usage: scripts/gap_trace.sh <variant> <spp> <world>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gt && VARIANT=$1 SPP=$2 WORLD=$3 rocprofv3 --kernel-trace -f csv -d gpurun_out/gt -o gt -- python3 scripts/render_once.py > gpurun_out/gt.log 2>&1
tail -1 gpurun_out/gt.log | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"])
rows = rows[first:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("kernels %d  span %.2f ms  busy %.2f ms  gaps %.2f ms" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
per = collections.defaultdict(lambda: [0, 0.0])
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pb::", "").split("<")[0]
    per[k][0] += 1; per[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in sorted(per.items(), key=lambda x: -x[1][1]):
    print("  %-22s n=%4d total %.2f ms avg %.1f us" % (k, n, us / 1e3, us / n))
import statistics
g = sorted(gaps)
print("gaps: n=%d median %.1f us p90 %.1f us max %.1f us; >20us: %d totalling %.2f ms" % (len(g), statistics.median(g), g[int(len(g) * .9)], g[-1], sum(1 for x in g if x > 20), sum(x for x in g if x > 20) / 1e3))
# k_trace durations over iterations
tr = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_trace" in r["Kernel_Name"]]
print("k_trace us:", " ".join("%.0f" % x for x in tr))
PY
rm -rf gpurun_out/gt
