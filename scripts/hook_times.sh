#!/bin/bash
# scripts/hook_times.sh N...: per-call durations (us) of the hook kernels of scripts/quad_check.py for N rays in the C2 scene
export TMPDIR=/tmp
root=$(pwd)
for N in "$@"; do
  export N
  d=/tmp/qc_$N; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace -f csv -d $d -o qc -- python3 $root/scripts/quad_check.py 2>&1 | grep -E "c2 scene|rror")
  python3 - $d <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t = collections.OrderedDict()
for r in rows:
    if "hook" in r["Kernel_Name"]:
        n = r["Kernel_Name"].replace("void pb::", "").split("(")[0]
        t.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in t.items():
    print("  %-32s last 4 calls (us): %s" % (n, " ".join("%.0f" % x for x in v[-4:])))
PY
done
