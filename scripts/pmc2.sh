#!/bin/bash
# quick PMC passes on a small render (guards: per-pass timeout).  usage: scripts/pmc2.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $pass -f csv -d $out/p$i -o p$i -- python3 scripts/render_once.py > $out/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $out <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for k, v in sorted(agg.items()):
        if not k.startswith("pb::"): continue
        line = k + " | " + "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()))
        print(line); fo.write(line + "\n")
PY
rm -rf $out/p*/
