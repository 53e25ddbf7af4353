#!/bin/bash
# PMC passes over the single-dispatch traversal micro-benchmark.  usage: scripts/pmc_tb.sh <tag> "<pass1 counters>" ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tb_$tag; mkdir -p $out
i=0
for pass in "$@"; do
  i=$((i+1))
  REPS=1 timeout 300 rocprofv3 --pmc $pass -f csv -d $out/p$i -o p$i -- python3 scripts/trace_bench.py > $out/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $out <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        if "k_hook" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for k, v in sorted(agg.items()):
        for c, x in sorted(v.items()):
            line = "%s %-36s %.5g" % (k, c, x); print(line); fo.write(line + "\n")
PY
rm -rf $out/p*/
