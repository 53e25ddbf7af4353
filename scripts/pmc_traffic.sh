#!/bin/bash
# HBM-side traffic of one full C2 render (1920x1080 x SPP, default 64): FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2), summed per kernel.
# usage: scripts/pmc_traffic.sh <tag>      (env SPP, VARIANT are passed on to scripts/render_once.py)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/traffic_$tag; mkdir -p $out
i=0
for pass in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout ${PASS_TIMEOUT:-600} rocprofv3 --pmc $pass -f csv -d $out/p$i -o p$i -- python3 scripts/render_once.py > $out/p$i.log 2>&1
  echo "pass $pass rc=$?"
done
python3 - $out <<'PY'
import sys, glob, csv, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
res = {}
for k, v in sorted(agg.items()):
    n = len(disp[k])
    fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    res[k] = {"dispatches": n, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
              "hbm_bytes_per_dispatch_raw": (fetch + write) * 1024 / max(n, 1),
              "hbm_bytes_per_dispatch_fetch_x2": (2 * fetch + write) * 1024 / max(n, 1)}
    print("%-44s n=%5d fetch=%.4g KiB write=%.4g KiB  raw/launch=%.4g B" % (k[:44], n, fetch, write, res[k]["hbm_bytes_per_dispatch_raw"]))
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $out/p*/
