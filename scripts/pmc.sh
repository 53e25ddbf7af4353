#!/bin/bash
# PMC passes over a short bench run, aggregated per kernel.  usage: scripts/pmc.sh <tag> [bench args...]
# (counters only: gpurun refuses --pmc combined with tracing)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -f csv -d $out/p$i -o p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline "$@" > $out/p$i.log 2>&1
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
PASSES
python3 - $out <<'PY'
import sys, glob, csv, collections, json, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE"):
            seen.add(key)
    
res = {k: dict(v) for k, v in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(k)
    for c, x in sorted(v.items()):
        print("   %-36s %.4g" % (c, x))
PY
rm -rf $out/p*/  # keep only the summary (raw CSVs are large)
