#!/bin/bash
# scripts/kt.sh "<label>" [ENV=VAL ...]: one render of a benchmark scene (VARIANT=ggx|sss|hair, SPP, one path group; KT_KERNEL: the kernel whose launches are listed, default k_trace) under
# rocprofv3 --kernel-trace with the given environment; prints the k_trace launches (ms) and the per-kernel totals.
export TMPDIR=/tmp VARIANT=${VARIANT:-ggx} SPP=${SPP:-64} PBRHIP_STREAMS=${PBRHIP_STREAMS:-1} REPS=1
label=$1; shift
for kv in "$@"; do export "$kv"; done
[ -n "$PBRHIP_LIB" ] && export PBRHIP_LIB=$(realpath $PBRHIP_LIB)
d=/tmp/kt_$$; rm -rf $d
root=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace -f csv -d $d -o kt -- python3 $root/scripts/render_once.py > $d.out 2>&1) || { echo "$label: FAILED"; tail -5 $d.out; }
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$label" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
import os
kk = os.environ.get('KT_KERNEL', 'k_trace')
tr = [dur(r) for r in rows if kk in r['Kernel_Name']]
tot = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].replace('void pb::', '').replace('pb::', '').split('(')[0]
    tot[n] = tot.get(n, 0.0) + dur(r)
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6 if rows else 0
print('== %s | %s sum %.2f ms | %s' % (sys.argv[2], kk, sum(tr), ' '.join('%.2f' % x for x in tr)))
print('   span %.2f | ' % span + ' '.join('%s %.2f' % (k, v) for k, v in tot.items() if v > 0.05))
PY
rm -rf $d $d.out
