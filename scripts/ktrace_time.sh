#!/bin/bash
# scripts/ktrace_time.sh lib.so ...: k_trace time of one C2 render (64 spp, one path group) per library, WIDE on / off,
# from rocprofv3 --kernel-trace --stats (TMPDIR=/tmp)
export TMPDIR=/tmp VARIANT=${VARIANT:-ggx} SPP=${SPP:-64} PBRHIP_STREAMS=1 REPS=2
for lib in "$@"; do
  for w in 1 0; do
    d=/tmp/kt_$$_$w; rm -rf $d
    (cd /tmp && PBRHIP_LIB=$(realpath $OLDPWD/$lib) PBRHIP_WIDE=$w rocprofv3 --kernel-trace --stats -f csv -d $d -o kt -- python3 $OLDPWD/scripts/render_once.py > /dev/null 2>&1)
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "== $lib WIDE=$w"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('k_trace','k_shade_principled','k_tail')):
        print('  %-44s calls %4s total %8.2f ms' % (r['Name'][:44], r['Calls'], float(r['TotalDurationNs'])/1e6))
" 
    rm -rf $d
  done
done
