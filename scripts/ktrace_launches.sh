#!/bin/bash
# scripts/ktrace_launches.sh lib.so: per-launch k_trace durations (ms) of one C2 render (64 spp, one path group), WIDE=1 and WIDE=0
export TMPDIR=/tmp VARIANT=${VARIANT:-ggx} SPP=${SPP:-64} PBRHIP_STREAMS=1 REPS=1
lib=$(realpath $1)
for w in 1 0; do
  d=/tmp/ktl_$$_$w; rm -rf $d
  (cd /tmp && PBRHIP_LIB=$lib PBRHIP_WIDE=$w rocprofv3 --kernel-trace -f csv -d $d -o kt -- python3 $OLDPWD/scripts/render_once.py > /dev/null 2>&1)
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    if 'k_trace' in r['Kernel_Name']:
        out.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print('WIDE=$w k_trace launches (ms):', ' '.join('%.2f'%x for x in out), '| sum %.2f'%sum(out))
for name in ('k_shade_principled','k_classify','k_compact','k_tail','k_generate'):
    v=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows if name in r['Kernel_Name']]
    print('   ', name, '%.2f' % sum(v), '|', ' '.join('%.2f'%x for x in v))
"
  rm -rf $d
done
