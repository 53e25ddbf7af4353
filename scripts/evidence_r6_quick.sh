#!/bin/bash
# Round-6 evidence without the long parts (whole-frame parity, tolerance at the configurations' own spp): bench lines with live PMC passes,
# kernel statistics under rocprofv3, scaling proxy, C5 split
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
for w in c2 c3 c4; do python3 scripts/profile_round.py $w > $out/r6_${w}_profile.log 2>&1; cp $out/r6_${w}_pmc.json profiles/ 2>/dev/null; done
python3 bench.py --workload c2 --steps 5 --warmup 1 2> $out/r6_c2_bench.err | tail -1 > $out/r6_c2_bench.json
python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2> $out/r6_c3_bench.err | tail -1 > $out/r6_c3_bench.json
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline 2> $out/r6_c4_bench.err | tail -1 > $out/r6_c4_bench.json
SPP=64 python3 scripts/c5_split.py 2>&1 | tail -2 > $out/r6_c5_split.txt
(SCHED_CONFIGS='[{}]' REPS=5 python3 scripts/sched_ab.py ggx 2>&1 | grep "world1\|per-kernel"; python3 scripts/c5_shard.py 2>&1 | tail -3) > $out/r6_scaling_proxy.txt
for w in c2 c3 c4; do python3 -c "
import json
d=json.loads(open('$out/r6_${w}_bench.json').read())
r=d['roofline']
print('$w', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms | host_layer', round(d['host_layer']['value'],1), round(d['host_layer']['ms_per_step'],2), '| frac', round(r['frac'],3), 'solo', round(r['solo']['frac'],3), 'hbm-counter', r.get('frac_hbm_counter'), 'valu', (r.get('valu') or {}).get('frac'), 'gather', (r.get('gather') or {}).get('frac'), 'bound', r.get('bound'), '|', {k: round(v,1) for k,v in r['kernel_ms_per_step'].items() if v})
"; done
cat $out/r6_c5_split.txt | cut -c1-260; cat $out/r6_scaling_proxy.txt | cut -c1-300
