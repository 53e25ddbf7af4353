#!/bin/bash
# Round-6 evidence on a GPU box, written to gpurun_out/profiles/ (copy what is to be judged into profiles/):
#   r6_<w>_pmc.json, r6_<w>_kernel_stats.csv   scripts/profile_round.py c2|c3|c4 (PMC passes of one render; rocprofv3 --kernel-trace --stats of the bench command)
#   r6_<w>_bench.json                          `python bench.py --workload <w> --steps 3 --warmup 1` (C2 with the CPU baseline) -- run AFTER the PMC passes,
#                                              so that the line carries the counter-based fields of the same kernel sources
#   r6_whole_frame_parity.txt                  scripts/whole_frame_check.py: the complete 1920x1080 frames of C2 / C3 / C4, GPU vs oracle, bit for bit
#   r6_tolerance_at_config_spp.txt             pytest -k own_spp: the 1e-4 bar against oracle[libm] at the configurations' own sample counts (asserted, not projected)
#   r6_c5_split.txt, r6_scaling_proxy.txt      per-kernel split of the C5 frame at 64 spp; rank 0's share of 2 / 4 / 8-rank jobs (C2), C5 eighth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
for w in c2 c3 c4; do python3 scripts/profile_round.py $w > $out/r6_${w}_profile.log 2>&1; cp $out/r6_${w}_pmc.json profiles/ 2>/dev/null; done
python3 bench.py --workload c2 --steps 5 --warmup 1 2> $out/r6_c2_bench.err | tail -1 > $out/r6_c2_bench.json
python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2> $out/r6_c3_bench.err | tail -1 > $out/r6_c3_bench.json
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline 2> $out/r6_c4_bench.err | tail -1 > $out/r6_c4_bench.json
SPP=${WF_SPP:-8} python3 scripts/whole_frame_check.py > $out/r6_whole_frame_parity.txt 2>&1
SPP=64 python3 scripts/c5_split.py 2>&1 | tail -2 > $out/r6_c5_split.txt
(SCHED_CONFIGS='[{}]' REPS=5 python3 scripts/sched_ab.py ggx 2>&1 | grep "world1\|per-kernel"; python3 scripts/c5_shard.py 2>&1 | tail -3) > $out/r6_scaling_proxy.txt
for w in c2 c3 c4; do python3 -c "
import json
d=json.loads(open('$out/r6_${w}_bench.json').read())
r=d['roofline']
print('$w', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms | host_layer', round(d['host_layer']['value'],1), round(d['host_layer']['ms_per_step'],2), '| frac', round(r['frac'],3), 'solo', round(r['solo']['frac'],3), 'hbm-counter', r.get('frac_hbm_counter'), 'valu', (r.get('valu') or {}).get('frac'), 'gather', (r.get('gather') or {}).get('frac'), 'bound', r.get('bound'), '|', {k: round(v,1) for k,v in r['kernel_ms_per_step'].items() if v})
"; done
timeout 2400 python -m pytest tests/test_gpu_configs.py -x -q -s -m gpu -k "own_spp" 2>&1 | grep -oE "host libm:.*|c[1-5]: [0-9]+x.*|[0-9]+ passed.*|[0-9]+ failed.*|[0-9]+ xfailed.*" | sort -u > $out/r6_tolerance_at_config_spp.txt
cat $out/r6_tolerance_at_config_spp.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/r6_smoke.txt
cat $out/r6_whole_frame_parity.txt | grep differing; cat $out/r6_c5_split.txt | cut -c1-260; cat $out/r6_scaling_proxy.txt | cut -c1-260
