#!/bin/bash
# second part of the round-4 evidence: the measured libm tolerance at the configurations' own spp, the C2 bench line with the CPU baseline,
# the gather micro-benchmark, the oracle's thread scaling on the host
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_configs.py -x -q -s -m gpu -k "own_spp or both_math" 2>&1 | grep -oE "c[1-5]: [0-9]+x.*|[0-9]+ passed.*|[0-9]+ failed.*" > $out/r4_tolerance_at_config_spp.txt
python3 bench.py --workload c2 --steps 3 --warmup 1 2> $out/r4_c2_bench.err | tail -1 > $out/r4_c2_bench.json
timeout 300 scripts/ubench/vmem_quads > $out/r4_gather_ubench.txt 2>&1
timeout 300 python scripts/cpu_scaling.py > $out/r4_cpu_scaling.txt 2>&1
cat $out/r4_tolerance_at_config_spp.txt; python3 -c "
import json
d=json.loads(open('$out/r4_c2_bench.json').read()); c=d['cpu_baseline']
print(round(d['value'],1), round(d['ms_per_step'],2), {k: c[k] for k in ('value','cores','threads','hardware_threads','speedup_over_one_thread','parallel_efficiency','schedule_efficiency')}, c['one_thread'])
"
