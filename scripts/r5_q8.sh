#!/bin/bash
# round 5: first run of the O tree (8-wide): parity subset, then A/B against the Q tree
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/profiles; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "soup or alternative or trace_hooks or brute_force or axis_aligned" 2>&1 | tail -5
SPP=8 timeout 300 python scripts/q8_probe.py 2>&1 | grep -v "^pbrhip: free" | tee $out/r5_q8_probe.txt
VARIANT=sss SPP=8 timeout 300 python scripts/q8_probe.py 2>&1 | grep -v "^pbrhip: free" | tee -a $out/r5_q8_probe.txt
SPP=64 bash scripts/kt.sh "c2 64spp O tree"
SPP=64 bash scripts/kt.sh "c2 64spp Q tree" PBRHIP_WIDE8=0
